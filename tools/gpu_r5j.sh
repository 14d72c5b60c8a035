#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_adapter.py tests/test_gpu_multi_edge.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r5j_tests.log 2>&1; tail -8 gpurun_out/r5j_tests.log
B=tests/cpp/mini_g2o/build
for mode in "bench:100000:1000000:5:prior:huber" "bench:100000:1000000:5:huber"; do
for solver in lm_fix6_3_hipdev; do
for hy in 1 0; do
env G2OHIP_ADAPTER_TIMING=1 G2OHIP_ADAPTER_HYBRID=$hy timeout 600 $B/g2o_host none $B/libg2o_solver_hip.so $solver 6 /tmp/ab.json $mode 2> /tmp/ab.err
python3 - <<EOP
import json
d = json.load(open("/tmp/ab.json"))
its = d["iterations"]
print("$mode $solver hybrid=$hy: iteration0 %.3f s, then %.3f ms per LM iteration (write-back %.3f ms), chi2 %s" % (its[0]["iteration_s"], 1e3 * sum(i["iteration_s"] for i in its[1:]) / (len(its) - 1), 1e3 * sum(i["timeUpdate"] for i in its[1:]) / (len(its) - 1), [i["chi2"] for i in its][-1]))
EOP
done; done; done
