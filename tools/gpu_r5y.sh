#!/bin/bash
# hybrid loop after the selected-vertex read-back: adapter + multi-edge tests, then the metric configuration with a prior edge
make -s -C tests/cpp/mini_g2o || exit 1
timeout 900 python -m pytest tests/test_gpu_adapter.py tests/test_gpu_multi_edge.py -m gpu -x -q > gpurun_out/r5y_tests.log 2>&1; tail -4 gpurun_out/r5y_tests.log
B=tests/cpp/mini_g2o/build
for mode in "bench:100000:1000000:5:prior:huber:tight" "bench:100000:1000000:5:prior:huber"; do
for rep in 1 2; do
timeout 600 $B/g2o_host none $B/libg2o_solver_hip.so lm_fix6_3_hipdev 10 /tmp/ab.json $mode 2> /tmp/ab.err
python3 - <<EOP
import json
d = json.load(open("/tmp/ab.json"))
its = d["iterations"]
print("$mode lm_fix6_3_hipdev hybrid: iteration0 %.3f s, then %.3f ms per LM iteration (write-back %.3f ms), chi2 %s" % (its[0]["iteration_s"], 1e3 * sum(i["iteration_s"] for i in its[2:]) / (len(its) - 2), 1e3 * sum(i["timeUpdate"] for i in its[2:]) / (len(its) - 2), [i["chi2"] for i in its][-1]))
EOP
done; done 2>&1 | tee gpurun_out/r5y_hybrid.txt
