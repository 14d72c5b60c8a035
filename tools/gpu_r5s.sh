#!/bin/bash
# look-ahead trial of the device-resident LM driver: adapter tests, then solve(i) back to back at the metric configuration
# (bench:...:tight) with the look-ahead on / off
make -s -C tests/cpp/mini_g2o || exit 1
python -m pytest tests/test_gpu_adapter.py -x -q 2>&1 | tail -8
B=tests/cpp/mini_g2o/build
for rep in 1 2; do
for la in 1 0; do
for opt in ":tight" ":tight:huber"; do
  G2OHIP_ADAPTER_LOOKAHEAD=$la $B/g2o_host none $B/libg2o_solver_hip.so lm_fix6_3_hipdev 12 /tmp/la.json bench:100000:1000000:5$opt 2> /tmp/la.err || { tail -5 /tmp/la.err; continue; }
  python3 - $la "$opt" <<'EOP'
import json, sys
d = json.load(open("/tmp/la.json"))
it = [i["iteration_s"] * 1e3 for i in d["iterations"]]
print("lookahead", sys.argv[1], sys.argv[2], "iteration ms:", " ".join("%.2f" % v for v in it), "| mean of 2..11: %.3f" % (sum(it[2:]) / len(it[2:])), "trials", [i["levenbergIterations"] for i in d["iterations"]], "chi2", d["iterations"][-1]["chi2"])
EOP
done; done; done 2>&1 | tee gpurun_out/r5s_lookahead.txt
