#!/bin/bash
# look-ahead soak: long LM runs (rejected trials included) with and without the look-ahead, final chi2 and trial counts must be equal
B=tests/cpp/mini_g2o/build
for mode in "bench:3000:30000:5:tight:huber" "bench:3000:30000:5:tight:huber:prior" "bench:20000:200000:5:tight"; do
for la in 1 0; do
  G2OHIP_ADAPTER_LOOKAHEAD=$la $B/g2o_host none $B/libg2o_solver_hip.so lm_fix6_3_hipdev 150 /tmp/soak_$la.json $mode 2> /tmp/soak.err || { echo "rc $?"; tail -3 /tmp/soak.err; }
done
python3 - "$mode" <<'EOP'
import json, sys
a = json.load(open("/tmp/soak_1.json")); b = json.load(open("/tmp/soak_0.json"))
ta = [i["levenbergIterations"] for i in a["iterations"]]; tb = [i["levenbergIterations"] for i in b["iterations"]]
print(sys.argv[1], "iterations", len(ta), len(tb), "trials equal", ta == tb, "rejections", sum(t - 1 for t in ta), "final chi2", a["iterations"][-1]["chi2"], b["iterations"][-1]["chi2"], "equal", a["iterations"][-1]["chi2"] == b["iterations"][-1]["chi2"])
EOP
done 2>&1 | tee gpurun_out/r5_lookahead_soak.txt
