#!/bin/bash
make -s -C tests/cpp/mini_g2o || exit 1
B=tests/cpp/mini_g2o/build
for la in 1 0; do
  G2OHIP_ADAPTER_VERBOSE=1 G2OHIP_ADAPTER_LOOKAHEAD=$la $B/g2o_host none $B/libg2o_solver_hip.so gn_fix6_3_hipdev 10 /tmp/la.json bench:100000:1000000:5:tight 2> /tmp/la.err; rc=$?
  echo "rc $rc"; tail -3 /tmp/la.err
  [ $rc = 0 ] && python3 - $la <<'EOP'
import json, sys
d = json.load(open("/tmp/la.json"))
it = [i["iteration_s"] * 1e3 for i in d["iterations"]]
print("gn lookahead", sys.argv[1], "iteration ms:", " ".join("%.2f" % v for v in it), "| mean of 2..: %.3f" % (sum(it[2:]) / len(it[2:])), "chi2", d["iterations"][-1]["chi2"])
EOP
done 2>&1 | tee gpurun_out/r5u_gn_lookahead.txt
