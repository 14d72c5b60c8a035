#!/bin/bash
make -s -C tests/cpp/mini_g2o || exit 1
B=tests/cpp/mini_g2o/build
for th in ""; do
env G2OHIP_ADAPTER_TIMING=1 G2OHIP_SETUP_TIMING=1 ${th:+G2OHIP_ADAPTER_THREADS=$th} $B/g2o_host none $B/libg2o_solver_hip.so lm_fix6_3_hipdev 8 /tmp/ab.json bench:100000:1000000:5 2> gpurun_out/r5f_setup_$th.err
python3 - <<EOP
import json
d = json.load(open("/tmp/ab.json"))
its = d["iterations"]
print("threads=$th graph_s", d["graph_s"], "init_s", d["initializeOptimization_s"], "iteration0_s", its[0]["iteration_s"], "then ms/it", 1e3 * sum(i["iteration_s"] for i in its[1:]) / (len(its) - 1), "timeUpdate ms", 1e3 * sum(i["timeUpdate"] for i in its[1:]) / (len(its) - 1))
EOP
grep -E "g2ohip_adapter|setup" gpurun_out/r5f_setup_$th.err | cut -c1-900 | head -30
done
python tools/probe/setup_time.py
