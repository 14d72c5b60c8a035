#!/bin/bash
# SparseOptimizer::optimize() through the g2o plugin (mini-g2o host, tests/cpp/mini_g2o) at BASELINE configs 3 and 4, per phase:
# the algorithm's BatchStatistics + the adapter's own split (G2OHIP_ADAPTER_TIMING=1).  Output: gpurun_out/adapter_optimize.jsonl
make -s -C tests/cpp/mini_g2o || exit 1
B=tests/cpp/mini_g2o/build
OUT=gpurun_out/adapter_optimize.jsonl
: > $OUT
run() {  # tag P L iterations env...   (SOLVER=lm_fix6_3_hipdev in the environment: the device-resident driver)
  tag=$1; P=$2; L=$3; it=$4; shift 4
  env G2OHIP_ADAPTER_TIMING=1 "$@" $B/g2o_host none $B/libg2o_solver_hip.so ${SOLVER:-lm_fix6_3_hip} $it /tmp/ab.json bench:$P:$L:5 2> /tmp/ab.err || { tail -5 /tmp/ab.err; return; }
  python3 - "$tag" <<'EOP' >> gpurun_out/adapter_optimize.jsonl
import json, sys
d = json.load(open("/tmp/ab.json"))
ph = [json.loads(l) for l in open("/tmp/ab.err") if l.startswith("{\"g2ohip_adapter_phases_ms\"")]
d["tag"] = sys.argv[1]
d["adapter"] = ph[-1]["g2ohip_adapter_phases_ms"] if ph else None
print(json.dumps(d))
EOP
  tail -1 $OUT | cut -c1-400
}
run config3_fast_pinned      50000  500000 6
run config3_fast_pageable    50000  500000 6 G2OHIP_ADAPTER_PINNED=0
run config3_generic_pinned   50000  500000 4 G2OHIP_ADAPTER_FASTPATH=0
run config4_fast_pinned     100000 1000000 6
run config4_fast_pageable   100000 1000000 6 G2OHIP_ADAPTER_PINNED=0
run config4_generic_pinned  100000 1000000 4 G2OHIP_ADAPTER_FASTPATH=0
SOLVER=lm_fix6_3_hipdev run config3_device_loop      50000  500000 8
SOLVER=lm_fix6_3_hipdev run config4_device_loop     100000 1000000 8
SOLVER=lm_fix6_3_hipdev run config4_device_loop_no_writeback 100000 1000000 8 G2OHIP_ADAPTER_WRITEBACK=0
