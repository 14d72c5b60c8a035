#!/bin/bash
python -m pytest tests/test_gpu_adapter.py tests/test_gpu_edge_classes.py tests/test_gpu_multi_edge.py tests/test_gpu_lm.py -m gpu -x -q > gpurun_out/r5e_tests.log 2>&1; tail -4 gpurun_out/r5e_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5e_b.log 2>&1; python - <<EOP
import json
d=json.loads([l for l in open("gpurun_out/r5e_b.log") if l.startswith("{")][-1])
print("bench", d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, d["residual_rel"])
EOP
make -s -C tests/cpp/mini_g2o || exit 1
B=tests/cpp/mini_g2o/build
OUT=gpurun_out/r5_adapter_optimize.jsonl
: > $OUT
run() {  # tag P L iterations env...
  tag=$1; P=$2; L=$3; it=$4; shift 4
  env G2OHIP_ADAPTER_TIMING=1 "$@" $B/g2o_host none $B/libg2o_solver_hip.so ${SOLVER:-lm_fix6_3_hipdev} $it /tmp/ab.json bench:$P:$L:5 2> /tmp/ab.err || { tail -5 /tmp/ab.err; return; }
  python3 - "$tag" <<'EOP' >> gpurun_out/r5_adapter_optimize.jsonl
import json, sys
d = json.load(open("/tmp/ab.json"))
ph = [json.loads(l) for l in open("/tmp/ab.err") if l.startswith("{\"g2ohip_adapter_phases_ms\"")]
d["tag"] = sys.argv[1]
d["adapter"] = ph[-1]["g2ohip_adapter_phases_ms"] if ph else None
its = d["iterations"]
d["ms_per_lm_iteration_after_first"] = 1e3 * sum(i["iteration_s"] for i in its[1:]) / max(1, len(its) - 1)
d["timeUpdate_ms_after_first"] = 1e3 * sum(i["timeUpdate"] for i in its[1:]) / max(1, len(its) - 1)
print(json.dumps(d))
EOP
  python3 -c "
import json
d=json.loads(open('$OUT').read().strip().split('\n')[-1]); print(d['tag'], 'iteration0 %.3f s' % d['iterations'][0]['iteration_s'], 'then %.3f ms / LM iteration' % d['ms_per_lm_iteration_after_first'], 'write-back %.3f ms' % d['timeUpdate_ms_after_first'], 'chi2 last', d['iterations'][-1]['chi2'])"
}
run config3_device_loop              50000  500000 8
run config4_device_loop             100000 1000000 8
run config4_device_loop_threads16   100000 1000000 8 G2OHIP_ADAPTER_THREADS=16
run config4_device_loop_threads32   100000 1000000 8 G2OHIP_ADAPTER_THREADS=32
run config4_device_loop_pageable    100000 1000000 8 G2OHIP_ADAPTER_PINNED=0
run config4_device_loop_no_writeback 100000 1000000 8 G2OHIP_ADAPTER_WRITEBACK=0
