#!/bin/bash
# band-chain kernel: its tests, then the bench with band_kernel off and on
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_band_chain.py -m gpu -x -q > gpurun_out/band_tests.log 2>&1; tail -25 gpurun_out/band_tests.log
for opt in ${OPTS:-"band_kernel=0" "band_kernel=1"}; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --opt $opt > gpurun_out/band_$opt.log 2>&1
  python - "$opt" <<EOP
import json, sys
try:
    d=json.loads([l for l in open("gpurun_out/band_%s.log" % sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, d["residual_rel"], d["solve_ok"])
except Exception as e:
    print(sys.argv[1], "failed", e); print(open("gpurun_out/band_%s.log" % sys.argv[1]).read()[-1500:])
EOP
done
