#!/bin/bash
# round 6: new three-phase wave_front factorisation: parity tests, bench, kernel trace
python -m pytest tests/test_gpu_band_chain.py tests/test_gpu_parity.py tests/test_gpu_posegraph.py tests/test_gpu_lm.py -m gpu -x -q > gpurun_out/r6a_tests.log 2>&1; tail -5 gpurun_out/r6a_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6a_bench.log 2>&1; python - <<EOP
import json
d=json.loads([l for l in open("gpurun_out/r6a_bench.log") if l.startswith("{")][-1])
print(d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, d["residual_rel"])
EOP
TAG=r6a bash tools/gpu_ktrace.sh
python tools/posegraph_solve_time.py 2>&1 | tail -12
