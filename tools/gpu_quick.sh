#!/bin/bash
# quick check on the GPU box: parity tests (output kept in gpurun_out/quick_tests.log) + one bench line
python -m pytest ${QUICK_TESTS:-tests/test_gpu_parity.py tests/test_gpu_lm.py} -m gpu -x -q > gpurun_out/quick_tests.log 2>&1; tail -3 gpurun_out/quick_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/b3.log 2>&1; python - <<EOP
import json
d=json.loads([l for l in open("gpurun_out/b3.log") if l.startswith("{")][-1])
print(d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, d["residual_rel"])
EOP
