#!/bin/bash
python -m pytest tests/test_gpu_band_chain.py -m gpu -x -q > gpurun_out/r5d_tests.log 2>&1; tail -4 gpurun_out/r5d_tests.log
for tb in 2 1; do
python bench.py --opt tree_backward=$tb --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5d_b$tb.log 2>&1; python - <<EOP
import json
d=json.loads([l for l in open("gpurun_out/r5d_b$tb.log") if l.startswith("{")][-1])
print("tree_backward=$tb", d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, d["residual_rel"])
EOP
done
TAG=tb3 bash tools/gpu_ktrace.sh --opt tree_backward=2
