#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_band_chain.py -m gpu -x -q > gpurun_out/r5h_tests.log 2>&1; tail -6 gpurun_out/r5h_tests.log
for tf in 1 0; do
timeout 300 python bench.py --opt tree_factor=$tf --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5h_b$tf.log 2>&1; python - <<EOP
import json
d=json.loads([l for l in open("gpurun_out/r5h_b$tf.log") if l.startswith("{")][-1])
print("tree_factor=$tf", d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, d["residual_rel"])
EOP
done
TAG=tf bash tools/gpu_ktrace.sh
