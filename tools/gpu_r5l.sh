#!/bin/bash
for o in "ba_lazy_pose=0" "ba_lazy_pose=1"; do
python bench.py --opt $o --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5l.log 2>&1; python - <<EOP
import json
d=json.loads([l for l in open("gpurun_out/r5l.log") if l.startswith("{")][-1])
print("$o", d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, d["residual_rel"])
EOP
done
python -m pytest tests -m gpu -x -q > gpurun_out/r5l_tests.log 2>&1; tail -3 gpurun_out/r5l_tests.log
