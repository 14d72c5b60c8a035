#!/bin/bash
for o in "schur_group=8" "schur_group=4" "schur_group=16"; do
python bench.py --opt $o --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5o.log 2>&1; python - <<EOP
import json
d=json.loads([l for l in open("gpurun_out/r5o.log") if l.startswith("{")][-1])
print("$o", d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k in ("schur_tiles",)}, d["residual_rel"])
EOP
done
