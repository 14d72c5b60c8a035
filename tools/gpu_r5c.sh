#!/bin/bash
python -m pytest tests/test_gpu_band_chain.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -x -q -s > gpurun_out/r5c_tests.log 2>&1; tail -4 gpurun_out/r5c_tests.log
for tb in 1 2 0; do
python bench.py --opt tree_backward=$tb --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5c_b$tb.log 2>&1; python - <<EOP
import json
d=json.loads([l for l in open("gpurun_out/r5c_b$tb.log") if l.startswith("{")][-1])
print("tree_backward=$tb", d["value"], {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()}, d["residual_rel"])
EOP
done
TAG=tb2 bash tools/gpu_ktrace.sh
for n in 2 4 8; do python bench.py --emulate 0/$n --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('emulate 0/$n', d['value'])"; done
