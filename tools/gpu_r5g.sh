#!/bin/bash
python -m pytest tests/test_gpu_multirank.py -m gpu -x -q > gpurun_out/r5g_tests.log 2>&1; tail -4 gpurun_out/r5g_tests.log
timeout 900 python bench.py --gpus 8 --comm staged --check-oracle --steps 5 --warmup 3 > gpurun_out/r5_staged8_fullsize.json 2> gpurun_out/r5_staged8_fullsize.err; tail -3 gpurun_out/r5_staged8_fullsize.err
python - <<EOP
import json
d=json.loads([l for l in open("gpurun_out/r5_staged8_fullsize.json") if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "n_gpus", "solve_ok", "collectives", "dx_pose_rel_err")}, d.get("shard"))
EOP
bash tools/gpu_r5f.sh
