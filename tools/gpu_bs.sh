python -m pytest tests/test_gpu_lm.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -m gpu -q -x 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -5
for o in 1 0; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --opt ba_recompute_backsub=$o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('recompute=$o', round(d['value'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()}, d['residual_rel'])"; done
