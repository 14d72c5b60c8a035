python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -5
for o in 1 0; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --opt ba_skip_hpl=$o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('skip_hpl=$o', round(d['value'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()}, d['residual_rel'])"; done
