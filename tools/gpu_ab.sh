for o in "$@"; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --opt $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$o:', round(d['value'],4), round(d['kernels']['chol_factor(all levels)']['avg_ms'],4), d['residual_rel'])"
done
