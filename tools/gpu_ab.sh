for v in "$@"; do
G2OHIP_LIB=$PWD/variants/$v/libg2ohip.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v:', round(d['value'],4), round(d['roofline']['avg_launch_ms'],4), d['residual_rel'])"
done
