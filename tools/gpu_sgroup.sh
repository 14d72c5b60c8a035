for g in 2 4 8 16; do python bench.py --no-cpu-baseline --opt schur_group=$g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('schur_group $g', round(d['value'],4), round(d['kernels']['schur_tiles']['avg_ms'],4))"; done
