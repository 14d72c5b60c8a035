# phase costs of the Schur tile kernel: the kernel with parts switched off (results are wrong then: timing only)
#   2: no destination loop, 8: no linearisation of the observations (10 = both)
for a in 0 2 8 10; do
G2OHIP_SCHUR_ABL=$a python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('abl $a', round(d['kernels']['schur_tiles']['avg_ms'],4))"
done
