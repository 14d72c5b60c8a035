for t in 39936 45056 53248 65536; do python bench.py --no-cpu-baseline --opt schur_tile_bytes=$t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile $t', round(d['value'],4), round(d['kernels']['schur_tiles']['avg_ms'],4), round(d['kernels']['chol_factor(band chains)']['avg_ms'],4))"; done
