# Schur tiles at five workgroups per CU (variant build with -DG2OHIP_SCHUR_OCC=5 and smaller tiles) against four
for t in 39936 31744; do python bench.py --no-cpu-baseline --opt schur_tile_bytes=$t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('occ4 tile $t', round(d['value'],4), round(d['kernels']['schur_tiles']['avg_ms'],4))"; done
for t in 31744 28672; do G2OHIP_LIB=$PWD/variants/occ5/libg2ohip.so python bench.py --no-cpu-baseline --opt schur_tile_bytes=$t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('occ5 tile $t', round(d['value'],4), round(d['kernels']['schur_tiles']['avg_ms'],4))"; done
