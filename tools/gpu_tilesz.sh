#!/bin/bash
# Schur tile size (LDS bytes per tile = workgroups per CU) against partial blocks per destination and kernel time
for rep in 1 2; do
for b in 39936 53248 79872 31744; do
G2OHIP_PLAN_DUMP=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --opt schur_tile_bytes=$b 2> gpurun_out/tilesz.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('schur_tile_bytes=$b', 'step', round(d['value'],4), 'tiles', round(k['schur_tiles']['avg_ms'],4), 'band', round(k['chol_factor(band chains)']['avg_ms'],4), 'tree', round(k['chol_factor(all levels)']['avg_ms'],4))"
grep "schur tiles:" gpurun_out/tilesz.err | tail -1
done
done
