#!/bin/bash
# Schur tiles at five waves per SIMD (variants/occ5: -DG2OHIP_SCHUR_OCC=5, 96 VGPRs, spills) against the product (four), at the tile
# sizes that fit five / four workgroups per CU
for rep in 1 2; do
for cfg in "product:" "product:schur_tile_bytes=31744" "occ5:schur_tile_bytes=31744" "occ5:"; do
v=${cfg%%:*}; o=${cfg#*:}
if [ "$v" = product ]; then L=$PWD/openslam_g2o_amd/lib/libg2ohip.so; else L=$PWD/variants/$v/libg2ohip.so; fi
OPT=""; [ -n "$o" ] && OPT="--opt $o"
G2OHIP_LIB=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline $OPT 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$cfg', round(d['value'],4), 'tiles', round(d['kernels']['schur_tiles']['avg_ms'],4), 'band', round(d['kernels']['chol_factor(band chains)']['avg_ms'],4))"
done
done
