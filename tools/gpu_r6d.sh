#!/bin/bash
# A/B of product-library variants inside one call (3 rounds): bash tools/gpu_r6d.sh [bench args --] libA libB ...  ("product" = the in-tree library)
ARGS=""
if [ "$1" = "--args" ]; then ARGS="$2"; shift 2; fi
for rep in 1 2 3; do
for v in "$@"; do
if [ "$v" = product ]; then L=$PWD/openslam_g2o_amd/lib/libg2ohip.so; else L=$PWD/variants/$v/libg2ohip.so; fi
G2OHIP_LIB=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', round(d['value'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
done
