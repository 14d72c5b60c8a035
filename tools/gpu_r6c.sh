#!/bin/bash
# A/B of stamps variants inside one call: bash tools/gpu_r6c.sh variantA variantB ...   (each: variants/<name>/libg2ohip.so built with -DG2OHIP_CHOL_STAMPS)
for rep in 1 2; do
for v in "$@"; do
G2OHIP_LIB=$PWD/variants/$v/libg2ohip.so G2OHIP_CHOL_STAMPS_PRINT=1 G2OHIP_CHOL_TIMELINE=$PWD/gpurun_out/timeline_$v.txt python bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph off 2> gpurun_out/stamps_err_$v.txt > /dev/null
echo "== $v: $(grep '^launch  0' gpurun_out/stamps_err_$v.txt | tail -1)"
python tools/tree_handoff.py gpurun_out/timeline_$v.txt | sed -n 4,6p
G2OHIP_LIB=$PWD/variants/$v/libg2ohip.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items() if 'chol' in k})"
done
done
