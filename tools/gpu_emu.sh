#!/bin/bash
# emulated rank of an N-rank job (collectives skipped): bash tools/gpu_emu.sh [lib ...]   ("product" = the in-tree library)
for rep in 1 2; do
for v in "${@:-product}"; do
if [ "$v" = product ]; then L=$PWD/openslam_g2o_amd/lib/libg2ohip.so; else L=$PWD/variants/$v/libg2ohip.so; fi
for n in 8 2; do
G2OHIP_LIB=$L python bench.py --emulate 0/$n --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v N=$n', round(d['value'],4), round(sum(v['avg_ms']*v['launches_per_step'] for v in d['kernels'].values()),4))"
done
done
done
