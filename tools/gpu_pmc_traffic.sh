#!/bin/bash
# HBM traffic + matrix-core counters per kernel slot (three PMC passes) -> gpurun_out/refresh/<prefix>_pmc_traffic.json
P=${1:-r3}; R=$PWD; OUT=$R/gpurun_out/refresh; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o f -- python $R/bench.py --no-cpu-baseline --graph off --steps 2 --warmup 1 > $OUT/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o w -- python $R/bench.py --no-cpu-baseline --graph off --steps 2 --warmup 1 > $OUT/pmc_w.log 2>&1
rm -rf /tmp/pm
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/pm -o m -- python $R/bench.py --no-cpu-baseline --graph off --steps 2 --warmup 1 > $OUT/pmc_m.log 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) $(find /tmp/pm -name "*.db" | head -1) > $OUT/${P}_pmc_traffic.json
cat $OUT/${P}_pmc_traffic.json | python -c "
import json,sys
t=json.load(sys.stdin)
for k,v in t.items():
    if not k.startswith('_'): print(k, round(v['hbm_bytes_per_step']/1e9,3), v['launches_per_step'])"
