#!/bin/bash
for nl in 16 12; do
G2OHIP_PLAN_DUMP=1 python bench.py --emulate 0/8 --steps 3 --warmup 2 --no-cpu-baseline --nd-leaf $nl 2>&1 | grep -i "band chains\|tree factor\|tree backward\|rejected" | head -8
done | tee gpurun_out/r5w_plan.txt
