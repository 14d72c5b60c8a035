#!/bin/bash
# emulated rank of 8 / 4: options that lost at N = 1 where every SIMD is full
for n in 8 4; do for o in base tree_factor=1 tree_backward=2 tree_backward=0; do
  if [ "$o" = base ]; then OPT=""; else OPT="--opt $o"; fi
  python bench.py --emulate 0/$n --steps 20 --warmup 5 --no-cpu-baseline $OPT 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('N=$n $o:', round(d['value'],4), {kk:round(v['avg_ms'],4) for kk,v in k.items() if 'chol' in kk})"
done; done 2>&1 | tee gpurun_out/r5v_emulated_options.txt
