#!/bin/bash
# A/B of solver options on the GPU box: bash tools/gpu_ab.sh opt=value [opt=value ...]   ("base" = no option)
for o in "$@"; do
if [ "$o" = base ]; then OPT=""; else OPT="--opt $o"; fi
python bench.py --steps 20 --warmup 5 --no-cpu-baseline $OPT 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('$o:', round(d['value'],4), 'factor', round(k['chol_factor(all levels)']['avg_ms'],4), 'tiles', round(k['schur_tiles']['avg_ms'],4), 'solve', round(k['chol_solve(all levels)']['avg_ms'],4), 'backsub', round(k['back_substitute']['avg_ms'],4), d['residual_rel'], d['solver_stats']['numLevels'], d['solver_stats']['choleskyNNZ'])"
done
