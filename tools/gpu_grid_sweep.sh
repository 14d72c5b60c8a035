#!/bin/bash
# option sweep on the grid workload (10 000 cameras): bash tools/gpu_grid_sweep.sh "opt=v[,opt=v]" ...
for o in "$@"; do
OPTS=""; if [ "$o" != base ]; then for kv in ${o//,/ }; do OPTS="$OPTS --opt $kv"; done; fi
timeout 300 python bench.py --workload grid --poses ${GRID_P:-10000} --steps 5 --warmup 2 --no-cpu-baseline $OPTS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']; st=d['solver_stats']
print('$o', 'step', round(d['value'],3), 'factor', round(k['chol_factor(all levels)']['avg_ms'],3), 'solve', round(k['chol_solve(all levels)']['avg_ms'],3), 'TF', round(d['roofline']['achieved'],2), 'nnz', st['choleskyNNZ'], 'levels', st['numLevels'], 'fronts', st['numFronts'], 'maxdim', st['maxFrontDim'], d['residual_rel'])"
done
