#!/bin/bash
# round 6: stamps of one level-2 front + per-level own times (stamps variant), then the bench line of the product library
bash tools/gpu_tree_timeline.sh "" 2>&1 | grep "^launch  0\|level n  *[0-9]*  own\|slots"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()}, d['residual_rel'])"
