#!/bin/bash
# grouped trailing updates of long in-place chains (options big_group / big_group_merge_tiles: analysis time, G2OHIP_OPTIONS): hub graphs at scale
for o in big_group=1 big_group=8,big_group_merge_tiles=0 big_group=8 big_group=8,big_group_merge_tiles=8192 big_group=16; do
  echo "$o"
  HUBS=1 G2OHIP_OPTIONS="$o" python tools/probe/loops_scale.py 4500 20000 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(' 9k front', round(d['ms_per_iteration'],2), d['residual_rel'], d['maxFrontDim'])"
  HUBS=1 G2OHIP_OPTIONS="$o" python tools/probe/loops_scale.py 10000 60000 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(' 20k front', round(d['ms_per_iteration'],2), d['residual_rel'], d['maxFrontDim'])"
done
