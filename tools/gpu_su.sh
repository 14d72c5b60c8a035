#!/bin/bash
# pivot blocks + panel rows in one launch (big_panel_solve_kernel) by workgroup limit: grid graph, A/B in one call
for c in 0 256 384 512 768; do
  export G2OHIP_PS_MAX=$c
  echo "== ps_max $c"
  bash tools/gpu_grid_sweep.sh base
  GRID_P=50000 bash tools/gpu_grid_sweep.sh base
done
