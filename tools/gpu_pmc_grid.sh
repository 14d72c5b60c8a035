#!/bin/bash
# Counters of the scratch-slab kernels on the grid workload (10 000 cameras unless GRID_P says otherwise): HBM traffic (FETCH_SIZE / WRITE_SIZE,
# KiB; gfx950 tallies 64 B per 128 B request on streaming reads: MI355X_MICROARCH.md), matrix-core and busy cycles, each set in a pass of its own (FETCH_SIZE and WRITE_SIZE together abort rocprofv3).
#   bash tools/gpu_pmc_grid.sh  -> gpurun_out/pmc_grid.txt   (3 steps + 1 warm-up per pass: sums over 4 iterations incl. the warm-up)
set -u
R=$PWD; OUT=$R/gpurun_out/pmc_grid.txt; : > $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); rm -rf /tmp/pg$i
  timeout 900 rocprofv3 --kernel-trace --pmc $set -d /tmp/pg$i -o p -- python $R/bench.py --workload grid --poses ${GRID_P:-10000} --no-cpu-baseline --graph off --steps 3 --warmup 1 > $R/gpurun_out/pmc_grid_pass$i.log 2>&1
  for k in big_front_update_kernel big_extend_gather_kernel big_panel_solve_kernel big_forward_kernel big_backward_kernel big_fill_kernel; do
    python $R/tools/pmc_kernel.py $(find /tmp/pg$i -name "*.db" | head -1) $k >> $OUT 2>> $R/gpurun_out/pmc_grid.err
  done
done
cd $R; cat $OUT
