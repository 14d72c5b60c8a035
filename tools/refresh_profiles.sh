#!/bin/bash
# Regenerates what profiles/ holds (run on the GPU box from the repo root; results under gpurun_out/refresh/).
#   bash tools/refresh_profiles.sh r2        (round prefix of the file names)
set -u
P=${1:-r3}
OUT=$PWD/gpurun_out/refresh
mkdir -p $OUT
python bench.py --steps 20 --warmup 3 > $OUT/${P}_bench.json 2> $OUT/bench.err
python lm_bench.py > $OUT/${P}_lm_config5.json 2> $OUT/lm.err
for n in 2 4 8; do python bench.py --emulate 0/$n --no-cpu-baseline --steps 20 --warmup 3 > $OUT/emu_$n.json 2>> $OUT/bench.err; done
export TMPDIR=/tmp
R=$PWD
cd /tmp
rm -rf /tmp/pk /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --stats -d /tmp/pk -o k -- python $R/bench.py --no-cpu-baseline --graph off --steps 5 --warmup 2 > $OUT/trace.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/pk -name "*.db" | head -1) $OUT/${P}_kernel_stats.csv
python $R/tools/level_times.py $(find /tmp/pk -name "*.db" | head -1) 2 > $OUT/${P}_level_times.txt
cd $R
bash tools/gpu_pmc_traffic.sh $P > $OUT/pmc_traffic.log 2>&1
# instruction / stall counters of the three largest kernels (summaries: $OUT/${P}_pmc_<kernel>.txt)
for k in ba_schur_tile_kernel band_wave_kernel wave_front_kernel; do TAG=_$k bash tools/gpu_pmc_wave.sh $k > /dev/null 2>&1; cp gpurun_out/pmcw_$k/summary.txt $OUT/${P}_pmc_$k.txt; done
ls -la $OUT
