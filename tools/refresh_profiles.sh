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
# strong-scaling emulation collated into one file
python - <<EOP
import json
o = {}
for n in (2, 4, 8):
    try:
        o[str(n)] = json.loads([l for l in open("$OUT/emu_%d.json" % n) if l.startswith("{")][-1])
    except Exception as e:
        o[str(n)] = {"error": str(e)}
json.dump(o, open("$OUT/${P}_rank_emulation.json", "w"), indent=1)
EOP
# pose graphs (the same solver without the Schur step)
python tools/posegraph_solve_time.py > $OUT/${P}_posegraph.txt 2> $OUT/pg.err
python tools/posegraph_lm_time.py > $OUT/${P}_posegraph_lm.jsonl 2>> $OUT/pg.err
# phase costs by switching parts of the two largest kernels off (timing only)
bash tools/gpu_schur_abl.sh > $OUT/${P}_schur_tile_ablation.txt 2>&1
bash tools/gpu_band_abl.sh > $OUT/${P}_band_ablation.txt 2>&1
# the k-block issue-rate probe the band kernel's design rests on
(cd tools/probe && hipcc --offload-arch=gfx950 -O3 -o /tmp/kblock_probe kblock_probe.hip 2>/dev/null && /tmp/kblock_probe) > $OUT/${P}_kblock_probe.txt 2>&1
# in-kernel stamps of the tree levels (stamps variant: make -C openslam_g2o_amd/csrc VARIANT=stamps EXTRA="-DG2OHIP_CHOL_STAMPS -DWSTAMP_BLOCK=2044")
if [ -f variants/stamps/libg2ohip.so ]; then bash tools/gpu_tree_timeline.sh "" > $OUT/${P}_tree_stamps.txt 2>&1; fi
ls -la $OUT
