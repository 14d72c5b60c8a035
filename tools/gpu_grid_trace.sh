#!/bin/bash
# kernel trace of the grid workload: per-kernel totals of 3 steps -> gpurun_out/grid_trace.csv, one iteration launch by launch -> gpurun_out/grid_timeline.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/gt
rocprofv3 --kernel-trace --stats -d /tmp/gt -o k -- python $R/bench.py --workload grid --poses ${GRID_P:-10000} --no-cpu-baseline --graph off --steps 3 --warmup 1 "$@" > $R/gpurun_out/grid_trace.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/gt -name "*.db" | head -1) $R/gpurun_out/grid_trace.csv
python $R/tools/grid_timeline.py $(find /tmp/gt -name "*.db" | head -1) $R/gpurun_out/grid_timeline.txt
cd $R; python - <<EOP
import csv
for r in list(csv.reader(open("gpurun_out/grid_trace.csv")))[1:16]:
    print(r[0][:80].replace("void g2ohip::(anonymous namespace)::",""), "calls", r[1], "total ms %.2f" % (float(r[2])/1e6), "avg us %.1f" % (float(r[3])/1e3))
EOP
