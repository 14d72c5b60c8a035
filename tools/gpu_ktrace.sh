#!/bin/bash
# kernel trace of a few bench steps (plain launches): per-kernel averages -> gpurun_out/ktrace_${TAG}.csv
TAG=${TAG:-x}
export TMPDIR=/tmp
R=$PWD
cd /tmp
rm -rf /tmp/pk_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/pk_$TAG -o k -- python $R/bench.py --no-cpu-baseline --graph off --steps 5 --warmup 2 "$@" > $R/gpurun_out/ktrace_$TAG.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/pk_$TAG -name "*.db" | head -1) $R/gpurun_out/ktrace_$TAG.csv
cd $R
python - <<EOP
import csv
for r in list(csv.reader(open("gpurun_out/ktrace_$TAG.csv")))[1:14]:
    print(r[0][:90].replace("void g2ohip::(anonymous namespace)::",""), r[1], r[3])
EOP
