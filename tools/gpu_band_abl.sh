#!/bin/bash
# timing ablations of the band kernel (wrong results): which part of a chain costs what
R=$PWD; export TMPDIR=/tmp; cd /tmp
for abl in ${ABLS:-0 1 2 3 7}; do
  rm -rf /tmp/ba$abl
  G2OHIP_BAND_ABL=$abl timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ba$abl -o p -- python $R/bench.py --no-cpu-baseline --graph off --steps 10 --warmup 2 ${BENCH_ARGS:-} > $R/gpurun_out/bandabl_$abl.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/ba$abl -name "*.db" | head -1) /tmp/ba$abl.csv
  python - <<EOP
import csv
for r in csv.reader(open("/tmp/ba$abl.csv")):
    if "band_" in r[0]: print("abl $abl", r[0][40:75], "avg us %.1f" % (float(r[3]) / 1e3))
EOP
done
