#!/bin/bash
# per-kernel durations of the factor slot with the band kernel (kernel trace, plain launches)
R=$PWD; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for nw in ${NWS:-1}; do
  rm -rf /tmp/bp$nw
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/bp$nw -o p -- python $R/bench.py --no-cpu-baseline --graph off --steps 10 --warmup 2 --opt band_kernel=$nw ${BENCH_ARGS:-} > $R/gpurun_out/bandprof_$nw.log 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/bp$nw -name "*.db" | head -1) $R/gpurun_out/bandprof_$nw.csv
  echo "== band_waves=$nw"; head -8 $R/gpurun_out/bandprof_$nw.csv | cut -c1-60,200-
  python - <<EOP
import csv
for r in csv.reader(open("$R/gpurun_out/bandprof_$nw.csv")):
    if "band_chain" in r[0] or "wave_front" in r[0] or "schur_tile" in r[0]: print(r[0][:70], r[1:])
EOP
done
