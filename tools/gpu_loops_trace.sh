#!/bin/bash
# kernel trace of one iteration of a loop-closure BA graph (tools/probe/loops_kernels.py): the launch sequence of the last solve
#   bash tools/gpu_loops_trace.sh [poses landmarks laps]   (HUBS, OPTS from the environment) -> gpurun_out/loopstrace/
R=$PWD; OUT=$R/gpurun_out/loopstrace; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/lt
rocprofv3 --kernel-trace --stats -d /tmp/lt -o k -- python $R/tools/probe/loops_kernels.py "$@" > $OUT/run.log 2>&1
DB=$(find /tmp/lt -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB $OUT/stats.csv
python - $DB > $OUT/seq.txt <<'EOP'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.display_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
names = [r[0] for r in rows]
idx = [i for i, n in enumerate(names) if "ba_assemble_poses" in n]
a, b = idx[-2], idx[-1]
t0 = rows[a][1]; prev = t0
for r in rows[a:b]:
    k = re.search(r'(\w+_kernel)', r[0])
    print("%8.1f us  +%6.1f gap  %7.1f us  grid %7d x %4d  %s" % ((r[1] - t0) / 1e3, (r[1] - prev) / 1e3, (r[2] - r[1]) / 1e3, r[3] // max(r[4], 1), r[4], k.group(1) if k else r[0][:60]))
    prev = r[2]
EOP
cd $R; head -25 $OUT/stats.csv | cut -c1-160; wc -l $OUT/seq.txt
