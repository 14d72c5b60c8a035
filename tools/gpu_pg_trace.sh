#!/bin/bash
# kernel trace of the pose-graph solves (sphere fixture): per-kernel totals and the launch sequence of one solve
#   bash tools/gpu_pg_trace.sh [graph] [opt=value ...]  -> gpurun_out/pgtrace/
R=$PWD; OUT=$R/gpurun_out/pgtrace; mkdir -p $OUT
G=${1:-sphere}; shift
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pg
PG_GRAPHS=$G rocprofv3 --kernel-trace --stats -d /tmp/pg -o k -- python $R/tools/posegraph_solve_time.py use_graph=0 "$@" > $OUT/run.log 2>&1
DB=$(find /tmp/pg -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB $OUT/stats.csv
python - $DB > $OUT/seq.txt <<'EOP'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.display_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
# the last solve: from the last 'lambda_kernel' before the end
names = [r[0] for r in rows]
idx = [i for i, n in enumerate(names) if "lambda_kernel" in n]
a = idx[-3] if len(idx) >= 3 else 0
b = idx[-1]
t0 = rows[a][1]
prev = t0
for r in rows[a:b]:
    print("%8.1f us  +%6.1f gap  %7.1f us  grid %7d x %4d  %s" % ((r[1] - t0) / 1e3, (r[1] - prev) / 1e3, (r[2] - r[1]) / 1e3, r[3] // max(r[4], 1), r[4], r[0][:70]))
    prev = r[2]
EOP
cd $R; head -30 $OUT/stats.csv; wc -l $OUT/seq.txt
