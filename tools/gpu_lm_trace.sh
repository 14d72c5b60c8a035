R=$PWD; OUT=$R/gpurun_out/lmtrace; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/lmt
rocprofv3 --kernel-trace --stats -d /tmp/lmt -o k -- python $R/lm_bench.py > $OUT/run.log 2>&1
DB=$(find /tmp/lmt -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB $OUT/stats.csv
python - $DB > $OUT/seq.txt <<'EOP'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute("select s.display_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
n=len(rows)
# last ~120 dispatches
t0=rows[max(0,n-140)][1]; prev=t0
import re
for r in rows[max(0,n-140):]:
    k=re.search(r'(\w+_kernel|\w+Buffer\w*)', r[0])
    print("%9.1f +%7.1f %8.1f grid %7d  %s" % ((r[1]-t0)/1e3, (r[1]-prev)/1e3, (r[2]-r[1])/1e3, r[3]//max(r[4],1), k.group(1) if k else r[0][:40]))
    prev=r[2]
EOP
