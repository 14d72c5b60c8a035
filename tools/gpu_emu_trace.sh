# kernel timeline of one rank of an N-rank job run alone (bench.py --emulate 0/N): where the per-rank time goes
N=${1:-8}
R=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/et
rocprofv3 --kernel-trace -d /tmp/et -o t -- python $R/bench.py --emulate 0/$N --no-cpu-baseline --steps 4 --warmup 3 > $R/gpurun_out/emu_trace.log 2>&1
python - <<EOP
import sqlite3, glob
db = glob.glob("/tmp/et/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select d.start, d.end, s.kernel_name, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# the last iteration: from the last ba_assemble_poses / first kernel of a step to the end
names = [r[2] for r in rows]
last = max(i for i, n in enumerate(names) if "ba_assemble_poses" in n or "assemble_vertex" in n)
prev = max(i for i, n in enumerate(names[:last]) if "ba_assemble_poses" in n or "assemble_vertex" in n)
t0 = rows[prev][0]
pe = t0
for s, e, n, g, w in rows[prev:last]:
    print("%8.1f  gap %6.1f  dur %7.1f  %s  grid %d" % ((s - t0) / 1e3, (s - pe) / 1e3, (e - s) / 1e3, n[:70], g // max(w, 1)))
    pe = e
print("step total us", (rows[last][0] - t0) / 1e3)
EOP
