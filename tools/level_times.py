#!/usr/bin/env python
"""Per-launch durations of the multifrontal kernels in the last bench iteration of a rocpd DB."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute("select s.display_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.group_segment_size from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
nlev = int(sys.argv[2])
for key in ("band_wave", "wave_front", "front_factor", "front_forward", "front_backward"):
    sel = [r for r in rows if key in r[0]][-nlev:]
    if not sel:
        continue
    print(key, "total %.0f us" % (sum(r[2] - r[1] for r in sel) / 1e3))
    print("   grid:ldsB:us ", " ".join("%d:%d:%.0f" % (r[3] // r[4], r[5], (r[2] - r[1]) / 1e3) for r in sel))
