#!/usr/bin/env python
"""Launch-by-launch timeline of ONE iteration out of a rocprofv3 kernel trace (rocpd database): the launches between the last two
Schur tile kernels, in start order -- offset, duration, gap to the previous end on any stream, short name, workgroups.
Usage: python tools/grid_timeline.py results.db [out.txt]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
dcols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
gcol = "grid_size_x" if "grid_size_x" in dcols else ("grid_x" if "grid_x" in dcols else None)
wcol = "workgroup_size_x" if "workgroup_size_x" in dcols else None
sel = "d.start, d.end, s.%s" % name_col + (", d.%s" % gcol if gcol else ", 0") + (", d.%s" % wcol if wcol else ", 1")
rows = sorted(db.execute("select %s from %s d join %s s on d.kernel_id = s.id" % (sel, kd, ks)))
marks = [i for i, r in enumerate(rows) if "schur_tile_kernel" in r[2]]
a, b = marks[-2], marks[-1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
t0, prev_end = rows[a][0], rows[a][0]
for st, en, nm, g, w in rows[a:b]:
    nm = re.sub(r"\(.*", "", nm.replace("void ", "").replace("g2ohip::(anonymous namespace)::", "").replace("g2ohip::", ""))
    out.write("%9.1f us  dur %7.1f  gap %6.1f  %-44s wgs %d\n" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, nm[:44], (g // max(w, 1)) if w else g))
    prev_end = max(prev_end, en)
