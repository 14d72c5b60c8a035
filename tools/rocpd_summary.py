#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration, share) from a rocprofv3 rocpd
SQLite database (`rocprofv3 --kernel-trace --stats ...` writes <name>_results.db).
Usage: python tools/rocpd_summary.py results.db [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    q = ("select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, kd, ks, name_col))
    rows = list(db.execute(q))
    total = float(sum(r[2] for r in rows)) or 1.0
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, c, t, mn, mx in rows:
        out.writerow([n, c, t, "%.1f" % (t / c), "%.2f" % (100.0 * t / total), mn, mx])
    assert "start" in cols and "end" in cols


if __name__ == "__main__":
    main()
