#!/usr/bin/env python
"""Sum of one rocprofv3 PMC counter per kernel name from a rocpd DB: python tools/pmc_kernel.py results.db [substr]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
rows = db.execute("select s.display_name, i.name, sum(p.value), count(*) from %s d join %s s on d.kernel_id=s.id join %s p on p.event_id=d.id "
                  "join %s i on p.pmc_id=i.id group by s.display_name, i.name order by 3 desc" % (kd, ks, pe, pi))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for n, c, v, k in rows:
    if sub in n:
        print("%-14s %16.0f  calls %4d  %s" % (c, v, k, n[:90]))
