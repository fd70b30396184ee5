#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd database.  Usage: pmc_counters.py <results.db> [kernel substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
q = (f"select s.kernel_name, i.name, count(*), avg(p.value) from {pmc} p join {disp} d on p.event_id=d.event_id "
     f"join {sym} s on d.kernel_id=s.id join {info} i on p.pmc_id=i.id group by s.kernel_name, i.name")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for name, ctr, n, avg in c.execute(q):
    name = name.split("(")[0]
    if flt in name:
        print(f"{name:40s} {ctr:28s} launches {n:5d}  avg {avg:16.1f}")
