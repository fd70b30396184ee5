#!/usr/bin/env python3
"""Per (kernel, launch shape) averages of every counter in a rocprofv3 --pmc rocpd database.
Usage: pmc_counters.py <results.db> [kernel substring ...]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
q = (f"select s.kernel_name, d.grid_size_x / d.workgroup_size_x, d.grid_size_y / d.workgroup_size_y, i.name, count(*), avg(p.value) "
     f"from {pmc} p join {disp} d on p.event_id=d.event_id "
     f"join {sym} s on d.kernel_id=s.id join {info} i on p.pmc_id=i.id group by s.kernel_name, d.grid_size_x, d.grid_size_y, i.name")
flt = sys.argv[2:] or [""]
for name, gx, gy, ctr, n, avg in c.execute(q):
    name = name.split("(")[0]
    if any(f in name for f in flt):
        print(f"{name:36s} {f'{gx}x{gy}':>9s} {ctr:30s} launches {n:5d}  avg {avg:16.1f}")
