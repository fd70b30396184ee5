#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per (kernel, launch shape) count / total / mean / min / max duration (µs).
Launches of one symbol with different grids are different rows (a frame batch runs as frame groups; a row must describe
ONE launch shape).  Usage: python tools/rocpd_stats.py results.db [> profiles/summary.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select s.kernel_name, d.grid_size_x / d.workgroup_size_x, d.grid_size_y / d.workgroup_size_y, d.workgroup_size_x, "
        "count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    span = c.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    print(f"# source: {path}")
    print(f"# total kernel time {tot/1e3:.1f} us over a dispatch span of {(span[1]-span[0])/1e3:.1f} us; one row per (kernel, grid)")
    print(f"{'kernel':52s} {'workgroups':>12s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'share':>7s}")
    for name, gx, gy, wg, n, t, a, mn, mx in rows:
        short = name.split("(")[0][:52]
        print(f"{short:52s} {f'{gx}x{gy}x{wg}t':>12s} {n:7d} {t/1e3:12.1f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*t/tot:6.1f}%")


if __name__ == "__main__":
    main(sys.argv[1])
