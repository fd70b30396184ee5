#!/usr/bin/env python3
"""Kernel-by-kernel timeline of the LAST optimize() in a rocprofv3 --kernel-trace rocpd database: start offset, duration and
the gap to the previous kernel's end (µs).  Usage: python tools/rocpd_timeline.py results.db [n_kernels]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "d.queue_id" if "queue_id" in cols else "0"
scol = "d.stream_id" if "stream_id" in cols else "0"
rows = c.execute(f"select s.kernel_name, d.start, d.end, {qcol}, {scol} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                 "order by d.start").fetchall()
rows = rows[-n:]
t0 = rows[0][1]
prev_end = None
tot_gap = 0.0
for name, st, en, qid, sid in rows:
    short = name.split("(")[0].replace("_Z", "")[:44]
    gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
    tot_gap += max(gap, 0.0)
    print(f"{(st - t0) / 1e3:9.2f} us  dur {(en - st) / 1e3:7.2f}  gap {gap:6.2f}  q{qid} s{sid}  {short}")
    prev_end = en
print(f"span {(rows[-1][2] - t0) / 1e3:.1f} us, sum of gaps {tot_gap:.1f} us")
