#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE rocpd databases into HBM bytes per launch, per (kernel, launch shape).

Correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: the counters are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads, so the read side is doubled; WRITE_SIZE is taken as is (uncalibrated).
For every kernel symbol the entry under "kernels" is its MOST FREQUENT launch shape in the run (the shape optimize()
replays); all shapes are listed under "by_shape".
Usage: pmc_summary.py <fetch.db> <write.db> <label> [out.json] [points_per_frame]   (the last one lets bench.py check that a
record belongs to the workload it is timing)"""
import json
import sqlite3
import sys


def per_kernel(path):
    c = sqlite3.connect(path)
    q = ("select s.kernel_name, d.grid_size_x / d.workgroup_size_x, d.grid_size_y / d.workgroup_size_y, count(*), avg(p.value) "
         "from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id=d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x, d.grid_size_y")
    return {(r[0].split("(")[0], f"{r[1]}x{r[2]}"): (r[3], r[4]) for r in c.execute(q)}


def main():
    fetch, write, label = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), sys.argv[3]
    out = {"label": label, "unit": "bytes per launch", "correction": "FETCH_SIZE KiB x2 (gfx950 half-count), WRITE_SIZE KiB x1",
           "kernels": {}, "by_shape": {}}
    if len(sys.argv) > 5:
        out["points_per_frame"] = float(sys.argv[5])
    best = {}
    for key in sorted(set(fetch) | set(write)):
        k, shape = key
        fr = fetch.get(key, (0, 0.0)); wr = write.get(key, (0, 0.0))
        rd = fr[1] * 1024 * 2; wb = wr[1] * 1024
        rec = {"workgroups": shape, "launches": fr[0] or wr[0], "read_bytes": round(rd), "write_bytes": round(wb), "hbm_bytes": round(rd + wb)}
        out["by_shape"][f"{k} [{shape}]"] = rec
        if k not in best or rec["launches"] > best[k]["launches"]:
            best[k] = rec
    out["kernels"] = best
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
