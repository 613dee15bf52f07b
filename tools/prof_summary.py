#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (kernel-trace) into per-kernel stats text.
usage: python tools/prof_summary.py <results.db> [more.db ...]"""
import sqlite3
import sys

for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                       "from kernels group by name order by 6 desc").fetchall()
    total = sum(r[5] for r in rows)
    print(f"# {path}")
    print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%time':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'grid':>8s} {'wg':>5s}")
    for r in rows:
        print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]/1e3:9.2f} {r[3]/1e3:9.2f} {r[4]/1e3:9.2f} {100*r[5]/total:6.1f} {r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d} {r[10]:8d} {r[11]:5d}")
