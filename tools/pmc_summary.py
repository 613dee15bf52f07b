#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 results .db."""
import sqlite3, sys
for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print(f"# {path}")
    q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
         "group by kernel_name, counter_name order by kernel_name, counter_name")
    try:
        rows = cur.execute(q).fetchall()
    except Exception as e:
        print("columns:", cols, "error:", e); continue
    for r in rows:
        print(f"{r[0][:60]:60s} {r[1]:14s} n={r[2]:4d} avg={r[3]:14.2f} min={r[4]:14.2f} max={r[5]:14.2f}")
