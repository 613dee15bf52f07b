#!/usr/bin/env python3
"""profiles/pmc_k_tile.json from the rocprofv3 --pmc passes of THIS build (tools/gpu_profile_round.sh runs them on the GPU box:
FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only beside --pmc).  Per workload: the Jacobian kernel's average
counters per launch, traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE doubled: MI355X_MICROARCH.md §HBM, gfx950
reports half of a wide coalesced read; the counters are in KB), the algorithmic bytes, and `kernel_sources_sha16` — bench.py
refuses the file when that differs from the sources it runs.

    python tools/pmc_to_json.py OUT.json  NAME:EDGES:M:FETCH.db:WRITE.db [...]
"""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from batrack_amd import _lib  # noqa: E402



def kernel_avg(db, counter):
    """(kernel name, launches, average value per launch) of the Jacobian kernel with the most launches in this pass."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    # the step's kernels by their template names (a planner kernel like k_edge_tables is none of them); mode 2 of k_tile / k_etile /
    # k_stream / k_edge2u is the step's last kernel, k_edge2's first template argument is a tile count
    rows = [r for r in rows if re.search(r"\bk_(tile|etile|stream|edge|edge2)<", r[0]) and "upd" not in r[0].lower()
            and not re.search(r"\bk_(tile|etile|stream|edge)<2", r[0])]
    if not rows:
        raise SystemExit(f"{db}: no Jacobian kernel with counter {counter}")
    # the pose+structure instantiation: the one with the largest average (the structure-only / update modes move less)
    r = max(rows, key=lambda x: x[2])
    return r[0], int(r[1]), float(r[2])


def main():
    out_path, specs = sys.argv[1], sys.argv[2:]
    out = {"_comment": "Jacobian-kernel HBM traffic per launch from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the build "
                       "whose sources hash to kernel_sources_sha16; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, FETCH doubled per "
                       "MI355X_MICROARCH.md §HBM; algorithmic bytes = 40 E + 20 m + 72 N (SURVEY.md §8d)",
           "kernel_sources_sha16": _lib.kernel_sources_sha16()}
    for spec in specs:
        name, E, m, n_all, fdb, wdb = spec.split(":")
        E, m, n_all = int(E), int(m), int(n_all)
        kf, nf, fetch = kernel_avg(fdb, "FETCH_SIZE")
        kw, nw, write = kernel_avg(wdb, "WRITE_SIZE")
        alg = 40 * E + 20 * m + 72 * n_all
        traffic = int(round((2.0 * fetch + write) * 1024.0))
        out[name] = {"edges": E, "kernel": kf[:80], "launches": [nf, nw], "FETCH_SIZE_KB": round(fetch, 2), "WRITE_SIZE_KB": round(write, 2),
                     "traffic_bytes": traffic, "algorithmic_bytes": alg, "traffic_over_algorithmic": round(traffic / alg, 3)}
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
