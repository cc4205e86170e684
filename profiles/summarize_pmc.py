#!/usr/bin/env python3
"""Per-kernel means of the counters in one or more rocprofv3 --pmc result DBs.
usage: summarize_pmc.py a_results.db [b_results.db ...]"""
import sqlite3
import sys
from collections import defaultdict


def load(path):
    db = sqlite3.connect(path)
    cols = [d[1] for d in db.execute('pragma table_info("counters_collection")')]
    rows = db.execute("select * from counters_collection").fetchall()
    out = defaultdict(lambda: defaultdict(list))
    ki, ci, vi = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"), cols.index("counter_name"), cols.index("value")
    di = cols.index("dispatch_id")
    acc = defaultdict(float)
    for r in rows:
        acc[(r[ki], r[ci], r[di])] += r[vi]
    for (k, c, _), v in acc.items():
        out[k][c].append(v)
    return out


if __name__ == "__main__":
    for p in sys.argv[1:]:
        for k, cs in load(p).items():
            if "dust::" not in k:
                continue
            for c, vals in sorted(cs.items()):
                print(f"{k[:48]:<50}{c:<26}n={len(vals):<4} mean={sum(vals)/len(vals):>16.1f} min={min(vals):>16.1f} max={max(vals):>16.1f}")
