#!/usr/bin/env python3
"""Turns a rocprofv3 (rocpd sqlite) result into the per-kernel summary committed under profiles/.
usage: summarize_rocprof.py <results.db> [--json out.json]"""
import json
import sqlite3
import sys


def summarize(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = []
    for r in rows:
        out.append({"kernel": r[0], "calls": r[1], "total_ms": r[2] / 1e6, "avg_us": r[3] / 1e3, "min_us": r[4] / 1e3,
                    "max_us": r[5] / 1e3, "pct": 100.0 * r[2] / total, "vgpr": r[6], "sgpr": r[7], "lds_bytes": r[8],
                    "grid": r[9], "block": r[10]})
    return out


if __name__ == "__main__":
    rows = summarize(sys.argv[1])
    print(f"{'kernel':<72}{'calls':>7}{'avg_us':>11}{'min_us':>11}{'max_us':>11}{'pct':>8}{'vgpr':>6}{'sgpr':>6}{'lds':>8}{'grid':>8}{'blk':>5}")
    for r in rows:
        print(f"{r['kernel'][:70]:<72}{r['calls']:>7}{r['avg_us']:>11.2f}{r['min_us']:>11.2f}{r['max_us']:>11.2f}{r['pct']:>8.2f}"
              f"{r['vgpr']:>6}{r['sgpr']:>6}{r['lds_bytes']:>8}{r['grid']:>8}{r['block']:>5}")
    if "--json" in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
