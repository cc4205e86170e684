"""per-dispatch view of a rocprofv3 kernel trace: mean duration of k_primary_ao and k_tile_order and the gaps between consecutive
dispatches, over windows of N dispatches in launch order. usage: trace_segments.py results.db [window]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows = db.execute("select name, start, end from kernels order by start").fetchall()
pao = [(s, e) for n, s, e in rows if "k_primary_ao<0>" in n]
allk = [(n, s, e) for n, s, e in rows]
print("dispatches", len(rows), "k_primary_ao<0>", len(pao))
for i in range(0, len(pao) - win + 1, win):
    seg = pao[i:i + win]
    dur = sum(e - s for s, e in seg) / win / 1e3
    span = (seg[-1][1] - seg[0][0]) / win / 1e3
    others = [(n, s, e) for n, s, e in allk if seg[0][0] <= s <= seg[-1][1] and "k_primary_ao" not in n]
    od = sum(e - s for n, s, e in others) / win / 1e3
    names = {}
    for n, s, e in others:
        k = n.split("(")[0][-28:]
        names[k] = names.get(k, 0) + 1
    print(f"  launches {i:5d}..{i + win:5d}: trace {dur:7.1f} us, step (start to start) {span:7.1f} us, other kernels {od:5.1f} us/step {names}")
