"""prints the sequence of kernel durations (us) of a rocprofv3 trace between two launch indices of k_primary_ao. usage: trace_seq.py db first count"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
first, count = int(sys.argv[2]), int(sys.argv[3])
rows = db.execute("select name, start, end from kernels").fetchall()
try:
    rows += [("memcpy " + str(n), s, e) for n, s, e in db.execute("select name, start, end from memory_copies").fetchall()]
except Exception as ex:
    print("no memory copies:", ex)
rows.sort(key=lambda r: r[1])
idx = -1
prev_end = None
out = []
for n, s, e in rows:
    if "k_primary_ao" in n:
        idx += 1
    if first <= idx < first + count:
        tag = "T" if "k_primary_ao" in n else ("S" if "tile_order" in n else ("M" if n.startswith("memcpy") else ("C" if "copyBuffer" in n else "o")))
        out.append(f"{tag}{(e - s) / 1e3:.0f}" + (f"(+{(s - prev_end) / 1e3:.0f})" if prev_end else ""))
    prev_end = e
print(" ".join(out))
