import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
print(" ".join(f"{(e - s) / 1e3:.1f}" for n, s, e in db.execute("select name, start, end from kernels order by start") if sys.argv[2] in n))
