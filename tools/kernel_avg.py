"""Average duration of the kernels whose name contains a pattern, last N launches, from a rocprofv3 kernel trace
(developer aid):  python tools/kernel_avg.py <results.db> <pattern> [last=20]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = [(r[0], (r[2] - r[1]) / 1e3) for r in c.execute("select name, start, end from kernels order by start") if pat in r[0]]
rows = rows[-last:]
if rows:
    d = sorted(x[1] for x in rows)
    print(f"{pat}: {len(d)} launches, mean {sum(d) / len(d):.2f} us, median {d[len(d) // 2]:.2f} us, min {d[0]:.2f} us   {rows[-1][0][:70]}")
else:
    print(f"{pat}: no launches")
