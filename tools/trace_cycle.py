"""Timeline of the kernels between two sweep-kernel launches, from a rocprofv3 kernel trace
(developer aid):  python tools/trace_cycle.py <bench_results.db> [which sweep]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "sweep_kernel" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
if k < 0:
    k += len(idx) - 1          # counted from the last pair of sweeps
a, b = idx[k], idx[k + 1]
t0 = rows[a][1]
prev_end = None
busy = gaps = 0.0
for r in rows[a:b + 1]:
    nm = r[0].split("(")[0].replace("void ", "")[:64]
    gap = (r[1] - prev_end) / 1e3 if prev_end else 0.0
    dur = (r[2] - r[1]) / 1e3
    if r is not rows[a] and r is not rows[b]:
        busy += dur
    if prev_end:
        gaps += max(gap, 0.0)
    print(f"{(r[1] - t0) / 1e3:10.1f} us  dur {dur:9.1f} us  gap_before {gap:7.1f} us  {nm}")
    prev_end = r[2]
print(f"between the two sweeps: {busy:.1f} us of kernels, {gaps:.1f} us idle")
