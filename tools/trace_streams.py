"""Kernels between two sweeps of a rocprofv3 kernel trace, with their streams — overlapping chains show as
interleaved rows (developer aid):

    rocprofv3 --kernel-trace -d out -o t -- python3 tools/shard_cycle.py c5 8 10
    python tools/trace_streams.py out/.../t_results.db [how many cycles = 2] [must contain = resample_kernel]
"""
import sqlite3
import sys

db = sys.argv[1]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 2
needle = sys.argv[3] if len(sys.argv) > 3 else "resample_kernel"
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "sweep_kernel" in r[0]]
shown = 0
for k in range(len(idx) - 1, 0, -1):
    a, b = idx[k - 1], idx[k]
    if not any(needle in r[0] for r in rows[a:b]):
        continue
    t0 = rows[a][2]
    last_end = {}
    for r in rows[a:b + 1]:
        name = r[0].split("(")[0].replace("void ", "")[:56]
        idle = (r[1] - max(last_end.values())) / 1e3 if last_end else 0.0
        print(f"{(r[1] - t0) / 1e3:9.1f} -> {(r[2] - t0) / 1e3:9.1f} us  dur {(r[2] - r[1]) / 1e3:7.1f}  "
              f"stream {r[3]}  {'idle before %5.1f' % idle if idle > 0 else ' ' * 17}  {name}")
        last_end[r[3]] = r[2]
    print()
    shown += 1
    if shown == want:
        break
