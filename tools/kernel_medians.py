"""Median duration of every kernel in a rocprofv3 kernel trace (developer aid):
    python tools/kernel_medians.py <results.db> [substring ...]"""
import sqlite3
import sys
from collections import defaultdict

import numpy as np

c = sqlite3.connect(sys.argv[1])
want = sys.argv[2:]
d = defaultdict(list)
for name, start, end in c.execute("select name, start, end from kernels order by start"):
    short = name.split("(")[0].replace("void ", "")[:70]
    if not want or any(w in short for w in want):
        d[short].append((end - start) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -np.sum(kv[1])):
    print(f"{k:70s} n={len(v):5d}  median {np.median(v):9.2f} us  mean {np.mean(v):9.2f}  min {np.min(v):9.2f}")
