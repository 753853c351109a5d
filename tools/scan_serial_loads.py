"""Loops that wait for every single load (developer aid): compiles each csrc/*.hip to gfx950 assembly and lists the
basic blocks that branch back to themselves, hold one or two global loads and an `s_waitcnt vmcnt(0)` — the shape
`for (...) acc += a[i]` compiles to: one round trip to memory per iteration, one behind the other.  Binary searches and
ragged remainders are expected entries; a fold of partial sums or a staging loop is not.

    python tools/scan_serial_loads.py [name filter]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as tmp:
    procs = []
    for src in sorted(glob.glob(os.path.join(ROOT, "optbayesexpt_amd", "csrc", "*.hip"))):
        out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
        procs.append((out, subprocess.Popen(
            ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
             "-I" + os.path.join(ROOT, "optbayesexpt_amd", "csrc"), "--cuda-device-only", "-S", src, "-o", out],
            stderr=subprocess.DEVNULL)))
    for out, proc in procs:
        if proc.wait() != 0:
            continue
        kern, cur, blocks = None, None, []
        for line in open(out):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                kern, cur = m.group(1), ["entry", []]
                blocks.append((kern, cur))
                continue
            m = re.match(r"^(\.LBB\d+_\d+):", line)
            if m and kern:
                cur = [m.group(1), []]
                blocks.append((kern, cur))
                continue
            t = line.strip()
            if cur is not None and t and not t.startswith((";", ".")):
                cur[1].append(t)
        seen = set()
        for kern, (label, ins) in blocks:
            loads = [i for i in ins if i.startswith(("global_load", "flat_load", "buffer_load"))]
            back = [i for i in ins if i.startswith("s_cbranch") and i.split()[-1] == label]
            wait0 = [i for i in ins if i.startswith("s_waitcnt") and "vmcnt(0)" in i]
            if back and wait0 and 1 <= len(loads) <= 2 and pat in kern and (kern, len(ins)) not in seen:
                seen.add((kern, len(ins)))
                print(f"{os.path.basename(out)[:-2]:13s} {label:12s} {len(loads)} load(s) in {len(ins):3d} instructions  {kern[7:120]}")
