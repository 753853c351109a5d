"""FP64 VALU instructions in the hottest loop of each sweep kernel (developer aid).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ioptbayesexpt_amd/csrc \
          --cuda-device-only -S optbayesexpt_amd/csrc/obe_sweep.hip -o /tmp/sweep.s
    python tools/count_isa.py /tmp/sweep.s 'LorentzILi1EEELi8ELb0' 16

prints the instruction mix of the basic block with the most FP64 instructions and the issue
slots per evaluation (v_rcp_f64 counted as 4 slots: quarter rate), given the evaluations one
trip of that loop performs (2 particles x 8 settings = 16 for the pair loop)."""
import re
import sys
from collections import Counter

path, pattern, evals = sys.argv[1], sys.argv[2], int(sys.argv[3])
lines = open(path).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*sweep_kernel\w*:", l)]
for k, s in enumerate(starts):
    if pattern not in lines[s]:
        continue
    e = starts[k + 1] if k + 1 < len(starts) else len(lines)
    blocks, name, cur = [], "entry", []
    for l in lines[s:e]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
        else:
            t = l.strip()
            if t and not t.startswith((";", ".")):
                cur.append(t.split()[0])
    blocks.append((name, cur))
    f64 = lambda b: sum(1 for op in b if re.match(r"v_\w+_f64", op))
    print(lines[s].split(":")[0])
    # the loops: blocks that end in a conditional branch (pair loop, then the 2x-unrolled remainder loop)
    loops = [(n, b) for n, b in blocks if n != "entry" and any(op.startswith("s_cbranch") for op in b)]
    for name, top in sorted(loops, key=lambda nb: -f64(nb[1]))[:2]:
        c = Counter(top)
        n_rcp = sum(v for op, v in c.items() if op.startswith("v_rcp_f64"))
        n_f64 = f64(top)
        slots = (n_f64 - n_rcp) + 4 * n_rcp
        print(f"  block {name}: {n_f64} FP64 VALU ({n_rcp} v_rcp_f64), {len(top)} instructions in all")
        print(f"    per evaluation ({evals} per trip): {(n_f64 - n_rcp) / evals:.3f} + {n_rcp / evals:.4f} rcp"
              f" = {slots / evals:.3f} issue slots")
        print("    " + ", ".join(f"{op} {v}" for op, v in c.most_common(8)))
