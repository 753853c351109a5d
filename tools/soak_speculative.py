"""Long runs with and without the sweep enqueued behind the update, compared bit for bit (developer aid, GPU):

    python tools/soak_speculative.py [minutes=3] [seed=0] [max_cycles=0 (no limit)]

Random cloud / grid sizes, both classes, thresholds that resample rarely or often, random things done between
pdf_update() and the next opt_setting().  Uses the helpers of tests/test_gpu_speculative.py."""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import optbayesexpt_amd as obe            # noqa: E402
import test_gpu_speculative as t          # noqa: E402

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = np.random.default_rng(seed)
warnings.simplefilter("ignore")
t_end = time.time() + 60 * minutes
runs = cycles = taken = 0
while time.time() < t_end and not (max_cycles and cycles >= max_cycles):
    n = int(g.choice([3000, 20000, 70000, 300000]))
    ns = int(g.choice([300, 1500, 2048, 5000]))
    noise = bool(g.integers(2))
    thr = float(g.choice([0.1, 0.5, 0.9]))
    n_cyc = int(g.integers(40, 160))
    hooks = g.integers(0, 6, n_cyc)

    def between(o, c):
        k = hooks[c]
        if k == 0:
            return o.covariance().ravel()
        if k == 1:
            return o.std()
        if k == 2 and c % 7 == 0:
            return np.asarray(o.opt_setting())
        if k == 3 and c % 11 == 0:
            w = o.particle_weights.copy()
            w[::3] *= 0.25
            o.particle_weights = w / w.sum()
            return o.mean()
        return np.zeros(1)

    kw = dict(n_particles=n, n_settings=ns, noise_param=noise, threshold=thr, seed=int(g.integers(1 << 30)))
    plain = t.cycles(t.make(obe, False, **kw), n_cyc, between, seed=seed + runs)
    for mode in (True, "auto"):
        o = t.make(obe, mode, **kw)
        cnt = t.counted(o)
        t.same(t.cycles(o, n_cyc, between, seed=seed + runs), [dict(e, sweep=dict(e["sweep"])) for e in plain])
        taken += cnt["taken"]
    runs += 1
    cycles += 3 * n_cyc
print(f"soak: {runs} experiments, {cycles} cycles, {taken} speculative sweeps used, all equal")
