"""The resample on one stream and on two (OBE_RESAMPLE_STREAMS=1 / 2) must be the same resample (developer aid, GPU):

    OBE_RESAMPLE_STREAMS=1 python tools/check_resample_streams.py out1.npz [cycles=120]
    OBE_RESAMPLE_STREAMS=2 python tools/check_resample_streams.py out2.npz
    python tools/check_resample_streams.py out1.npz out2.npz --compare

Cycles of a large cloud (the generator's chain is long enough for the side stream) that resample almost every
time; per cycle: chosen setting, resample flag, generator state, and digests of weights, particles and indices."""
import hashlib
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if "--compare" in sys.argv:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    same = all(np.array_equal(a[k], b[k]) for k in a.files) and sorted(a.files) == sorted(b.files)
    print(f"{len(a['flags'])} cycles, {int(a['flags'].sum())} resamples: {'identical' if same else 'DIFFERENT'}")
    sys.exit(0 if same else 1)

import optbayesexpt_amd as obe            # noqa: E402
import test_gpu_speculative as t          # noqa: E402

out = sys.argv[1]
n_cyc = int(sys.argv[2]) if len(sys.argv) > 2 else 120
warnings.simplefilter("ignore")
digest = lambda x: np.frombuffer(hashlib.sha256(np.ascontiguousarray(x).tobytes()).digest()[:8], dtype=np.uint64)[0]
rows = dict(settings=[], flags=[], state=[], w=[], p=[], idx=[])
for noise, n in ((True, 800000), (False, 1200000)):
    o = t.make(obe, "auto", n_particles=n, n_settings=700, noise_param=noise, threshold=0.9, seed=77)
    meas = np.random.default_rng(3)
    for c in range(n_cyc):
        s = o.opt_setting()
        o.pdf_update((s, t.lorentz(s[0], 3.1, 2.2, 0.4, 0.1) + 0.3 * meas.standard_normal(), 0.3))
        rows["settings"].append(s[0])
        rows["flags"].append(bool(o.just_resampled))
        st = o.rng.bit_generator.state["state"]["state"]
        rows["state"].append(np.uint64(st & 0xFFFFFFFFFFFFFFFF))
        rows["w"].append(digest(o.particle_weights))
        rows["p"].append(digest(o.particles))
        rows["idx"].append(digest(o.last_resample_indices_device.cpu().numpy()) if o.just_resampled else np.uint64(0))
np.savez(out, **{k: np.array(v) for k, v in rows.items()})
print(f"{len(rows['flags'])} cycles, {sum(rows['flags'])} resamples -> {out}")
