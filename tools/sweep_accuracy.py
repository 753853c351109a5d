"""Absolute accuracy of the full sweep against the oracle's two-pass variance with IEEE divisions
(developer aid): max relative error of the variance over sampled settings, fresh cloud and after
updates, shifted and unshifted.   python tools/sweep_accuracy.py [c2|c3]   (OBE_VARIANT=<name>)"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
if os.environ.get("OBE_VARIANT"):
    from optbayesexpt_amd import _lib as _l
    _l._LIB = _l.HipLib(os.path.join(ROOT, "tools", "_variants", f"libobe_hip_{os.environ['OBE_VARIANT']}.so"), allow_variant=True)
import oracle
from oracle import models as om

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
settings, prior, cons, true, sigma = bench.make_workload(cfg)
o = bench.build_obe(cfg, None, settings, prior.copy(), cons)
o.rng = np.random.default_rng(1234)
sim = np.random.default_rng(4321)
ns = settings[0].size
sample = np.unique(np.r_[0, ns - 1, np.random.default_rng(5).integers(0, ns, 24)])
fn = om.lorentzian
warnings.simplefilter("ignore")
tag = os.environ.get("OBE_VARIANT", "tree")
for stage in range(4):
    w = np.array(o.particle_weights)
    p = np.array(o.particles)
    ref = oracle.yvar_full_sweep(fn, oracle.flatten_settings(settings)[:, sample], p, w, cons, chunk=1 << 15)[0]
    for mode in ("always", "never"):
        o.tuning_parameters["sweep_shift"] = mode
        got = o.yvar_from_parameter_draws()[0][sample]
        print(f"{tag:6s} {cfg} after {stage * 4:2d} updates, shift {mode:6s}: max rel. error {np.max(np.abs(got - ref) / ref):.2e}  "
              f"(kappa {o.last_sweep['kappa']:.3g})")
    o.tuning_parameters["sweep_shift"] = "auto"
    for _ in range(4):
        x = o.opt_setting()
        o.pdf_update((x, float(fn(x, true, cons)) + sigma * sim.standard_normal(), sigma))
