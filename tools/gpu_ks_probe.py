"""Routes and conditioning records of a wide plan (B = 128, N = 1e5, width 32, 16 chunks) with the prefix as a walk and as the parallel scan."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["CLR_WIDE_SCAN_CAP"] = "100000"
from bench import make_inputs
from celerite_amd import batch
B, nchunk = 128, 16
coeffs, t, diag, y = make_inputs(B, 100000, 0, 16, seed=B, d_spread=True)
plan = batch.BatchedGP(B, 100000, 0, 16)
plan.set_series(t, diag, y)
out = {}
for mode in ("walk", "multilevel"):
    plan.set_prefix_mode(mode); plan.set_chunks(nchunk); plan.set_coefficients(*coeffs)
    ll, ld, q, st = plan.log_likelihood()
    lv = plan.exact_levels()
    g, m = plan.conditioning()
    e = plan.measured_error()
    out[mode] = (ld, q)
    bad = np.nonzero(lv)[0]
    print(mode, "routes", np.bincount(lv, minlength=3).tolist(), "gamma max %.2e mu min %.2e eG max %.2e  gamma*eG max %.2e  gamma/mu max %.2e" % (g.max(), m.min(), e.max(), (g * e).max(), (g / m).max()))
    for b in bad[:5]:
        print("   problem", b, "level", lv[b], "gamma %.3e mu %.3e eG %.3e gamma*eG %.3e" % (g[b], m[b], e[b], g[b] * e[b]))
print("scan vs walk: logdet %.2e quad %.2e" % (np.max(np.abs(out["walk"][0] - out["multilevel"][0]) / np.abs(out["walk"][0])), np.max(np.abs(out["walk"][1] - out["multilevel"][1]) / np.abs(out["walk"][1]))))
plan.close()
