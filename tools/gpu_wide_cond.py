import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
from bench import make_inputs
coeffs, t, diag, y = make_inputs(256, 100000, 0, 16, 11, d_spread=True)
for ratio in ("1.18", "1.25"):
    os.environ["CLR_WIDE_FIRST_RATIO"] = ratio
    plan = batch.BatchedGP(256, 100000, 0, 16)
    plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
    ll, ld, q, st = plan.log_likelihood()
    lev = plan.exact_levels(); g, m = plan.conditioning(); e = plan.measured_error()
    bad = np.nonzero(lev)[0]
    print("ratio", ratio, "chunks", plan.chunks, "levels", np.bincount(lev, minlength=3), "gamma max %.3g mu min %.3g g/mu max %.3g g*e max %.3g" % (g.max(), m.min(), (g / m).max(), (g * e).max()))
    for b in bad:
        print("   problem", b, "gamma %.4g mu %.4g gamma/mu %.4g eG %.3g gamma*eG %.3g" % (g[b], m[b], g[b] / m[b], e[b], g[b] * e[b]))
    plan.close()
