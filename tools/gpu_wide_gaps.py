"""Wide plans on series that are NOT densely sampled everywhere: the lazy flavour (automatic since round 5 whenever max c x
max dx < 2: a gap sends its wave through the full sincos / exp for one batch) against the plain flavour (set_summarize_mode(0)).
256 x 1e5 at widths 32 (16 complex terms) and 14 (2 real + 6 complex): the dense series of BASELINE configs[4]; the same with
1 % / 10 % of the steps stretched 300-fold (observing gaps); the whole time axis stretched 30-fold and 300-fold (sparse).
Per case: the kernel chosen, ms per step by HIP events, routes, deviation from the CPU oracle on 4 problems."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
if os.environ.get("CLR_LIB"):
    batch.LIB_PATH = os.environ["CLR_LIB"]
from bench import make_inputs
from oracle import ref
B, N = 256, 100000
for (JR, JC) in ((0, 16), (2, 6)):
    coeffs, t0, diag, y = make_inputs(B, N, JR, JC, 11, d_spread=True)
    rng = np.random.default_rng(3)
    for label, frac, stretch, scale in (("dense", 0.0, 1.0, 1.0), ("1 % gaps x300", 0.01, 300.0, 1.0), ("10 % gaps x300", 0.10, 300.0, 1.0),
                                        ("axis x30", 0.0, 1.0, 30.0), ("axis x300", 0.0, 1.0, 300.0)):
        dt = np.diff(t0, axis=1, prepend=0.0) * scale
        if frac > 0:
            dt = np.where(rng.random(dt.shape) < frac, dt * stretch, dt)
        t = np.cumsum(dt, axis=1)
        yy = np.sin(3.0 * t / scale)
        for mode in (-1, 0):
            plan = batch.BatchedGP(B, N, JR, JC)
            plan.set_series(t, diag, yy); plan.set_coefficients(*coeffs)
            plan.set_summarize_mode(mode)
            plan.enqueue(); plan.synchronize()
            tot, k = plan.run_timed(3)
            ll, ld, q, st = plan.results()
            idx = np.arange(0, B, B // 4)
            l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[idx] for c in coeffs], t[idx], diag[idx], yy[idx])
            sb = plan.selection_bounds()
            print("width %2d %-15s mode %2d %-28s %6.2f ms (summarize %6.2f replay %5.2f) routes %s  vs oracle %.1e / %.1e  status equal %s  cmax*dxmax %.2e" % (
                JR + 2 * JC, label, mode, plan.summarize_kernel(), tot / 3, k["summarize"] / 3, k["replay"] / 3,
                np.bincount(plan.exact_levels(), minlength=3).tolist(), float(np.max(np.abs(ld[idx] - d0) / np.abs(d0))),
                float(np.max(np.abs(q[idx] - q0) / np.abs(q0))), bool(np.array_equal(st[idx], s0)), sb["cmax"] * sb["dxmax"]), flush=True)
            plan.close()
