import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from bench import make_inputs
from celerite_amd import batch
from oracle import ref
coeffs, t, diag, y = make_inputs(256, 100000, 0, 16, 11, d_spread=True)
plan = batch.BatchedGP(256, 100000, 0, 16)
plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
plan.enqueue(); plan.synchronize()
tot, k = plan.run_timed(3)
ll, ld, q, st = plan.results()
lv = plan.exact_levels(); g, m = plan.conditioning()
print("ms/step", tot/3, {a: b/3 for a, b in k.items()})
print("levels", np.bincount(lv, minlength=3), "gamma/mu max %.2e median %.2e resid max %.2e" % ((g/m).max(), np.median(g/m), plan.last_residual.max()))
l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[:4] for c in coeffs], t[:4], diag[:4], y[:4])
print("dev", np.max(np.abs(ld[:4]-d0)/np.abs(d0)), np.max(np.abs(q[:4]-q0)/np.abs(q0)))
