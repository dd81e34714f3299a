"""Headline shape (B=1024, N=1e5, 2 real + 3 complex): summarize time per kernel variant + parity on 8 problems."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_inputs
from celerite_amd import batch
from oracle import ref
B, N = 1024, 100000
coeffs, t, diag, y = make_inputs(B, N, 2, 3, seed=42)
l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[:8] for c in coeffs], t[:8], diag[:8], y[:8])
plan = batch.BatchedGP(B, N, 2, 3)
plan.set_series(t, diag, y)
plan.set_coefficients(*coeffs)
for mode in (-1, 1, 0, -1):
    plan.set_summarize_mode(mode)
    plan.enqueue(); plan.synchronize()
    tot, k = plan.run_timed(10)
    ll, ld, q, st = plan.results()
    print("mode %2d %-28s step %.3f ms summarize %.3f prefix %.3f  parity logdet %.1e quad %.1e status %s" % (
        mode, plan.summarize_kernel(), tot / 10, k["summarize"] / 10, k["prefix"] / 10,
        np.max(np.abs(ld[:8] - d0) / np.abs(d0)), np.max(np.abs(q[:8] - q0) / np.abs(q0)), bool((st[:8] == s0).all())), flush=True)
