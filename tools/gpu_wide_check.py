"""BASELINE config 4 (B=256, N=1e5, width 32): time per kernel, routes, parity on a sample; A/B of the
wide summarize with and without the lazy decay."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bench import make_inputs
from celerite_amd import batch
from oracle import ref
coeffs, t, diag, y = make_inputs(256, 100000, 0, 16, 11, d_spread=True)
l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[:4] for c in coeffs], t[:4], diag[:4], y[:4])
plan = batch.BatchedGP(256, 100000, 0, 16)
plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
outs = {}
for mode in (0, -1, 0, -1):
    plan.set_summarize_mode(mode)
    plan.enqueue(); plan.synchronize()
    tot, k = plan.run_timed(3)
    ll, ld, q, st = plan.results()
    outs[mode] = (ld, q)
    lv = plan.exact_levels(); g, m = plan.conditioning()
    print("mode %2d chunks %s ms/step %.2f" % (mode, plan.chunks, tot / 3), {a: round(b / 3, 2) for a, b in k.items()},
          "levels", np.bincount(lv, minlength=3), "dev vs oracle %.1e %.1e" % (np.max(np.abs(ld[:4] - d0) / np.abs(d0)), np.max(np.abs(q[:4] - q0) / np.abs(q0))))
print("lazy vs plain over the batch: logdet %.1e quad %.1e" % (np.max(np.abs(outs[-1][0] - outs[0][0]) / np.abs(outs[0][0])), np.max(np.abs(outs[-1][1] - outs[0][1]) / np.abs(outs[0][1]))))
