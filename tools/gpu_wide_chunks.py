"""BASELINE config 4 (B=256, N=1e5, width 32): chunk-count sweep of the batched wide flow (time per kernel, routes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_inputs
from celerite_amd import batch
coeffs, t, diag, y = make_inputs(256, 100000, 0, 16, 11, d_spread=True)
plan = batch.BatchedGP(256, 100000, 0, 16)
plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
ref = None
for nc in (0, 24, 32, 48, 64):
    plan.set_chunks(nc)
    plan.enqueue(); plan.synchronize()
    tot, k = plan.run_timed(3)
    ll, ld, q, st = plan.results()
    if ref is None: ref = (ld.copy(), q.copy())
    lv = plan.exact_levels()
    print("chunks %s ms/step %.2f" % (plan.chunks, tot / 3), {a: round(b / 3, 2) for a, b in k.items()},
          "levels", np.bincount(lv, minlength=3), "vs auto: %.1e %.1e" % (np.max(np.abs(ld - ref[0]) / np.abs(ref[0])), np.max(np.abs(q - ref[1]) / np.abs(ref[1]))), flush=True)
