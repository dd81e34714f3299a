"""Width 32 (BASELINE config 4 family, N = 1e5): chunk-count sweep of the batched wide flow for several batch sizes
(time per kernel, routes).  Round 3: with the re-calibrated routing no problem of this family is replayed, so the
trade-off is summarize rounds against the sequential prefix + the correct phase."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_inputs
from celerite_amd import batch
for B, counts in ((256, (0, 4, 6, 8, 10, 12, 16, 24)), (64, (0, 8, 16, 32)), (128, (0, 8, 16)), (512, (0, 2, 4, 8)), (1024, (0, 2, 4))):
    coeffs, t, diag, y = make_inputs(B, 100000, 0, 16, 11, d_spread=True)
    plan = batch.BatchedGP(B, 100000, 0, 16)
    plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
    ref = None
    for nc in counts:
        plan.set_chunks(nc)
        plan.enqueue(); plan.synchronize()
        tot, k = plan.run_timed(3)
        ll, ld, q, st = plan.results()
        if ref is None: ref = (ld.copy(), q.copy())
        lv = plan.exact_levels()
        print("B %4d chunks %-10s%s ms/step %6.2f" % (B, plan.chunks, " (auto)" if nc == 0 else "", tot / 3), {a: round(b / 3, 2) for a, b in k.items() if b > 0.03},
              "levels", np.bincount(lv, minlength=3), "vs auto: %.1e %.1e" % (np.max(np.abs(ld - ref[0]) / np.abs(ref[0])), np.max(np.abs(q - ref[1]) / np.abs(ref[1]))), flush=True)
    plan.close()
