"""A/B of two BUILDS (CLR_LIB) on the PLAIN flavour of the wide kernels (no lazy decay): the summarize of a chunked plan with
set_summarize_mode(0) and the one-sweep-per-problem recurrence of a big batch; per-kernel times, oracle deviation on 4 problems."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
batch.LIB_PATH = os.environ["CLR_LIB"]
from bench import make_inputs
from oracle import ref
for (B, N, JR, JC, mode, what) in [(256, 100000, 0, 16, 0, "chunked, plain summarize"), (256, 100000, 2, 6, 0, "chunked, plain summarize"),
                                   (2048, 20000, 0, 16, -1, "one sweep per problem"), (2048, 20000, 2, 6, -1, "one sweep per problem"),
                                   (64, 100000, 0, 32, 0, "chunked, plain summarize")]:
    coeffs, t, diag, y = make_inputs(B, N, JR, JC, 11, d_spread=True)
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
    if mode >= 0:
        plan.set_summarize_mode(mode)
    plan.enqueue(); plan.synchronize()
    tot, k = plan.run_timed(3)
    ll, ld, q, st = plan.results()
    idx = np.arange(0, B, B // 4)
    ll0, ld0, q0, st0 = ref.batch_log_likelihood(0.0, *[c[idx] for c in coeffs], t[idx], diag[idx], y[idx])
    print(os.path.basename(os.environ["CLR_LIB"]), "B=%d N=%d width %d %s (%s): ms/step %.2f" % (B, N, JR + 2 * JC, what, plan.summarize_kernel(), tot / 3),
          {a: round(b / 3, 2) for a, b in k.items() if b / 3 > 0.005}, "chunks", plan.chunks,
          "vs oracle logdet %.1e quad %.1e" % (float(np.max(np.abs(ld[idx] - ld0) / np.abs(ld0))), float(np.max(np.abs(q[idx] - q0) / np.abs(q0)))),
          "checksum %.12e" % float(np.sum(ld)), flush=True)
    plan.close()
