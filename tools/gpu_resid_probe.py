import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from celerite_amd import batch
from oracle import ref
from _cases import synthetic, coeffs_of
for fam in ("bench",):
    for (JR, JC) in [(8, 0), (5, 0), (6, 1), (3, 0), (2, 3), (0, 4)]:
        for N, nchunk in [(8000, 64), (8000, 250), (100000, 64)]:
            case = synthetic(6, N, JR, JC, fam, seed=3)
            plan = batch.BatchedGP(6, N, JR, JC); plan.set_chunks(nchunk)
            plan.set_series(case["t"], case["diag"], case["y"]); plan.set_coefficients(*coeffs_of(case))
            plan.set_exact(True); plan.set_certificate(1e6, 1e300)
            ll, ld, q, st = plan.log_likelihood(); gam, mu = plan.conditioning(); res = plan.last_residual
            S = 2
            l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[:S] for c in coeffs_of(case)], case["t"][:S], case["diag"][:S], case["y"][:S])
            dev = max(np.max(np.abs(ld[:S] - d0) / np.abs(d0)), np.max(np.abs(q[:S] - q0) / np.abs(q0)))
            plan.close()
            print("%-6s (%d,%d) N=%6d chunks %3d: gamma/mu %.2e resid_max %.2e replay-vs-oracle %.2e" % (fam, JR, JC, N, nchunk, (gam/mu).max(), res.max(), dev))
