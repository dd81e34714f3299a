# -*- coding: utf-8 -*-
"""Mid-size wide batches (B = 32..256 series of N = 1e5, widths 16 / 32): ms per evaluation by chunk count, with the prefix as a walk (one workgroup per
problem) and as the parallel scan (CLR_WIDE_SCAN_CAP lifted)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from celerite_amd import batch
os.environ["CLR_WIDE_SCAN_CAP"] = "100000"
for JC in (8, 16):
    for B in (32, 64, 128, 256):
        coeffs, t, diag, y = make_inputs(B, 100000, 0, JC, seed=B, d_spread=True)
        plan = batch.BatchedGP(B, 100000, 0, JC)
        plan.set_series(t, diag, y)
        row = []
        for nchunk in (0, 8, 16, 32, 64, 128):
            if B * max(nchunk, 1) > 16384: continue
            for mode in ("walk", "multilevel"):
                plan.set_prefix_mode(mode)
                plan.set_chunks(nchunk)
                plan.set_coefficients(*coeffs)
                plan.log_likelihood()
                t0 = time.perf_counter()
                for _ in range(3):
                    plan.set_coefficients(*coeffs); ll, ld, q, st = plan.log_likelihood()
                dt = (time.perf_counter() - t0) / 3
                row.append("%s/%s: %.2f" % (plan.chunks[0] if nchunk else "auto%d" % plan.chunks[0], "walk" if mode == "walk" else "scan", dt * 1e3))
        print("width %d B=%d ms by chunks/prefix: %s  routes %s" % (2 * JC, B, "  ".join(row), np.bincount(plan.exact_levels(), minlength=3).tolist()), flush=True)
        plan.close()
