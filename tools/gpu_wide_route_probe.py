# -*- coding: utf-8 -*-
"""One series at width 26 (4 real + 11 complex), N = 1e5: route (0 chunk summaries, 1 checked chunked replay, 2 sequential), conditioning
record and time by chunk count and prefix form, through a one-problem plan (the same kernels the object API runs)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celerite_amd import batch
for (JR, JC) in ((4, 11), (2, 7), (0, 16)):
    N = 100000
    rng = np.random.RandomState(JR * 100 + JC)
    t = np.sort(rng.uniform(0, 0.05 * N, N))
    yerr = rng.uniform(0.3, 0.5, N)
    y = rng.randn(N)
    co = (np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
          np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)))
    co = [c[None, :] for c in co]
    plan = batch.BatchedGP(1, N, JR, JC)
    plan.set_series(t[None], (yerr ** 2)[None], y[None])
    for mode in ("walk", "multilevel"):
        for nchunk in (16, 46, 64, 128, 256, 390):
            for exact in (False, True):
                plan.set_prefix_mode(mode)
                plan.set_chunks(nchunk)
                plan.set_exact(exact)
                plan.set_coefficients(*co)
                ll, ld, q, st = plan.log_likelihood()
                t0 = time.perf_counter()
                for _ in range(3):
                    plan.set_coefficients(*co); ll, ld, q, st = plan.log_likelihood()
                dt = (time.perf_counter() - t0) / 3
                g, m = plan.conditioning()
                print("(%d,%d) %-10s chunks %4d exact %d: route %s  %.2f ms  logdet %.15e quad %.15e  gamma %.2e mu %.2e resid %.2e eG %.2e"
                      % (JR, JC, mode, plan.chunks[0], exact, plan.exact_levels().tolist(), dt * 1e3, ld[0], q[0], g[0], m[0], plan.last_residual[0], plan.measured_error()[0]), flush=True)
    plan.close()
