# -*- coding: utf-8 -*-
"""Which chunk lengths send the width-26 series of tools/gpu_single.py to the sequential route, and what the conditioning record says."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celerite_amd import batch
JR, JC, N = 4, 11, 100000
rng = np.random.RandomState(JR * 100 + JC)
t = np.sort(rng.uniform(0, 0.05 * N, N))
yerr = rng.uniform(0.3, 0.5, N)
y = rng.randn(N)
co = (np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
      np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)))
print("c_real", co[1], "c_comp", co[4])
co = [c[None, :] for c in co]
plan = batch.BatchedGP(1, N, JR, JC)
plan.set_series(t[None], (yerr ** 2)[None], y[None])
seen = set()
for nchunk in range(150, 400):
    plan.set_chunks(nchunk)
    if plan.chunks in seen: continue
    seen.add(plan.chunks)
    for exact in (False, True):
        plan.set_exact(exact)
        plan.set_coefficients(*co)
        ll, ld, q, st = plan.log_likelihood()
        lv = plan.exact_levels().tolist()
        g, m = plan.conditioning()
        if lv[0] == 2 or nchunk % 50 == 0:
            print("chunks %s exact %d: route %s status %s logdet %.12e gamma %.2e mu %.2e resid %.2e eG %.2e chunkwise %.2e"
                  % (plan.chunks, exact, lv, st.tolist(), ld[0], g[0], m[0], plan.last_residual[0], plan.measured_error()[0], plan.conditioning_chunkwise()[0]), flush=True)
plan.close()
