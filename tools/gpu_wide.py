# -*- coding: utf-8 -*-
"""Development probe: BASELINE config 5 (width 32 = 16 complex terms, N = 1e5, batch 256)
on the wide (wave-per-problem) path: oracle parity on a sample + HIP-event timing."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celerite_amd import batch  # noqa: E402
from oracle import ref  # noqa: E402


def mk(B, N, JR, JC, seed):
    rng = np.random.RandomState(seed)
    t = np.sort(rng.rand(B, N), axis=1); sig = rng.uniform(0.1, 0.2, (B, N)); y = np.sin(t)
    ar = np.exp(1.0 + 0.1 * rng.randn(B, JR)); cr = np.exp(0.1 + 0.1 * rng.randn(B, JR))
    ac = np.exp(0.1 + 0.1 * rng.randn(B, JC)); bc = np.zeros((B, JC))
    cc = np.exp(2.0 + 0.1 * rng.randn(B, JC)); dc = np.exp(rng.uniform(0.0, 3.0, (B, JC)))
    return (ar, cr, ac, bc, cc, dc), t, sig ** 2, y


for (B, N, JR, JC) in [(64, 100000, 0, 16), (256, 100000, 0, 16), (1024, 100000, 0, 16), (256, 100000, 2, 5), (1024, 100000, 2, 5), (1024, 100000, 0, 32)]:
    co, t, d, y = mk(B, N, JR, JC, 3)
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_series(t, d, y)
    plan.set_coefficients(*co)
    ll, ld, q, st = plan.log_likelihood()
    tot, k = plan.run_timed(3)
    S = 4
    t0 = time.time()
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[:S] for c in co], t[:S], d[:S], y[:S])
    cpu = (time.time() - t0) / S
    print("B=%d N=%d width %d (%d real + %d complex), %d chunks: %.2f ms per batch [%s] -> %.0f loglik/s ; parity (%d problems) "
          "logdet %.2e quad %.2e status ok %s ; CPU oracle %.1f ms each (%.0fx one core)" % (
              B, N, JR + 2 * JC, JR, JC, plan.chunks[0], tot / 3, " ".join("%s %.2f" % (a, v / 3) for a, v in k.items() if v > 0.003), B / (tot / 3) * 1e3, S,
              np.max(np.abs(ld[:S] - d0) / np.abs(d0)), np.max(np.abs(q[:S] - q0) / np.abs(q0)),
              bool((st == 0).all() and (s0 == 0).all()), cpu * 1e3, B / (tot / 3) * 1e3 * cpu), flush=True)
    plan.close()
