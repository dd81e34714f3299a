# -*- coding: utf-8 -*-
"""BASELINE configs[1] (256 x 1e4 x width 4) device time of the one-launch path, and parity against the oracle, under the
library given by CLR_LIB (A/B of two builds inside one gpurun call)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celerite_amd import batch
if os.environ.get("CLR_LIB"):
    batch.LIB_PATH = os.environ["CLR_LIB"]
import bench
from oracle import ref

for (B, N, JR, JC) in ((256, 10000, 0, 2), (256, 10000, 2, 1), (256, 2000, 1, 1)):
    coeffs, t, diag, y = bench.make_inputs(B, N, JR, JC, 7)
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_series(t, diag, y)
    plan.set_coefficients(*coeffs)
    ll, ld, q, st = plan.log_likelihood()
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[:32] for c in coeffs], t[:32], diag[:32], y[:32])
    err = max(np.max(np.abs(ld[:32] - d0) / np.abs(d0)), np.max(np.abs(q[:32] - q0) / np.abs(q0)))
    best = 1e9
    for _ in range(5):
        ms, k = plan.run_timed(50, relayout_each_step=False)
        best = min(best, ms / 50)
    draws = [coeffs] + bench.fresh_draws(coeffs, 3, 8)
    bench.real_loop(plan, draws, 20)
    t0 = time.perf_counter()
    bench.real_loop(plan, draws, 200, offset=1)
    loop = (time.perf_counter() - t0) / 200 * 1e3
    print("%s B=%d N=%d shape (%d,%d) small path %s: device %.4f ms  real loop %.4f ms  err %.1e  status %d  checksum %.13e" % (
        os.path.basename(batch.LIB_PATH), B, N, JR, JC, plan.small_mode_active(), best, loop, err, int(st.sum()), float(np.sum(ld))), flush=True)
    plan.close()
