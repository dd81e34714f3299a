# -*- coding: utf-8 -*-
"""Which route do the reference benchmark's kernels (one real term + IDENTICAL complex terms, examples/benchmark/run.py:80-84)
take in a plan?  levels, conditioning record, chunk count, time -- widths 8 .. 64, B = 1 and B = 64."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celerite_amd import batch
from oracle import ref

np.random.seed(42)
N = 65536
NMAX = int(sys.argv[1]) if len(sys.argv) > 1 else N   # (run.py draws max(N) times and uses the first n of them)
t = np.sort(np.random.rand(NMAX))[:N]; yerr = np.random.uniform(0.1, 0.2, NMAX)[:N]; y = np.sin(t)
for width in (8, 16, 32, 64):
    JR, JC = 1, (width - 1) // 2
    if (width - 1) % 2:
        JR = 2
    for B in (1, 64):
        a_real = np.full((B, JR), 1.0); c_real = np.full((B, JR), 0.1)
        a_comp = np.full((B, JC), 0.1); b_comp = np.zeros((B, JC)); c_comp = np.full((B, JC), 2.0); d_comp = np.full((B, JC), 1.6)
        plan = batch.BatchedGP(B, N, JR, JC)
        plan.set_series(t, yerr ** 2, y)
        plan.set_coefficients(a_real, c_real, a_comp, b_comp, c_comp, d_comp)
        ll, ld, q, st = plan.log_likelihood()
        t0 = time.perf_counter()
        for _ in range(3):
            plan.enqueue(); plan.results()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        gam, mu = plan.conditioning()
        eg = plan.measured_error()
        # forced-exact / materialising: the chunked replay checked against the scan (what CholeskySolver.compute runs)
        plan.set_exact(True)
        plan.log_likelihood()
        plan.conditioning()
        lv = np.bincount(plan.exact_levels(), minlength=3)[:3]
        t0 = time.perf_counter()
        plan.enqueue(); plan.results()
        ms_exact = (time.perf_counter() - t0) * 1e3
        print("   forced exact: levels %s  replay-vs-scan residual max %.2e  %.2f ms" % (lv, float(np.max(plan.last_residual)), ms_exact))
        plan.set_exact(False)
        l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, a_real[:1], c_real[:1], a_comp[:1], b_comp[:1], c_comp[:1], d_comp[:1], t, yerr ** 2, y)
        print("width %2d B=%2d chunks %s levels %s gamma %.2e mu %.2e eg %.2e  %.2f ms  logdet rel %.1e quad rel %.1e status %s" % (
            width, B, plan.chunks, np.bincount(plan.exact_levels(), minlength=3)[:3], gam.max(), mu.min(), eg.max(), ms,
            abs(ld[0] - d0[0]) / abs(d0[0]), abs(q[0] - q0[0]) / abs(q0[0]), st[:1]), flush=True)
        plan.close()
