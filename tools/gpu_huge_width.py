# -*- coding: utf-8 -*-
"""CholeskySolver at widths above 128 (csrc/huge_kernels.hip: S in HBM / L2): compute + dot_solve + solve latency against
the CPU oracle, widths 130 .. 1024 (the reference's published benchmark goes to 512, examples/benchmark/run.py:39)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import celerite_amd
from oracle import ref
from _cases import synthetic, coeffs_of

E, E2 = np.empty(0), np.empty((0, 0))
for JC, N in ((65, 2000), (128, 2000), (256, 2000), (512, 1000), (256, 20000)):
    J = 2 * JC
    case = synthetic(1, N, 0, JC, "accuracy", seed=JC)
    cs = coeffs_of(case, 0)
    t, diag, y = case["t"][0], case["diag"][0] + 0.05, case["y"][0]
    s = celerite_amd.CholeskySolver()
    s.compute(0.1, *cs, E, E2, E2, t, diag)
    t0 = time.perf_counter()
    s.compute(0.1, *cs, E, E2, E2, t, diag)
    tc = time.perf_counter() - t0
    t0 = time.perf_counter()
    q = s.dot_solve(y)
    td = time.perf_counter() - t0
    t0 = time.perf_counter()
    x = s.solve(y)
    ts = time.perf_counter() - t0
    r = ref.RefSolver()
    t0 = time.perf_counter()
    r.compute(0.1, *cs, E, E2, E2, t, diag)
    rc = time.perf_counter() - t0
    print("width %4d N=%5d: compute %.2f ms (%.2f us per step)  dot_solve %.2f ms  solve %.2f ms | CPU oracle compute %.1f ms | logdet rel %.1e  solve rel %.1e" % (
        J, N, tc * 1e3, tc / N * 1e6, td * 1e3, ts * 1e3, rc * 1e3, abs(s.log_determinant() - r.log_determinant()) / abs(r.log_determinant()),
        np.max(np.abs(x[:, 0] - r.solve(y)[:, 0])) / np.max(np.abs(x))), flush=True)
