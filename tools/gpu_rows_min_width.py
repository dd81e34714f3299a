# -*- coding: utf-8 -*-
"""Where should ``CholeskySolver.compute`` with GENERAL terms switch from the LDS-resident any-width kernel
(generic_kernels.hip) to the row-distributed one (rows_kernels.hip, padded to width 128)?  us per sample of both for
total widths 5 .. 40 (``CLR_ROWS_MIN_WIDTH``)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import celerite_amd
from celerite_amd import batch
from _cases import synthetic, coeffs_of

N = 20000
for JR, JC in ((1, 0), (1, 2), (2, 5), (2, 9), (2, 13), (2, 17)):
    case = synthetic(1, N, JR, JC, "accuracy", seed=JR + JC)
    cs = list(coeffs_of(case, 0))
    t, diag = case["t"][0], case["diag"][0] + 0.05
    a1, w1, a2, w2 = 0.3, 1.7, 0.2, 0.45
    U = np.vstack([a1 * np.cos(w1 * t), a1 * np.sin(w1 * t), a2 * np.cos(w2 * t), a2 * np.sin(w2 * t)])
    V = np.vstack([np.cos(w1 * t), np.sin(w1 * t), np.cos(w2 * t), np.sin(w2 * t)])
    A = np.full(N, a1 + a2)
    out = []
    for mn in ("200", "1"):
        batch.set_option("CLR_ROWS_MIN_WIDTH", mn)
        s = celerite_amd.CholeskySolver()
        s.compute(0.1, *cs, A, U, V, t, diag)
        t0 = time.perf_counter(); s.compute(0.1, *cs, A, U, V, t, diag); dt = time.perf_counter() - t0
        out.append((dt * 1e6 / N, s.log_determinant()))
    batch.set_option("CLR_ROWS_MIN_WIDTH", None)
    print("total width %2d: LDS kernel %.2f us per sample, rows kernel %.2f  (log det agree to %.1e)" % (
        JR + 2 * JC + 4, out[0][0], out[1][0], abs(out[0][1] - out[1][1]) / abs(out[0][1])), flush=True)
