# -*- coding: utf-8 -*-
"""The OBJECT API's factor (CholeskySolver.compute, one long series) against the binary128 recurrence and the oracle:
W, D, solve -- by position inside the chunk is not known here (the solver picks its own chunking), so worst entries."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import celerite_amd
from oracle import ref

for (N, JR, JC, spread) in ((100000, 2, 3, False), (100000, 0, 8, True), (100000, 0, 16, True), (20000, 2, 3, False), (3000, 1, 1, False)):
    coeffs, t, diag, y = bench.make_inputs(2, N, JR, JC, 42, d_spread=spread)
    p = 0
    cs = [c[p] for c in coeffs]
    e_, e2_ = np.empty(0), np.empty((0, 0))
    s = celerite_amd.CholeskySolver()
    s.compute(0.0, *cs, e_, e2_, e2_, t[p], diag[p])
    st = s.__getstate__()
    W, D = np.asarray(st[6]), np.asarray(st[7])
    x = s.solve(y[p])[:, 0]
    Wq, Dq, xq, ldq, qq = ref.quad_factor_solve(0.0, *cs, t[p], diag[p], y[p])
    r = ref.RefSolver()
    r.compute(0.0, *cs, e_, e2_, e2_, t[p], diag[p])
    xs = r.solve(y[p])[:, 0]
    W = W.reshape(Wq.shape) if W.shape == Wq.shape else W.reshape(Wq.shape[::-1]).T
    print("N=%d width %d: device vs truth  W %.2e  D %.2e  solve %.2e  logdet %.2e | oracle vs truth solve %.2e | device vs oracle solve %.2e" % (
        N, JR + 2 * JC, np.max(np.abs(W - Wq)) / np.max(np.abs(Wq)), np.max(np.abs(D - Dq) / np.abs(Dq)),
        np.max(np.abs(x - xq)) / np.max(np.abs(xq)), abs(s.log_determinant() - ldq) / abs(ldq),
        np.max(np.abs(xs - xq)) / np.max(np.abs(xq)), np.max(np.abs(x - xs)) / np.max(np.abs(xs))), flush=True)
