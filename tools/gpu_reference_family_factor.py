# -*- coding: utf-8 -*-
"""The reference benchmark's kernels (one or two real terms + IDENTICAL complex terms, examples/benchmark/run.py:80-84)
through ``CholeskySolver``: does the chunked replay's factor agree with the oracle when the replay-vs-scan state residual
is above the certificate's bound?  Runs the solver with the default bound and with a relaxed one
(CLR_SOLVER_CERT_RESID), prints compute time, W / D / solve / dot_solve deviations from the oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import celerite_amd
from celerite_amd import batch, terms
from oracle import ref

np.random.seed(42)
NMAX = 2 ** 19
t_all = np.sort(np.random.rand(NMAX)); yerr_all = np.random.uniform(0.1, 0.2, NMAX); y_all = np.sin(t_all)
E, E2 = np.empty(0), np.empty((0, 0))
bounds = sys.argv[1:] or ["1e-11", "1e-7"]   # (round 6, output check on: the first is the default flow)
for width in (4, 8, 16, 32, 64):
    j = width // 2
    kernel = terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) % 2):
        kernel += terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) // 2):
        kernel += terms.ComplexTerm(0.1, 2.0, 1.6)
    cs = [np.asarray(c, dtype=float) for c in kernel.coefficients]
    for N in (8192, 65536):
        t, yerr, y = t_all[:N], yerr_all[:N], y_all[:N]
        d = yerr ** 2
        r = ref.RefSolver()
        r.compute(0.0, *cs, E, E2, E2, t, d)
        _, _, J, logdet, rphi, ru, rW, rD = r.state()
        z = np.random.RandomState(5).randn(N)
        want_solve, want_ds = r.solve(y)[:, 0], r.dot_solve(z)
        for bound in bounds:
            batch.set_option("CLR_SOLVER_CERT_RESID", bound)
            s = celerite_amd.CholeskySolver()
            s.compute(0.0, *cs, E, E2, E2, t, d)
            t0 = time.perf_counter()
            for _ in range(3):
                s.compute(0.0, *cs, E, E2, E2, t, d)
            ms = (time.perf_counter() - t0) / 3 * 1e3
            ld = s.log_determinant()
            st = s.__getstate__()
            W, D = np.asarray(st[6]).reshape(rW.shape), np.asarray(st[7])
            sv = s.solve(y)[:, 0]
            print("width %2d N %6d bound %-6s route %-22s compute %8.3f ms | logdet %.1e  W %.1e  D %.1e  solve %.1e  dot_solve %.1e" % (
                width, N, bound, "%d/%d/%.1e" % s._route(), ms, abs(ld - logdet) / abs(logdet), np.max(np.abs(W - rW)) / np.max(np.abs(rW)),
                np.max(np.abs(D - rD) / np.abs(rD)), np.max(np.abs(sv - want_solve)) / np.max(np.abs(want_solve)),
                abs(s.dot_solve(z) - want_ds) / abs(want_ds)), flush=True)
        batch.set_option("CLR_SOLVER_CERT_RESID", None)
