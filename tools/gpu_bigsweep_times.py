# -*- coding: utf-8 -*-
"""``dot_solve`` / ``solve`` of one right-hand side at widths 80 .. 1024 on the reference benchmark's kernels: the chunked
affine scans (csrc/bigsweep_kernels.hip; first call = incl. building the chunk maps) against the sequential sweeps
(``CLR_NO_BIG_SWEEP``) and the CPU oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import celerite_amd
from celerite_amd import batch, terms
from oracle import ref


def timed(fn, reps=3):
    best = np.inf
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


np.random.seed(42)
NMAX = 2 ** 19
t_all = np.sort(np.random.rand(NMAX)); yerr_all = np.random.uniform(0.1, 0.2, NMAX); y_all = np.sin(t_all)
E, E2 = np.empty(0), np.empty((0, 0))
print("%5s %7s | %9s %9s %9s | %9s %9s %9s   (ms; first = incl. the chunk maps)" % ("width", "N", "ds first", "ds", "ds seq", "solve", "solve seq", "cpu solve"))
for width, N in ((80, 65536), (128, 65536), (128, 524288), (256, 65536), (512, 65536), (1024, 16384)):
    j = width // 2
    kernel = terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) % 2): kernel += terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) // 2): kernel += terms.ComplexTerm(0.1, 2.0, 1.6)
    cs = [np.asarray(c, dtype=float) for c in kernel.coefficients]
    t, d, y = t_all[:N], yerr_all[:N] ** 2, y_all[:N]
    s = celerite_amd.CholeskySolver()
    s.compute(0.0, *cs, E, E2, E2, t, d)
    z = np.random.RandomState(1).randn(N)
    t0 = time.perf_counter(); q1 = s.dot_solve(z); first = (time.perf_counter() - t0) * 1e3
    ds = timed(lambda: s.dot_solve(z))
    s.solve(z)
    sv = timed(lambda: s.solve(z))
    x_big = s.solve(z)[:, 0]
    batch.set_option("CLR_NO_BIG_SWEEP", "1")
    ds_seq = timed(lambda: s.dot_solve(z), 1)
    sv_seq = timed(lambda: s.solve(z), 1)
    x_seq = s.solve(z)[:, 0]
    batch.set_option("CLR_NO_BIG_SWEEP", None)
    r = ref.RefSolver(); r.compute(0.0, *cs, E, E2, E2, t, d)
    cpu = timed(lambda: r.solve(z), 1)
    print("%5d %7d | %9.3f %9.3f %9.3f | %9.3f %9.3f %9.3f   scan vs seq %.1e" % (width, N, first, ds, ds_seq, sv, sv_seq, cpu, np.max(np.abs(x_big - x_seq)) / np.max(np.abs(x_seq))), flush=True)
