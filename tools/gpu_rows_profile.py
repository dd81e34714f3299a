# -*- coding: utf-8 -*-
"""A short run of the object API's kernels above width 64 for rocprofv3 (tools/prof_run_r05.sh rows): compute
(rows_kernels.hip) at widths 128 / 512, dot_solve / solve (bigsweep_kernels.hip), dot_L (wsweep_kernels.hip) on the
reference benchmark's kernels."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import celerite_amd
from celerite_amd import terms

np.random.seed(42)
NMAX = 2 ** 19
t_all = np.sort(np.random.rand(NMAX)); yerr_all = np.random.uniform(0.1, 0.2, NMAX)
E, E2 = np.empty(0), np.empty((0, 0))
for width, N in ((128, 65536), (512, 16384)):
    j = width // 2
    kernel = terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) % 2): kernel += terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) // 2): kernel += terms.ComplexTerm(0.1, 2.0, 1.6)
    cs = [np.asarray(c, dtype=float) for c in kernel.coefficients]
    t, d = t_all[:N], yerr_all[:N] ** 2
    z = np.random.RandomState(1).randn(N)
    s = celerite_amd.CholeskySolver()
    for _ in range(2):
        s.compute(0.0, *cs, E, E2, E2, t, d)
        for _ in range(3):
            s.dot_solve(z)
        s.solve(z)
        s.dot_L(z)
    print("width", width, "N", N, "log det", s.log_determinant())
