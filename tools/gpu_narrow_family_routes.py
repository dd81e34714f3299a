# -*- coding: utf-8 -*-
"""Route and time of ``CholeskySolver.compute`` on the reference benchmark's kernels (identical complex terms,
examples/benchmark/run.py:80-84) at widths 2 .. 8 (the narrow plan kernels), t = sort(rand(N)) on [0, 1] for several N."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import celerite_amd
from celerite_amd import terms
from oracle import ref

E, E2 = np.empty(0), np.empty((0, 0))
for width in (2, 4, 6, 8):
    j = width // 2
    kernel = terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) % 2):
        kernel += terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) // 2):
        kernel += terms.ComplexTerm(0.1, 2.0, 1.6)
    cs = [np.asarray(c, dtype=float) for c in kernel.coefficients]
    for N in (4096, 16384, 65536, 262144):
        rng = np.random.RandomState(N + width)
        t = np.sort(rng.rand(N)); d = rng.uniform(0.1, 0.2, N) ** 2; y = np.sin(t)
        r = ref.RefSolver()
        t0 = time.perf_counter(); r.compute(0.0, *cs, E, E2, E2, t, d); cpu = time.perf_counter() - t0
        _, _, J, logdet, rphi, ru, rW, rD = r.state()
        s = celerite_amd.CholeskySolver()
        s.compute(0.0, *cs, E, E2, E2, t, d)
        t0 = time.perf_counter()
        for _ in range(3):
            s.compute(0.0, *cs, E, E2, E2, t, d)
        ms = (time.perf_counter() - t0) / 3 * 1e3
        route = s._route()
        st = s.__getstate__()
        W, D = np.asarray(st[6]).reshape(rW.shape), np.asarray(st[7])
        print("width %d N %6d route %d/%d/%.1e  compute %8.3f ms (cpu %7.3f)  logdet %.1e W %.1e D %.1e" % (
            width, N, route[0], route[1], route[2], ms, cpu * 1e3, abs(s.log_determinant() - logdet) / abs(logdet),
            np.max(np.abs(W - rW)) / np.max(np.abs(rW)), np.max(np.abs(D - rD) / np.abs(rD))), flush=True)
