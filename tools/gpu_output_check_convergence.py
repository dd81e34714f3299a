import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import celerite_amd
from celerite_amd import batch, terms
np.random.seed(42)
NMAX = 2 ** 19
t_all = np.sort(np.random.rand(NMAX)); yerr_all = np.random.uniform(0.1, 0.2, NMAX)
E, E2 = np.empty(0), np.empty((0, 0))
batch.set_option("CLR_OUTPUT_CHECK_TOL", "1e-300")
for width in (16, 32):
    j = width // 2
    kernel = terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) % 2): kernel += terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) // 2): kernel += terms.ComplexTerm(0.1, 2.0, 1.6)
    cs = [np.asarray(c, dtype=float) for c in kernel.coefficients]
    for N in (8192, 65536, 2**19):
        t, d = t_all[:N], yerr_all[:N] ** 2
        out = []
        for a in range(1, 9):
            batch.set_option("CLR_OUTPUT_CHECK_ATTEMPTS", str(a))
            s = celerite_amd.CholeskySolver()
            s.compute(0.0, *cs, E, E2, E2, t, d)
            out.append("%.1e" % s._route()[2])
        print("width", width, "N", N, "mismatch of attempt 1..8:", " ".join(out), flush=True)
