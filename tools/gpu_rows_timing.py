import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import celerite_amd
from celerite_amd import terms
np.random.seed(42)
NMAX = 2 ** 19
t_all = np.sort(np.random.rand(NMAX)); yerr_all = np.random.uniform(0.1, 0.2, NMAX)
E, E2 = np.empty(0), np.empty((0, 0))
width, N = 128, 65536
j = width // 2
kernel = terms.RealTerm(1.0, 0.1)
for k in range((2 * j - 1) % 2): kernel += terms.RealTerm(1.0, 0.1)
for k in range((2 * j - 1) // 2): kernel += terms.ComplexTerm(0.1, 2.0, 1.6)
cs = [np.asarray(c, dtype=float) for c in kernel.coefficients]
t, d = t_all[:N], yerr_all[:N] ** 2
s = celerite_amd.CholeskySolver()
s.compute(0.0, *cs, E, E2, E2, t, d)
t0 = time.perf_counter(); s.compute(0.0, *cs, E, E2, E2, t, d); ms = (time.perf_counter() - t0) * 1e3
st = s.__getstate__()
print("compute %.2f ms = %.3f us/step; cycles per step: pre %.0f  Y %.0f  B1+total %.0f  X %.0f  B2 %.0f" % ((ms, ms * 1e3 / N) + tuple(np.asarray(st[4]).reshape(width, N - 1)[:5, 0])))
