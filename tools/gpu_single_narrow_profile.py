"""One series of 1e5 samples through CholeskySolver (hint + compute + dot_solve + log_determinant) at widths 3 and 8, for the profiler."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd
for JR, JC in [(1, 1), (2, 3)]:
    N = 100000
    rng = np.random.RandomState(JR * 100 + JC)
    t = np.sort(rng.uniform(0, 0.05 * N, N))
    yerr = rng.uniform(0.3, 0.5, N)
    y = rng.randn(N)
    args = (0.0, np.exp(rng.uniform(-1, 0.5, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0.5, JC)),
            np.zeros(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(0, 3, JC)),
            np.empty(0), np.empty((0, 0)), np.empty((0, 0)), t, yerr ** 2)
    s = celerite_amd.CholeskySolver()
    for _ in range(5):
        s._hint_rhs(y); s.compute(*args)
        q, ld = s.dot_solve(y), s.log_determinant()
    print("width", JR + 2 * JC, "logdet", ld, "quad", q)
