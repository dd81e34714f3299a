import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import celerite_amd
from celerite_amd import terms
from oracle import ref
def best_of_3(fn, min_time=0.05):
    best = np.inf
    for _ in range(3):
        n, t0 = 0, time.perf_counter()
        while True:
            fn(); n += 1
            dt = time.perf_counter() - t0
            if dt >= min_time: break
        best = min(best, dt / n)
    return best
np.random.seed(42)
NMAX = 2 ** 19
t_all = np.sort(np.random.rand(NMAX)); yerr_all = np.random.uniform(0.1, 0.2, NMAX); y_all = np.sin(t_all)
E, E2 = np.empty(0), np.empty((0, 0))
for width, N in ((80, 8192), (128, 8192), (128, 65536), (256, 8192), (512, 8192), (1024, 2048)):
    j = width // 2
    kernel = terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) % 2): kernel += terms.RealTerm(1.0, 0.1)
    for k in range((2 * j - 1) // 2): kernel += terms.ComplexTerm(0.1, 2.0, 1.6)
    coeffs = kernel.coefficients
    gp = celerite_amd.GP(kernel)
    t, yerr, y = t_all[:N], yerr_all[:N], y_all[:N]
    gp.compute(t, yerr); ll = gp.log_likelihood(y)
    g_comp = best_of_3(lambda: gp.compute(t, yerr))
    r = ref.RefSolver(); cs = [np.asarray(c, dtype=float) for c in coeffs]; d = yerr ** 2
    t0 = time.perf_counter(); r.compute(0.0, *cs, E, E2, E2, t, d); c_comp = time.perf_counter() - t0
    ll0 = -0.5 * (r.dot_solve(y) + r.log_determinant() + N * np.log(2 * np.pi))
    print("width %4d N %6d gpu %9.3f ms (%.2f us/step) cpu %9.3f ms  x%.2f  rel %.1e" % (width, N, g_comp*1e3, g_comp*1e6/N, c_comp*1e3, c_comp/g_comp, abs(ll-ll0)/abs(ll0)), flush=True)
