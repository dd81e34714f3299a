"""CholeskySolver.grad_log_likelihood (width 8 = 2 real + 3 complex, 17 partials) against the series length: the plan
gradient parallel in n (from N = 2048) and the sequential tangent kernel (CLR_GRAD_SEQUENTIAL=1), ms per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import celerite_amd
from bench import make_inputs

e, e2 = np.empty(0), np.empty((0, 0))
for N in (512, 1024, 2048, 4096, 8192, 20000, 50000, 100000, 400000):
    coeffs, t, diag, y = make_inputs(1, N, 2, 3, 42)
    args = (0.1,) + tuple(c[0] for c in coeffs) + (e, e2, e2, t[0], y[0], diag[0])
    out = {}
    for mode in ("auto", "sequential"):
        if mode == "sequential":
            os.environ["CLR_GRAD_SEQUENTIAL"] = "1"
        try:
            s = celerite_amd.CholeskySolver()
            s.grad_log_likelihood(*args)
            reps = 3 if N >= 50000 and mode == "sequential" else 10
            t0 = time.perf_counter()
            for _ in range(reps):
                v, g = s.grad_log_likelihood(*args)
            out[mode] = ((time.perf_counter() - t0) / reps * 1e3, g)
        finally:
            os.environ.pop("CLR_GRAD_SEQUENTIAL", None)
    d = np.max(np.abs(out["auto"][1] - out["sequential"][1])) / np.max(np.abs(out["sequential"][1]))
    print("N %7d  grad_log_likelihood %8.3f ms   sequential kernel %8.3f ms   difference %.1e of the largest partial" % (
        N, out["auto"][0], out["sequential"][0], d), flush=True)
