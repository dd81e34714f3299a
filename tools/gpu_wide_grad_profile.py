"""One series of 1e5 samples, width 16, 32 and (round 6) 64, CholeskySolver.grad_log_likelihood (chunk-parallel forward mode) for the profiler."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd
from bench import make_inputs
e, e2 = np.empty(0), np.empty((0, 0))
for jc in (8, 16, 32):
    c, t, d, y = make_inputs(1, 100000, 0, jc, 47, d_spread=(jc >= 16))
    args = (0.01,) + tuple(x[0] for x in c) + (e, e2, e2, t[0], y[0], d[0])
    s = celerite_amd.CholeskySolver()
    for _ in range(4):
        v, g = s.grad_log_likelihood(*args)
    print("width", 2 * jc, "value", v, "max partial", float(np.max(np.abs(g))))
