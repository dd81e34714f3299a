"""One series of 1e5 samples, width 8: CholeskySolver.grad_log_likelihood (reverse mode on a one-problem plan) for the profiler."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd
from bench import make_inputs
e, e2 = np.empty(0), np.empty((0, 0))
c, t, d, y = make_inputs(1, 100000, 2, 3, 47)
args = (0.01,) + tuple(x[0] for x in c) + (e, e2, e2, t[0], y[0], d[0])
s = celerite_amd.CholeskySolver()
for _ in range(5):
    v, g = s.grad_log_likelihood(*args)
print("value", v, "max partial", float(np.max(np.abs(g))))
