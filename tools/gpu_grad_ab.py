"""A/B of two BUILDS on the headline gradient (B = 1024, N = 1e5, width 8, reverse mode): ms per call + a checksum."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from celerite_amd import batch
batch.LIB_PATH = os.environ["CLR_LIB"]
from bench import make_inputs
B, N = 1024, 100000
coeffs, t, diag, y = make_inputs(B, N, 2, 3, 42)
plan = batch.BatchedGP(B, N, 2, 3)
plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
v, g, st = plan.grad_log_likelihood()
batch.device_synchronize()
t0 = time.perf_counter()
for _ in range(5):
    v, g, st = plan.grad_log_likelihood()
dt = (time.perf_counter() - t0) / 5
print(os.path.basename(os.environ["CLR_LIB"]), "gradient %.2f ms per call, ok %d, checksum %.12e, info %s" % (dt * 1e3, int((st == 0).sum()), float(np.sum(g)), plan.grad_info()), flush=True)
