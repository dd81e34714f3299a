"""A/B of two BUILDS on BASELINE config 4 (B=256, N=1e5, width 32): see tools/gpu_ab_builds.py for the protocol."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
batch.LIB_PATH = os.environ["CLR_LIB"]
from bench import make_inputs
coeffs, t, diag, y = make_inputs(256, 100000, 0, 16, 11, d_spread=True)
plan = batch.BatchedGP(256, 100000, 0, 16)
plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
plan.enqueue(); plan.synchronize()
tot, k = plan.run_timed(3)
ll, ld, q, st = plan.results()
print(os.environ["CLR_LIB"], "ms/step %.2f" % (tot / 3), {a: round(b / 3, 2) for a, b in k.items()}, "checksum %.12e" % float(np.sum(ld)), flush=True)
