"""A/B of two BUILDS (CLR_LIB) on the MATERIALISING step of BASELINE config 4's plan (256 x 1e5 x width 32): the replay
(wide_scan_kernel, MODE 0) that writes the factor, and the batched solve on it."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
batch.LIB_PATH = os.environ["CLR_LIB"]
from bench import make_inputs
coeffs, t, diag, y = make_inputs(256, 100000, 0, 16, 11, d_spread=True)
plan = batch.BatchedGP(256, 100000, 0, 16)
plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
plan.enqueue(materialize=True); plan.synchronize()
tot, k = plan.run_timed(3, materialize=True)
x = plan.solve(); x = plan.solve()
print(os.path.basename(os.environ["CLR_LIB"]), "materialising step %.2f ms" % (tot / 3), {a: round(b / 3, 2) for a, b in k.items() if b / 3 > 0.005},
      "solve %.2f ms" % plan.solve_device_ms(), "checksum %.12e" % float(np.sum(x[::7, ::101])), flush=True)
