"""As gpu_ab_builds.py for BASELINE config 4 (width 32) and the accuracy family: kernel times + checksums of two builds."""
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
tot, k = plan.run_timed(4)
ll, ld, q, st = plan.results()
print(os.environ["CLR_LIB"], "config4 chunks", plan.chunks, "step_ms %.3f summarize_ms %.3f" % (tot / 4, k["summarize"] / 4), "checksum %.12e %.12e" % (float(np.sum(ld)), float(np.sum(q))), flush=True)
plan.close()
