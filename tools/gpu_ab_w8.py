"""A/B of two BUILDS (CLR_LIB) on the width-8 kernels OTHER than the headline summarize: prefix / correct of the headline
step, the materialising replay (both layouts), the warm-started recurrence (accuracy family), the reverse-mode gradient,
the batched solve and dot_L.  One line per quantity; checksums to compare builds."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from celerite_amd import batch
batch.LIB_PATH = os.environ["CLR_LIB"]
from bench import make_inputs
tag = os.path.basename(os.environ["CLR_LIB"])
B, N = 1024, 100000
coeffs, t, diag, y = make_inputs(B, N, 2, 3, seed=42)
plan = batch.BatchedGP(B, N, 2, 3)
plan.set_series(t, diag, y)
plan.set_coefficients(*coeffs)
plan.enqueue(); plan.synchronize()
tot, k = plan.run_timed(20)
ld = plan.results()[1]
print(tag, "step", {a: round(b / 20, 4) for a, b in k.items()}, "checksum %.12e" % float(np.sum(ld)), flush=True)
for layout in ("reference", "lean"):
    plan.set_factor_layout(layout)
    plan.enqueue(materialize=True); plan.synchronize()
    tot, k = plan.run_timed(5, materialize=True, relayout_each_step=False)
    plan.solve(); plan.solve(); ms_solve = plan.solve_device_ms()
    x = plan.dot_L(y); ms_dotl = plan.solve_device_ms()
    print(tag, "materialize", layout, "step %.3f replay %.3f" % (tot / 5, k["replay"] / 5), "solve %.3f dot_L %.3f" % (ms_solve, ms_dotl),
          "checksum %.12e" % float(np.sum(x[::97, ::101])), flush=True)
plan.set_factor_layout("reference")
t0 = time.perf_counter()
for _ in range(3):
    v, g, st = plan.grad_log_likelihood()
print(tag, "gradient %.3f ms" % ((time.perf_counter() - t0) / 3 * 1e3), "checksum %.12e" % float(np.sum(g)), flush=True)
plan.close()
from bench import make_inputs_accuracy
coeffs, t, diag, y = make_inputs_accuracy(B, N, 2, 3, 42)
if True:
    plan = batch.BatchedGP(B, N, 2, 3)
    plan.set_series(t, diag, y)
    plan.set_coefficients(*coeffs)
    plan.enqueue(); plan.synchronize()
    tot, k = plan.run_timed(20)
    print(tag, "accuracy family step %.3f" % (tot / 20), plan.warm_start().get("active"), "checksum %.12e" % float(np.sum(plan.results()[1])), flush=True)
    plan.close()
