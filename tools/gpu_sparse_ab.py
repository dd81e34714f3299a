"""Summarize kernels on a SPARSE series (the paper's accuracy family: mean spacing 0.8, not eligible for the lazy
kernels): single wave (mode 0) vs role split (mode 1) vs the automatic choice, B=1024, N=1e5."""
import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _cases import synthetic, coeffs_of
from celerite_amd import batch
if os.environ.get("CLR_LIB"):  # A/B of two builds on one box (tools/gpu_ab_builds.py)
    batch.LIB_PATH = os.environ["CLR_LIB"]
B, N = 1024, 100000
SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(2, 3), (4, 2), (0, 4)]
for JR, JC in ([(2, 3)] if os.environ.get("CLR_LIB") else SHAPES):
    case = synthetic(B, N, JR, JC, "accuracy", seed=3)
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_series(case["t"], case["diag"], case["y"]); plan.set_coefficients(*coeffs_of(case))
    ref = None
    for mode in ((1, 1) if os.environ.get("CLR_LIB") else (0, 1, 0, 1)):
        plan.set_summarize_mode(mode)
        plan.enqueue(); plan.synchronize()
        tot, k = plan.run_timed(5, relayout_each_step=False)
        ll, ld, q, st = plan.results()
        if ref is None: ref = (ld.copy(), q.copy())
        print((JR, JC), "mode %2d" % mode, plan.summarize_kernel(), "summarize %.3f ms  step %.3f ms" % (k["summarize"] / 5, tot / 5),
              "vs mode 0: %.1e %.1e" % (np.max(np.abs(ld - ref[0]) / np.abs(ref[0])), np.max(np.abs(q - ref[1]) / np.abs(ref[1]))), flush=True)
    plan.close()
