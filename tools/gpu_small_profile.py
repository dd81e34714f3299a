"""BASELINE configs[1] (B=256, N=1e4, width 4) for the profiler: the one-launch evaluation and the scan pipeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from celerite_amd import batch
coeffs, t, diag, y = make_inputs(256, 10000, 0, 2, 7)
for mode in (1, 0):
    plan = batch.BatchedGP(256, 10000, 0, 2)
    plan.set_small_mode(mode)
    plan.set_series(t, diag, y)
    plan.set_coefficients(*coeffs)
    for _ in range(6):
        plan.enqueue()
    ll, ld, q, st = plan.results()
    print("small mode", mode, "status ok", int((st == 0).sum()), "ll[0]", ll[0])
    plan.close()
