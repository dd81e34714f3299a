"""Width 64 (B = 256, N = 1e5, 32 complex terms) for the profiler: a few evaluations of the chunked wide scan at the padded width 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
from celerite_amd import batch
coeffs, t, diag, y = make_inputs(256, 100000, 0, 32, 288, d_spread=True)
plan = batch.BatchedGP(256, 100000, 0, 32)
plan.set_series(t, diag, y)
plan.set_coefficients(*coeffs)
for _ in range(3):
    plan.enqueue()
ll, ld, q, st = plan.results()
print("chunks", plan.chunks, "status ok", int((st == 0).sum()), "ll[0]", ll[0])
plan.close()
