"""General terms in the chunked wide scan (B=64, N=12000, (2,3)+4) for the profiler."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _cases import synthetic, coeffs_of
from celerite_amd import batch
B, N, JR, JC, JG = 64, 12000, 2, 3, 4
case = synthetic(B, N, JR, JC, "bench", seed=9)
t = case["t"]
z = (t - t.mean(axis=1, keepdims=True)) / (t.max(axis=1, keepdims=True) - t.min(axis=1, keepdims=True))
U = np.stack([np.vander(zz, JG).T for zz in z]); V = U * np.random.RandomState(1).rand(B, JG)[:, :, None]
A = np.sum(U * V, axis=1) + 1e-8
plan = batch.BatchedGP(B, N, JR, JC)
plan.set_series(case["t"], case["diag"], case["y"]); plan.set_coefficients(*coeffs_of(case), jitter=0.01)
plan.set_general(A, U, V)
for _ in range(6):
    plan.enqueue()
ll, ld, q, st = plan.results()
print("status ok", int((st == 0).sum()), "ll[0]", ll[0])
