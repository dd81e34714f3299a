# -*- coding: utf-8 -*-
"""Create / use / destroy soak: many plans and solver objects in one process (leaks, stale state, stream reuse);
prints the device memory in use after 10 and after 120 iterations (clr_device_memory)."""
import os, sys, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd
from celerite_amd import batch, terms, GP
from bench import make_inputs

def used():
    free, total = batch.device_memory()
    return (total - free) / 2**20

rng = np.random.RandomState(0)
u0 = None
for it in range(120):
    B, N = int(rng.choice([1, 7, 64, 300])), int(rng.choice([50, 1000, 5000, 20000]))
    JR, JC = [(1, 0), (2, 3), (0, 2), (1, 3), (3, 5), (0, 16)][it % 6]
    coeffs, t, diag, y = make_inputs(B, N, JR, JC, seed=it)
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
    ll, ld, q, st = plan.log_likelihood()
    assert np.all(np.isfinite(ld[st == 0]))
    if JR + 2 * JC <= 8 and it % 3 == 0:
        sp = batch.ShardedBatchedGP(B, N, JR, JC, devices=[0, 0, 0][: min(3, B)])
        sp.set_series(t, diag, y); sp.set_coefficients(*coeffs)
        ll2, ld2, q2, st2 = sp.log_likelihood()
        assert np.array_equal(ld2, ld) or np.allclose(ld2, ld, rtol=1e-13)
        sp.close()
    plan.close()
    k = terms.RealTerm(0.1, -1.0) + terms.ComplexTerm(-0.5, -1.0, 0.3)
    gp = GP(k); gp.compute(t[0], np.sqrt(diag[0])); v = gp.log_likelihood(y[0]); p = gp.predict(y[0], t[0][:50], return_cov=False)
    s = celerite_amd.solver.CARMASolver(-0.5, np.array([0.1, 0.05, 0.01]), np.array([0.2, 0.1]))
    s.log_likelihood(t[0][:200], y[0][:200], np.sqrt(diag[0][:200]))
    del gp, s
    if it == 10:
        gc.collect(); batch.device_synchronize(); u0 = used()
gc.collect(); batch.device_synchronize()
print("device memory in use: after 10 iterations %.1f MiB, after 120 iterations %.1f MiB" % (u0, used()))
