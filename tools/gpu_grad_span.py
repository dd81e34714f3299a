"""Reverse-mode plan gradient (B = 1024, N = 1e5, width 8) on both input families by CLR_GRAD_REBUILD_SPAN: stored states at
least `span` steps apart, the ones in between rebuilt forwards by the sweep (GradStore::span).  Partials against span = 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_inputs, make_inputs_accuracy
from celerite_amd import batch

B, N = 1024, 100000
for name, maker, spans in (("accuracy family", make_inputs_accuracy, (1, 2, 3, 4, 6, 8)), ("bench family", make_inputs, (1, 4))):
    coeffs, t, diag, y = maker(B, N, 2, 3, 42)
    plan = batch.BatchedGP(B, N, 2, 3)
    plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
    ref = None
    for span in spans:
        os.environ["CLR_GRAD_REBUILD_SPAN"] = str(span)
        v, g, st = plan.grad_log_likelihood()
        batch.device_synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            v, g, st = plan.grad_log_likelihood()
        dt = (time.perf_counter() - t0) / 3
        if ref is None: ref = g
        dev = np.max(np.abs(g - ref) / np.max(np.abs(ref), axis=1, keepdims=True))
        print("%s span %d: %.2f ms per call, ok %d, vs span 1 %.1e of the largest partial, %s" % (name, span, dt * 1e3, int((st == 0).sum()), dev, plan.grad_info()), flush=True)
    plan.close()
del os.environ["CLR_GRAD_REBUILD_SPAN"]
