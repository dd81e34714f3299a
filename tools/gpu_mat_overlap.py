"""Materialising step (summarize, VALU-bound + replay, HBM-bound): does cutting the batch into G plans on G streams,
free-running side by side, overlap one plan's summarize with another's replay?  Wall clock of `steps` materialising
evaluations of 1024 problems: one plan against G = 2, 4 plans of 1024 / G problems driven by G host threads."""
import sys, os, time, threading
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
from bench import make_inputs
B, N, steps = 1024, 100000, 10
coeffs, t, diag, y = make_inputs(B, N, 2, 3, seed=42)

def build(lo, hi, nchunk=0):
    p = batch.BatchedGP(hi - lo, N, 2, 3)
    if nchunk:
        p.set_chunks(nchunk)
    p.set_series(t[lo:hi], diag[lo:hi], y[lo:hi])
    p.set_coefficients(*[c[lo:hi] for c in coeffs])
    p.enqueue(materialize=True); p.synchronize()
    return p

for G, nchunk, stagger in ((1, 0, 0), (2, 0, 0), (2, 64, 0), (4, 64, 0), (2, 64, 1), (4, 64, 1), (1, 0, 0)):
    plans = [build(g * B // G, (g + 1) * B // G, nchunk) for g in range(G)]
    out = [None] * G
    def work(g):
        if stagger and g:  # start plan g one summarize-time late: a plain evaluation first
            plans[g].run_timed(g, materialize=False, relayout_each_step=False)
        out[g] = plans[g].run_timed(steps, materialize=True, relayout_each_step=False)
    for p in plans:
        p.synchronize()
    th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    wall = (time.perf_counter() - t0) * 1e3 / steps
    per = [round(o[0] / steps, 2) for o in out]
    k0 = {a: round(b / steps, 2) for a, b in out[0][1].items()}
    print("G=%d nchunk=%s stagger=%d: wall %.2f ms per 1024 problems; per-plan step ms %s; plan 0 kernels %s chunks %s"
          % (G, nchunk or "auto", stagger, wall, per, k0, plans[0].chunks), flush=True)
    for p in plans:
        p.close()
