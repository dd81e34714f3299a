"""Accuracy family at the headline shape (B = 1024, N = 1e5, width 8): chunk-count sweep of the warm-started
recurrence (device-only time per evaluation; warm-up rows and waves per SIMD trade against each other)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_inputs_accuracy
from celerite_amd import batch
if os.environ.get("CLR_LIB"): batch.LIB_PATH = os.environ["CLR_LIB"]   # A/B of two builds (tools/gpu_ab_builds.py)
coeffs, t, diag, y = make_inputs_accuracy(1024, 100000, 2, 3, 4242)
plan = batch.BatchedGP(1024, 100000, 2, 3)
ref = None
for nc in [int(a) for a in sys.argv[1:]] or (0, 64, 96, 128, 160, 192, 256, 320):
    plan.set_chunks(nc)
    plan.set_series(t, diag, y)
    plan.set_coefficients(*coeffs)
    plan.enqueue(); plan.synchronize()
    ms, k = plan.run_timed(6, relayout_each_step=False)
    ll, ld, q, st = plan.results()
    if ref is None: ref = (ld.copy(), q.copy())
    w = plan.warm_start()
    print("chunks %4d%s warm %s ms/eval %.3f kernels %s ok %d vs auto %.1e %.1e" % (
        nc, " (auto)" if nc == 0 else "", {a: w[a] for a in ("active", "chunks", "chunk_len", "warmup_min", "warmup_max") if a in w}, ms / 6,
        {a: round(b / 6, 3) for a, b in k.items() if b > 0.01}, int((st == 0).sum()),
        np.max(np.abs(ld - ref[0]) / np.abs(ref[0])), np.max(np.abs(q - ref[1]) / np.abs(ref[1]))), flush=True)
plan.close()
