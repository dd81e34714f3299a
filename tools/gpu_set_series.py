"""clr_batch_set_series at the headline shape (2.46 GB of pageable NumPy arrays): host time of the first call (pinned
staging created) and of later ones, against a plain hipMemcpy-style upload; results unchanged."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
from bench import make_inputs
B, N = 1024, 100000
coeffs, t, diag, y = make_inputs(B, N, 2, 3, seed=42)
plan = batch.BatchedGP(B, N, 2, 3)
for i in range(4):
    t0 = time.perf_counter()
    plan.set_series(t, diag, y)
    wall = (time.perf_counter() - t0) * 1e3
    print("set_series call %d: C side %.1f ms, Python wall %.1f ms (%.1f GB/s)" % (i, plan.selection_bounds()["set_series_host_ms"], wall, 2.4576 / wall * 1e3), flush=True)
plan.set_coefficients(*coeffs)
ll, ld, q, st = plan.log_likelihood()
print("checksum %.12e status_not_ok %d" % (float(np.sum(ld)), int((st != 0).sum())))
sp = batch.ShardedBatchedGP(B, N, 2, 3, devices=[0, 0])
for i in range(2):
    t0 = time.perf_counter()
    sp.set_series(t, diag, y)
    print("sharded (2 shards) set_series call %d: wall %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
