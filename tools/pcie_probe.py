import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from celerite_amd import batch
B, N = 1024, 100000
coeffs, t, diag, y = bench.make_inputs(B, N, 2, 3, seed=42)
batch.batch_log_likelihood(*[c[:8] for c in coeffs], t[:8], diag[:8], y[:8])  # warm the runtime
for rep in range(2):
    t0 = time.perf_counter()
    ll, ld, q, st = batch.batch_log_likelihood(*coeffs, t, diag, y)
    dt = time.perf_counter() - t0
    print("one-shot clr_batch_log_likelihood (create + 2.46 GB H2D from pageable NumPy arrays + evaluate + D2H + destroy): %.1f ms -> %.0f loglik/s" % (dt * 1e3, B / dt))
plan = batch.BatchedGP(B, N, 2, 3)
t0 = time.perf_counter(); plan.set_series(t, diag, y); plan.synchronize(); dt = time.perf_counter() - t0
print("set_series alone: %.1f ms = %.1f GB/s" % (dt * 1e3, 3 * B * N * 8 / dt / 1e9))
plan.close()
