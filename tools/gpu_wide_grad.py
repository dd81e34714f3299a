"""The plan gradient at widths 9..32 (chunk-parallel forward mode, wide_grad_kernels.hip) against the sequential tangent
kernel: deviation and time."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from celerite_amd import batch
from _cases import synthetic, coeffs_of
for (B, N, JR, JC, nchunk) in ((3, 6000, 4, 4, 6), (2, 20000, 0, 8, 16), (1, 100000, 0, 8, 16), (1, 100000, 0, 8, 32), (1, 100000, 0, 8, 48), (1, 100000, 2, 5, 32),
                               (1, 100000, 0, 16, 16), (1, 100000, 0, 16, 24), (8, 30000, 0, 16, 8)):
    case = synthetic(B, N, JR, JC, "bench", seed=JR + JC)
    jit = np.linspace(0.0, 0.05, B)
    os.environ["CLR_GRAD_SEQUENTIAL"] = "1"
    t0 = time.perf_counter()
    v0, g0, s0 = batch.batch_grad_log_likelihood(*coeffs_of(case), case["t"], case["diag"], case["y"], jitter=jit)
    t_seq = time.perf_counter() - t0
    del os.environ["CLR_GRAD_SEQUENTIAL"]
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_chunks(nchunk)
    plan.set_series(case["t"], case["diag"], case["y"])
    plan.set_coefficients(*coeffs_of(case), jitter=jit)
    v, g, st = plan.grad_log_likelihood()
    t0 = time.perf_counter()
    for _ in range(3):
        v, g, st = plan.grad_log_likelihood()
    t_par = (time.perf_counter() - t0) / 3
    scale = np.max(np.abs(g0), axis=1, keepdims=True)
    print("B=%d N=%d (%d,%d) chunks %s: value %.2e partials %.2e of the largest; status %s/%s; sequential %.1f ms (incl. transfers), plan %.2f ms, fallbacks %d"
          % (B, N, JR, JC, plan.chunks, np.max(np.abs(v - v0) / np.abs(v0)), np.max(np.abs(g - g0) / scale), st.tolist(), s0.tolist(), t_seq * 1e3, t_par * 1e3, plan.grad_fallbacks()), flush=True)
    plan.close()
