"""grad_log_likelihood at widths 33..64 through CholeskySolver: the chunk-wise tangents at the padded width 64 (round 6:
wide_grad_riders64_kernel + wide_grad_kernel<64, ., CHUNKED> + wide_grad_walk_kernel<64>) against the sequential tangent
kernel (CLR_GRAD_SEQUENTIAL).  Writes gpurun_out/r06zd_wide64_grad.txt."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd  # noqa: E402
from celerite_amd import batch  # noqa: E402

NO_GENERAL = (np.empty(0), np.empty((0, 0)), np.empty((0, 0)))
rng = np.random.RandomState(1)
lines = ["width  partials  N        sequential ms   parallel in n ms   ratio   max |dg| / max |g|"]
for JR, JC, N in ((2, 16, 20000), (2, 16, 100000), (0, 24, 100000), (0, 32, 20000), (0, 32, 100000), (0, 32, 400000)):
    x = np.sort(rng.uniform(0, 0.05 * N, N))
    diag = rng.uniform(0.1, 0.3, N)
    y = rng.randn(N)
    co = (np.exp(rng.uniform(-1, 1, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0, JC)),
          0.2 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(-1, 1.5, JC)))
    args = (0.1,) + co + NO_GENERAL + (x, y, diag)
    s = celerite_amd.CholeskySolver()
    out = {}
    for mode in ("seq", "par"):
        def call():
            if mode == "seq":
                with batch.option("CLR_GRAD_SEQUENTIAL"):
                    return s.grad_log_likelihood(*args)
            return s.grad_log_likelihood(*args)
        call()
        reps = 2 if mode == "seq" else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            v, g = call()
        out[mode] = ((time.perf_counter() - t0) / reps * 1e3, v, g)
    dev = np.max(np.abs(out["par"][2] - out["seq"][2])) / np.max(np.abs(out["seq"][2]))
    lines.append("%5d  %8d  %-7d  %13.1f   %16.1f   %5.1f   %.1e" % (JR + 2 * JC, len(g), N, out["seq"][0], out["par"][0],
                                                                     out["seq"][0] / out["par"][0], dev))
    print(lines[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/r06zd_wide64_grad.txt", "w").write("\n".join(lines) + "\n")
