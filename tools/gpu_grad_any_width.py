"""grad_log_likelihood above width 64 (csrc/grad_any_kernels.hip: one workgroup per partial, S and its tangent in an
HBM / L2 workspace): wall time of CholeskySolver.grad_log_likelihood per call and per sample, next to the wave-per-partial
kernel at width 64 and to the any-width kernel forced at width 64.  Writes gpurun_out/r06za_grad_any_width.txt."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import celerite_amd  # noqa: E402
from celerite_amd import batch  # noqa: E402

NO_GENERAL = (np.empty(0), np.empty((0, 0)), np.empty((0, 0)))
rng = np.random.RandomState(1)
s = celerite_amd.CholeskySolver()
lines = ["width  partials  N      ms/call   us/sample   kernel"]
for JR, JC, N, forced in ((0, 32, 4000, False), (0, 32, 4000, True), (2, 34, 4000, False), (0, 64, 4000, False), (0, 64, 20000, False),
                          (0, 128, 2000, False), (0, 256, 1000, False), (0, 512, 300, False)):
    x = np.sort(rng.uniform(0, 30, N))
    diag = rng.uniform(0.1, 0.3, N)
    y = rng.randn(N)
    co = (np.exp(rng.uniform(-1, 1, JR)), np.exp(rng.uniform(-2, 0, JR)), np.exp(rng.uniform(-1, 0, JC)),
          0.2 * rng.rand(JC), np.exp(rng.uniform(-2, 0, JC)), np.exp(rng.uniform(-1, 1.5, JC)))
    args = (0.1,) + co + NO_GENERAL + (x, y, diag)

    def call():
        if forced or JR + 2 * JC <= 64:
            with batch.option("CLR_GRAD_SEQUENTIAL"):
                if forced:
                    with batch.option("CLR_GRAD_ANY_WIDTH"):
                        return s.grad_log_likelihood(*args)
                return s.grad_log_likelihood(*args)
        return s.grad_log_likelihood(*args)

    call()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        v, g = call()
    ms = (time.perf_counter() - t0) / reps * 1e3
    kern = "workgroup per partial" if (forced or JR + 2 * JC > 64) else "wave per partial"
    lines.append("%5d  %8d  %-6d %8.2f  %9.2f   %s" % (JR + 2 * JC, len(g), N, ms, ms * 1e3 / N, kern))
    print(lines[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/r06za_grad_any_width.txt", "w").write("\n".join(lines) + "\n")
