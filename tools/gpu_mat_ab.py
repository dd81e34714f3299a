import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
from bench import make_inputs
B, N, JR, JC = 1024, 100000, 2, 3
W = JR + 2 * JC
steps = 8
coeffs, t, diag, y = make_inputs(B, N, JR, JC, seed=42)
bytes_step = B * (8.0 * N * (3 * W + 1) + 2 * 24.0 * N)
plan = batch.BatchedGP(B, N, JR, JC)
plan.set_series(t, diag, y)
plan.set_coefficients(*coeffs)
for lean in (0, 1, 0, 1):
    plan.set_factor_layout(lean)
    plan.enqueue(materialize=True); plan.synchronize()
    runs = []
    for _ in range(5):
        ms, k = plan.run_timed(steps, materialize=True, relayout_each_step=False)
        runs.append(ms / steps)
    med = sorted(runs)[2]
    print("layout %d: %6.2f ms (runs %s) frac(ref bytes) %.3f kernels %s" % (lean, med, " ".join("%.2f" % r for r in runs),
          bytes_step / (med * 1e-3) / 8e12, {a: round(b / steps, 2) for a, b in k.items() if b / steps > 0.005}), flush=True)
    rhs = np.random.default_rng(1).standard_normal((B, 1, N))
    x = plan.solve(rhs); print("  solve ms", plan.solve_device_ms())
    x = plan.solve(rhs); print("  solve ms", plan.solve_device_ms())
