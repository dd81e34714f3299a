"""One-launch evaluation of short narrow problems (small_batch_kernel) against the scan pipeline: BASELINE configs[1]
(256 x 1e4 x width 4) and neighbours; device-only step, real loop (coefficients in, results out), parity vs the oracle."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
from bench import make_inputs, fresh_draws, real_loop
from oracle import ref
for (B, N, JR, JC) in ((256, 10000, 0, 2), (256, 10000, 2, 1), (1024, 10000, 0, 2), (4096, 10000, 0, 2), (256, 2000, 1, 1), (64, 30000, 0, 2), (2048, 2000, 0, 2)):
    coeffs, t, diag, y = make_inputs(B, N, JR, JC, seed=7)
    draws = [coeffs] + fresh_draws(coeffs, 3, seed=8)
    S = min(B, 64)
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *[c[:S] for c in coeffs], t[:S], diag[:S], y[:S], nthreads=os.cpu_count())
    row = []
    for mode in (0, 1):
        plan = batch.BatchedGP(B, N, JR, JC)
        plan.set_small_mode(mode)
        plan.set_series(t, diag, y)
        plan.set_coefficients(*coeffs)
        ll, ld, q, st = plan.log_likelihood()
        active = plan.small_mode_active()
        err = max(np.max(np.abs(ld[:S] - d0) / np.abs(d0)), np.max(np.abs(q[:S] - q0) / np.abs(q0)))
        real_loop(plan, draws, 20)
        t0 = time.perf_counter(); real_loop(plan, draws, 100, offset=1); batch.device_synchronize()
        loop_ms = (time.perf_counter() - t0) * 10
        plan.set_coefficients(*coeffs); plan.enqueue(); plan.synchronize()
        dev_ms, k = plan.run_timed(50, relayout_each_step=False)
        row.append("mode %d active %s: real loop %.3f ms, device %.3f ms (%s) err %.1e status %d" % (
            mode, active, loop_ms, dev_ms / 50, " ".join("%s %.3f" % (a, b / 50) for a, b in k.items() if b / 50 > 0.002), err, int((st != s0[0]).sum() if False else (st != 0).sum())))
        plan.close()
    print("B=%d N=%d shape (%d,%d):\n   %s\n   %s" % (B, N, JR, JC, row[0], row[1]), flush=True)
