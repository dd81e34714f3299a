"""A/B of two BUILDS on BASELINE config 4 (B=256, N=1e5, width 32): see tools/gpu_ab_builds.py for the protocol.
Prints the per-kernel times of the real loop, the routes taken and the deviation from the CPU oracle on 8 problems."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
batch.LIB_PATH = os.environ["CLR_LIB"]
from bench import make_inputs
from oracle import ref
JC = int(os.environ.get("CLR_AB_JC", "16"))   # 16: BASELINE configs[4] (width 32); 32: the extra width-64 workload
JR = int(os.environ.get("CLR_AB_JR", "0"))
coeffs, t, diag, y = make_inputs(256, 100000, JR, JC, 11, d_spread=True)
plan = batch.BatchedGP(256, 100000, JR, JC)
if os.environ.get("CLR_AB_CHUNKS"):
    plan.set_chunks(int(os.environ["CLR_AB_CHUNKS"]))
plan.set_series(t, diag, y); plan.set_coefficients(*coeffs)
plan.enqueue(); plan.synchronize()
tot, k = plan.run_timed(3)
ll, ld, q, st = plan.results()
idx = np.arange(0, 256, 32)
ar, cr, ac, bc, cc, dc = coeffs
ll0, ld0, q0, st0 = ref.batch_log_likelihood(0.0, ar[idx], cr[idx], ac[idx], bc[idx], cc[idx], dc[idx], t[idx], diag[idx], y[idx])
eld = float(np.max(np.abs(ld[idx] - ld0) / np.abs(ld0))); eq = float(np.max(np.abs(q[idx] - q0) / np.abs(q0)))
print(os.path.basename(os.environ["CLR_LIB"]), "ms/step %.2f" % (tot / 3), {a: round(b / 3, 2) for a, b in k.items()},
      "routes", np.bincount(plan.exact_flags(), minlength=3).tolist(), "status", int((st != 0).sum()),
      "vs oracle logdet %.1e quad %.1e" % (eld, eq), "checksum %.12e" % float(np.sum(ld)), "chunks", plan.chunks,
      "gamma x eG max %.2e" % float(np.max(plan.conditioning()[0] * plan.measured_error())), flush=True)
