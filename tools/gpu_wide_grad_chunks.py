import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from celerite_amd import batch
if os.environ.get("CLR_LIB"): batch.LIB_PATH = os.environ["CLR_LIB"]; print(os.path.basename(batch.LIB_PATH))
from _cases import synthetic, coeffs_of
for (N, JR, JC) in ((100000, 0, 8), (100000, 0, 16), (100000, 4, 4), (20000, 0, 8), (20000, 0, 16)):
    case = synthetic(1, N, JR, JC, "bench", seed=JR + JC)
    plan = batch.BatchedGP(1, N, JR, JC)
    plan.set_series(case["t"], case["diag"], case["y"])
    plan.set_coefficients(*coeffs_of(case), jitter=0.01)
    row = []
    for nchunk in (8, 16, 24, 32, 48, 64, 96, 128):
        if N // nchunk < 128: continue
        plan.set_chunks(nchunk)
        plan.set_coefficients(*coeffs_of(case), jitter=0.01)
        plan.grad_log_likelihood()
        t0 = time.perf_counter()
        for _ in range(3):
            plan.grad_log_likelihood()
        row.append("%d: %.2f" % (plan.chunks[0], (time.perf_counter() - t0) / 3 * 1e3))
    print("N=%d (%d,%d) ms per gradient by chunk count: %s" % (N, JR, JC, "  ".join(row)), flush=True)
    plan.close()
