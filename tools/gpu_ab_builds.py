"""A/B of two BUILDS of libcelerite_hip.so on one box (box-to-box variation is 5-10 %): copy each build to
variants/lib<X>.so, then `for v in A B A B; do CLR_LIB=$PWD/variants/lib$v.so python tools/gpu_ab_builds.py; done`
inside ONE gpurun call.  Headline shape, summarize kernel time + a checksum of the log-determinants."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
batch.LIB_PATH = os.environ["CLR_LIB"]
from bench import make_inputs
B, N = 1024, 100000
coeffs, t, diag, y = make_inputs(B, N, 2, 3, seed=42)
plan = batch.BatchedGP(B, N, 2, 3)
plan.set_series(t, diag, y)
plan.set_coefficients(*coeffs)
plan.enqueue(); plan.synchronize()
for _ in range(2):
    tot, k = plan.run_timed(20)
    ll, ld, q, st = plan.results()
    print(os.environ["CLR_LIB"], "summarize_ms %.3f" % (k["summarize"] / 20), "checksum %.12e" % float(np.sum(ld)), flush=True)
