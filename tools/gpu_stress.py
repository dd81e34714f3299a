# -*- coding: utf-8 -*-
"""Adversarial sweep on the GPU: near-singular and indefinite problems (tests/_cases.adversarial)
through the batched path; the status word must equal the oracle's for every problem, whichever
route (replay-free or exact replay) settles it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from celerite_amd import batch  # noqa: E402
from oracle import ref  # noqa: E402
from _cases import adversarial, coeffs_of, ALL_WIDTH_SHAPES  # noqa: E402

n_total = n_exact = n_bad = mism = 0
worst_ok = 0.0
n_level = [0, 0, 0]
worst_level = [0.0, 0.0, 0.0]
WIDE = [(9, 0), (1, 4), (3, 5), (16, 0), (0, 8), (2, 9), (0, 16), (6, 13), (32, 0)]
SHAPES = (WIDE if os.environ.get("WIDE_ONLY") else ALL_WIDTH_SHAPES + WIDE + WIDE)   # narrow scan (widths 1..8) and wide scan (9..32)
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 600):
    JR, JC = SHAPES[trial % len(SHAPES)]
    N = (50, 200, 1000, 3000)[trial % 4]
    case = adversarial(4, N, JR, JC, seed=5000 + trial)
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
    n_bad += int((s0 != 0).sum())
    plan = batch.BatchedGP(4, N, JR, JC)
    plan.set_series(case["t"], case["diag"], case["y"])
    plan.set_coefficients(*coeffs_of(case))
    if os.environ.get("CERT"):
        plan.set_certificate(float(os.environ["CERT"]), 1e-11)
    for nchunk in (max(2, N // 40), max(2, N // 8)):
        plan.set_chunks(nchunk)
        ll, ld, q, st = plan.log_likelihood()
        n_total += 4
        n_exact += plan.exact_count()
        if not np.array_equal(st, s0):
            mism += 1
            print("STATUS MISMATCH trial", trial, JR, JC, N, nchunk, st, s0, flush=True)
        levels = plan.exact_levels() if JR + 2 * JC <= 8 else np.where(plan.exact_flags(), 2, 0)
        for p in range(4):
            if s0[p] != 0 or st[p] != 0 or not (np.isfinite(d0[p]) and np.isfinite(q0[p])):
                continue
            dev = max(abs(ld[p] - d0[p]) / abs(d0[p]), abs(q[p] - q0[p]) / abs(q0[p]))
            n_level[levels[p]] += 1
            worst_level[levels[p]] = max(worst_level[levels[p]], dev)
    plan.close()
print("problem x chunking combinations %d, indefinite problems %d, status mismatches %d" % (n_total, n_bad, mism))
print("positive definite problems by route: settled from the chunk summaries %d (worst deviation from the oracle, log-det or "
      "quadratic form, relative: %.2e); checked chunked replay %d (%.2e); sequential recurrence %d (%.2e)"
      % (n_level[0], worst_level[0], n_level[1], worst_level[1], n_level[2], worst_level[2]), flush=True)
