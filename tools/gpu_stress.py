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
WIDE = [(9, 0), (1, 4), (3, 5), (16, 0), (0, 8), (2, 9), (0, 16), (6, 13), (32, 0)]
SHAPES = ALL_WIDTH_SHAPES + WIDE + WIDE   # narrow scan (widths 1..8) and wide scan (9..32)
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 600):
    JR, JC = SHAPES[trial % len(SHAPES)]
    N = (50, 200, 1000, 3000)[trial % 4]
    case = adversarial(4, N, JR, JC, seed=5000 + trial)
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
    n_bad += int((s0 != 0).sum())
    plan = batch.BatchedGP(4, N, JR, JC)
    plan.set_series(case["t"], case["diag"], case["y"])
    plan.set_coefficients(*coeffs_of(case))
    for nchunk in (max(2, N // 40), max(2, N // 8)):
        plan.set_chunks(nchunk)
        ll, ld, q, st = plan.log_likelihood()
        n_total += 4
        n_exact += plan.exact_count()
        if not np.array_equal(st, s0):
            mism += 1
            print("STATUS MISMATCH trial", trial, JR, JC, N, nchunk, st, s0, flush=True)
        ok = (s0 == 0) & (st == 0) & np.isfinite(d0) & np.isfinite(q0) & (plan.exact_count() == 0)
        if ok.any():
            worst_ok = max(worst_ok, np.max(np.abs(ld[ok] - d0[ok]) / (1 + np.abs(d0[ok]))))
    plan.close()
print("problem x chunking combinations %d, settled without replay %d, indefinite problems %d, status mismatches %d, "
      "worst log-det deviation on replay-free batches %.2e" % (n_total, n_total - n_exact, n_bad, mism, worst_ok), flush=True)
