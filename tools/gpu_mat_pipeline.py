"""Materialising step at the headline shape (1024 x 1e5 x width 8): the plain sequence summarize -> prefix -> correct ->
replay against the pipeline over groups of problems (clr_batch_set_materialize_pipeline): groups x CUs of the summarize
streams x summarize streams x chunk count, plus the single-wave summarize co-resident with the replay (no masks).
Prints ms per materialising step (HIP events, first to last) and the whole-step fraction of the 8 TB/s roofline for the
reference layout's bytes (factor written + two passes over the series)."""
import sys, os, itertools
import numpy as np
sys.path.insert(0, os.getcwd())
from celerite_amd import batch
from bench import make_inputs

B, N, JR, JC = 1024, 100000, 2, 3
W = JR + 2 * JC
steps = int(os.environ.get("STEPS", "8"))
quick = os.environ.get("QUICK") == "1"
coeffs, t, diag, y = make_inputs(B, N, JR, JC, seed=42)
bytes_step = B * (8.0 * N * (3 * W + 1) + 2 * 24.0 * N)

plan = batch.BatchedGP(B, N, JR, JC)
plan.set_series(t, diag, y)
plan.set_coefficients(*coeffs)


def run(label):
    plan.enqueue(materialize=True); plan.synchronize()
    runs = []
    for _ in range(3):
        ms, k = plan.run_timed(steps, materialize=True, relayout_each_step=False)
        runs.append(ms / steps)
    med = sorted(runs)[1]
    print("%-64s %6.2f ms (runs %s)  whole-step frac %.3f  chunks %s  kernels %s" % (
        label, med, " ".join("%.2f" % r for r in runs), bytes_step / (med * 1e-3) / 8e12, plan.chunks,
        {a: round(b / steps, 2) for a, b in k.items() if b / steps > 0.005}), flush=True)
    return med


ref_out = {}
for nchunk in (0, 128, 256, 512):
    if nchunk:
        plan.set_chunks(nchunk)
        plan.set_coefficients(*coeffs)
    run("plain sequence, chunks %s" % (nchunk or "auto"))
    res = plan.results()
    ref_out[plan.chunks[0]] = (res, [plan.factor(p) for p in (0, 517, 1023)])

print("census: plan stream", plan.cu_census(0))
plan.set_materialize_pipeline(4, 144, 1)
print("census: 144-CU summarize stream", plan.cu_census(1), " replay stream", plan.cu_census(2))
plan.set_materialize_pipeline(4, 128, 1)
print("census: 128-CU summarize stream", plan.cu_census(1), " replay stream", plan.cu_census(2))

best = (1e9, None)
grid = [(nc, G, X, nS) for nc in (256, 512, 128) for G in (4, 8, 16) for X in (0, 128, 144, 160) for nS in (1, 2)]
if quick:
    grid = [(256, 8, 144, 2), (256, 8, 0, 2)]
for nc, G, X, nS in grid:
    plan.set_chunks(nc)
    plan.set_coefficients(*coeffs)
    plan.set_materialize_pipeline(G, X, nS)
    try:
        med = run("pipeline chunks %d groups %d summarize CUs %d streams %d" % (nc, G, X, nS))
    except Exception as e:
        print("FAILED", nc, G, X, nS, e, flush=True)
        continue
    if med < best[0]:
        best = (med, (nc, G, X, nS))
print("best pipeline:", best)

# bit-identity of the best pipeline against the plain sequence at the same chunking: results and factors
nc, G, X, nS = best[1]
plan.set_chunks(nc); plan.set_coefficients(*coeffs); plan.set_materialize_pipeline(G, X, nS)
plan.enqueue(materialize=True); plan.synchronize()
res = plan.results()
want_res, want_fac = ref_out[plan.chunks[0]]
same = all(np.array_equal(a, b) for a, b in zip(res, want_res))
for p, wf in zip((0, 517, 1023), want_fac):
    same = same and all(np.array_equal(a, b) for a, b in zip(plan.factor(p), wf))
print("pipeline == plain sequence, results and factors of problems 0, 517, 1023 bit for bit:", same)

# the judge's variant: the single-wave summarize (216 registers, 27 KB of LDS) co-resident with the replay, no masks
plan.set_summarize_mode(0)
for nc, G in ((64, 0), (64, 2), (64, 4), (128, 4), (256, 8)):
    plan.set_chunks(nc); plan.set_coefficients(*coeffs); plan.set_materialize_pipeline(G, 0, 2 if G else 1)
    run("single-wave summarize, chunks %d, groups %d (no masks)" % (nc, G))
plan.close()
