# -*- coding: utf-8 -*-
"""Round 3 A/B on one MI355X: (a) prefix phase, plain walk against multi-level plans, on the headline shape and on
BASELINE config 1 (chunk count x plan); (b) the materialising replay reading the chunk-interleaved copy against the
LDS-staged row-major arrays.  Prints one table per experiment (per-kernel HIP-event ms per step)."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celerite_amd import batch
import bench

def table(title, rows, cols):
    print("\n## " + title)
    print(" | ".join(["%-28s" % "variant"] + ["%9s" % c for c in cols]))
    for name, vals in rows:
        print(" | ".join(["%-28s" % name] + ["%9.4f" % v for v in vals]))
    sys.stdout.flush()

def prefix_ab(B, N, JR, JC, chunk_list, plans, steps, seed):
    coeffs, t, diag, y = bench.make_inputs(B, N, JR, JC, seed)
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_series(t, diag, y)
    plan.set_coefficients(*coeffs)
    rows = []
    for nchunk in chunk_list:
        if nchunk:
            plan.set_chunks(nchunk)
        plan.set_prefix_mode("walk")
        plan.enqueue(); plan.synchronize()
        ref = plan.results()
        ms, k = plan.run_timed(steps, relayout_each_step=False)
        rows.append(("chunks %d walk" % plan.chunks[0], [ms / steps, k["summarize"] / steps, k["prefix"] / steps, k["correct"] / steps, 0.0, 0.0]))
        plan.set_prefix_mode("multilevel")
        for lv, g in plans:
            plan.set_prefix_plan(lv, g)
            got = plan.prefix_plan
            if got[0] == 0:
                continue
            plan.enqueue(); plan.synchronize()
            out = plan.results()
            ms, k = plan.run_timed(steps, relayout_each_step=False)
            e1 = float(np.max(np.abs(out[1] - ref[1]) / np.abs(ref[1])))
            e2 = float(np.max(np.abs(out[2] - ref[2]) / np.abs(ref[2])))
            rows.append(("chunks %d plan %s%s" % (plan.chunks[0], got[1], " (auto)" if lv < 0 else ""),
                         [ms / steps, k["summarize"] / steps, k["prefix"] / steps, k["correct"] / steps, e1 * 1e12, e2 * 1e12]))
        plan.set_prefix_plan(-1, 0)
    plan.close()
    return rows

which = sys.argv[1:] or ["headline", "config1", "materialize"]
if "headline" in which:
    rows = prefix_ab(1024, 100000, 2, 3, [0], [(-1, 0), (1, 3), (1, 4), (1, 5), (1, 6), (1, 8), (2, 2), (2, 3), (2, 4), (3, 2)], 10, 42)
    table("headline B=1024 N=1e5 (2,3): ms per step", rows, ["step", "summarize", "prefix", "correct", "dld e-12", "dq e-12"])
if "config1" in which:
    rows = prefix_ab(256, 10000, 0, 2, [0, 125, 192, 256, 384, 512], [(-1, 0), (1, 4), (1, 6), (1, 8), (1, 12), (2, 3), (2, 4), (2, 5)], 20, 7)
    table("config 1 B=256 N=1e4 (0,2): ms per step", rows, ["step", "summarize", "prefix", "correct", "dld e-12", "dq e-12"])
if "materialize" in which:
    B, N, JR, JC = 1024, 100000, 2, 3
    coeffs, t, diag, y = bench.make_inputs(B, N, JR, JC, 42)
    plan = batch.BatchedGP(B, N, JR, JC)
    plan.set_series(t, diag, y)
    plan.set_coefficients(*coeffs)
    rows = []
    fb = B * 8.0 * N * (3 * 8 + 1) + B * 24.0 * N
    for rep in range(2):
        for src in (0, 1):
            plan.set_replay_source(src)
            plan.enqueue(materialize=True); plan.synchronize()
            ms, k = plan.run_timed(4, materialize=True, relayout_each_step=False)
            rows.append(("replay source %d (%s) #%d" % (src, ("interleaved copy", "staged row-major")[src], rep),
                         [ms / 4, k["summarize"] / 4, k["replay"] / 4, fb / (k["replay"] / 4 * 1e-3) / 1e12, fb / (k["replay"] / 4 * 1e-3) / 8e12]))
    plan.close()
    table("materialising step B=1024 N=1e5 (2,3)", rows, ["step ms", "summarize", "replay ms", "TB/s", "frac HBM"])
