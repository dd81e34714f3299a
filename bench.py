# -*- coding: utf-8 -*-
"""Headline benchmark: GP log-likelihoods / second on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2] per GPU; configs[3] = the same on 8 GPUs):
    B = 1024 independent (series x hyper-parameter draw) problems per GPU,
    N = 100000 samples each, width J = 8 (2 real + 3 complex celerite terms,
    the examples/benchmark/run.py:80-84 recipe with 10 % log-parameter scatter),
    fp64, synthetic "bench" family (run.py:66-69): t = sort(U(0,1)), sigma =
    U(0.1,0.2), y = sin t.

One "step" = one evaluation of all B log-likelihoods from inputs resident in
HBM in the public API's row-major layout: summarize -> prefix -> correct ->
replay -> finalize (celerite_amd/csrc).  summarize is the one pass over the
series (read through cooperative LDS-transposed tiles, "staged" layout: nothing is
cached between steps); correct turns every chunk's zero-start sums into its true
log-det / quadratic contributions (determinant lemma + Woodbury), and the replay
kernel -- the reference's recurrence step by step -- only runs for problems with a
chunk that could not be certified (config.problems_replayed; it exits at once for
the others).  config.exact_replay_ms_per_step is the same step with the replay
forced for every problem.  For
information, config.value_fixed_series is the rate with a cached
chunk-interleaved copy of the series (layout "interleaved", relayout outside the
timed region): the optimiser / MCMC case where only hyper-parameters change.

Multi-GPU: the batch axis shards embarrassingly -- one process per GPU, no
collective on the data path; torch.distributed (gloo) is used
only for the rendezvous, the timing barrier and the max-over-ranks reduction.
Weak scaling: every rank evaluates its own B problems.

Rank 0 prints ONE JSON line (see the keys below).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_VALU_TFLOPS = 78.6   # MI355X datasheet: 256 CU x 4 SIMD x 16 FMA lanes x 2 flop x 2.4 GHz
PEAK_HBM_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6300 measured)


def shard_bounds(total, rank, world):
    """Contiguous slice [lo, hi) of a `total`-long batch axis owned by `rank`."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def algorithmic_flops_per_loglik(N, W):
    """F_alg = N (5.5 W^2 + 9.5 W + 3) fp64 flop (fma = 2)."""
    return N * (5.5 * W * W + 9.5 * W + 3.0)


def algorithmic_bytes_per_loglik(N):
    """t, diag, y in + 16 B out."""
    return 24.0 * N + 16.0


def pmc_traffic(kernel, B, N, JR, JC, chunks):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/pmc_latest.json: FETCH_SIZE x 2 + WRITE_SIZE, see the file), if they
    were taken at this configuration; None otherwise (PMC cannot be read live)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        c = rec["config"]
        if (c["batch"], c["N"], c["J_real"], c["J_comp"], c["chunks"]) == (B, N, JR, JC, chunks):
            return rec["traffic_bytes_per_launch"].get(kernel)
    except Exception:
        pass
    return None


def make_inputs(B, N, J_real, J_comp, seed):
    rng = np.random.RandomState(seed)
    t = np.sort(rng.rand(B, N), axis=1)
    sig = rng.uniform(0.1, 0.2, (B, N))
    y = np.sin(t)
    a_real = np.exp(1.0 + 0.1 * rng.randn(B, J_real))
    c_real = np.exp(0.1 + 0.1 * rng.randn(B, J_real))
    a_comp = np.exp(0.1 + 0.1 * rng.randn(B, J_comp))
    b_comp = np.zeros((B, J_comp))
    c_comp = np.exp(2.0 + 0.1 * rng.randn(B, J_comp))
    d_comp = np.exp(1.6 + 0.1 * rng.randn(B, J_comp))
    return (a_real, c_real, a_comp, b_comp, c_comp, d_comp), t, sig ** 2, y


class Dist(object):
    """Rendezvous + barrier + max-reduce; a no-op for a single process."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.pg = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self.pg = dist

    def barrier(self):
        if self.pg:
            self.pg.barrier()

    def max(self, value):
        if not self.pg:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        self.pg.all_reduce(t, op=self.pg.ReduceOp.MAX)
        return float(t[0])

    def close(self):
        if self.pg:
            self.pg.destroy_process_group()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="problems per GPU")
    ap.add_argument("--nsamples", type=int, default=100000)
    ap.add_argument("--jreal", type=int, default=2)
    ap.add_argument("--jcomp", type=int, default=3)
    ap.add_argument("--chunks", type=int, default=0, help="scan chunks per problem (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args(argv)

    dist = Dist()
    if dist.world != args.gpus and dist.world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, dist.world))

    from celerite_amd import batch  # raises if the HIP extension is missing

    if batch.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libcelerite_hip has no CPU path")
    B, N, JR, JC = args.batch, args.nsamples, args.jreal, args.jcomp
    W = JR + 2 * JC

    coeffs, t, diag, y = make_inputs(B, N, JR, JC, seed=42 + dist.rank)
    plan = batch.BatchedGP(B, N, JR, JC, device=dist.local_rank % batch.device_count())
    if args.chunks:
        plan.set_chunks(args.chunks)
    plan.set_series(t, diag, y)          # host -> HBM, outside the timed region
    plan.set_coefficients(*coeffs)

    for _ in range(max(args.warmup, 0)):
        plan.enqueue()
    plan.synchronize()

    # ---- the timed region: exactly K steps between barrier + device sync -------
    dist.barrier()
    batch.device_synchronize()
    t0 = time.perf_counter()
    ev_total_ms, kernel_ms = plan.run_timed(args.steps, relayout_each_step=True)
    batch.device_synchronize()
    dist.barrier()
    dt = dist.max(time.perf_counter() - t0)

    ll, ld, q, st = plan.results()
    replayed = plan.exact_count()
    # the same step with the exact replay forced for every problem (A/B, for information)
    plan.set_exact(True)
    plan.enqueue()
    plan.synchronize()
    exact_total_ms, _ = plan.run_timed(max(args.steps // 2, 1))
    plan.set_exact(False)
    # fixed-series variant (cached interleaved copy), reported for information
    plan.set_layout("interleaved")
    plan.enqueue()
    plan.synchronize()
    fixed_total_ms, _ = plan.run_timed(max(args.steps // 2, 1), relayout_each_step=False)
    fixed_rate = B * max(args.steps // 2, 1) / (fixed_total_ms * 1e-3)
    plan.set_layout("staged")
    # the materialising variant (SURVEY.md 8d "A-ref": the factor phi, u, W, D of every
    # problem written to HBM, as B CholeskySolver.compute calls would) -- HBM-bound
    mat_steps = max(args.steps // 4, 2)
    plan.enqueue(materialize=True)
    plan.synchronize()
    mat_total_ms, mat_kernel_ms = plan.run_timed(mat_steps, materialize=True)

    out = None
    if dist.rank == 0:
        value = dist.world * B * args.steps / dt
        per = {k: v / args.steps for k, v in kernel_ms.items()}
        dom = max(per, key=per.get)
        dom_s = per[dom] * 1e-3
        flops = B * algorithmic_flops_per_loglik(N, W)
        bytes_ = B * algorithmic_bytes_per_loglik(N)
        step_s = sum(per.values()) * 1e-3
        out = {
            "metric": "GP log-likelihoods/sec, N=1e5 J=8 batch=1024; log_det rel-err vs CPU ref",
            "value": value,
            "unit": "log-likelihoods/s",
            "n_gpus": dist.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: batch=%d problems/GPU x N=%d samples, width J=%d "
                            "(%d real + %d complex terms), fp64, fused log-likelihood "
                            "(compute + dot_solve + log_determinant), chunked scan over N, one pass "
                            "over the series" % (B, N, W, JR, JC),
                "batch_per_gpu": B, "N": N, "width": W, "J_real": JR, "J_comp": JC,
                "scan_chunks": plan.chunks[0], "chunk_len": plan.chunks[1],
                "parallelism": "batch-sharded x%d, no collective" % dist.world,
                "series_layout": "staged (row-major arrays read through LDS tiles; no per-step relayout pass)",
                "value_fixed_series": fixed_rate * dist.world,
                "problems_replayed": replayed,
                "exact_replay_ms_per_step": exact_total_ms / max(args.steps // 2, 1),
            },
            "kernels_ms": per,
            "hip_event_ms_per_step": ev_total_ms / args.steps,
            "status_not_ok": int((st != 0).sum()),
            # dominant kernel, SURVEY.md 8(d) op (B): the fused likelihood is bound by the
            # fp64 vector ALU, not by HBM (24 N bytes vs 431 N flop per problem).
            "roofline": {
                "kernel": dom,
                "bound": "fp64_valu",
                "achieved": flops / dom_s / 1e12,
                "peak": PEAK_FP64_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": flops / dom_s / 1e12 / PEAK_FP64_VALU_TFLOPS,
                "traffic": pmc_traffic(dom, B, N, JR, JC, plan.chunks[0]),
                "launch_ms": per[dom],
                "algorithmic_flops_per_launch": flops,
                "hbm_view": {"bound": "hbm", "achieved": bytes_ / dom_s / 1e9, "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": bytes_ / dom_s / 1e9 / PEAK_HBM_GBS,
                             "algorithmic_bytes_per_launch": bytes_},
                "whole_step": {"achieved_tflops": flops / step_s / 1e12,
                               "frac_fp64_valu": flops / step_s / 1e12 / PEAK_FP64_VALU_TFLOPS},
            },
        }
        factor_bytes = B * 8.0 * N * (3 * W + 1)           # phi, u, W, D written
        mat_replay_s = mat_kernel_ms["replay"] / mat_steps * 1e-3
        out["materialize"] = {
            "what": "same step, additionally writing the factor (phi, u, W, D) of all problems to HBM",
            "ms_per_step": mat_total_ms / mat_steps,
            "value": B / (mat_total_ms / mat_steps * 1e-3) * dist.world,
            "kernels_ms": {k: v / mat_steps for k, v in mat_kernel_ms.items()},
            "roofline": {"kernel": "replay (materialising)", "bound": "hbm",
                         "achieved": (factor_bytes + bytes_) / mat_replay_s / 1e9, "peak": PEAK_HBM_GBS,
                         "unit": "GB/s", "frac": (factor_bytes + bytes_) / mat_replay_s / 1e9 / PEAK_HBM_GBS,
                         "bytes_per_launch": factor_bytes + bytes_,
                         "note": "factor written (8 N (3W+1) B per problem) + t, diag, y read, over the "
                                 "replay kernel's HIP-event time"},
        }
        if dist.world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baseline_and_parity(coeffs, t, diag, y, ld, q, B, N))
    dist.barrier()
    plan.close()
    dist.close()
    if out is not None:
        print(json.dumps(out))
    return out


def cpu_baseline_and_parity(coeffs, t, diag, y, ld_gpu, q_gpu, B, N):
    """Times the CPU oracle (a like-for-like port of the reference's
    cholesky.h loops; the reference itself needs Eigen and cannot be built
    here) on this box's host cores and checks the GPU results against it."""
    from oracle import ref

    # single thread, as the reference runs (no threads, GIL held): first S problems
    S = min(B, 1024)
    sub = [c[:S] for c in coeffs]
    t0 = time.perf_counter()
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *sub, t[:S], diag[:S], y[:S], nthreads=1)
    t1 = time.perf_counter() - t0
    cores = os.cpu_count() or 1
    S2 = min(B, max(cores * 8, 64))
    sub2 = [c[:S2] for c in coeffs]
    t0 = time.perf_counter()
    ref.batch_log_likelihood(0.0, *sub2, t[:S2], diag[:S2], y[:S2], nthreads=cores)
    t2 = time.perf_counter() - t0
    ok = s0 == 0
    return {
        "cpu_baseline": {
            "value": S / t1, "unit": "log-likelihoods/s", "cores": 1, "kind": "port",
            "sample": "the first %d of the %d problems of the GPU batch (N=%d, width 8), oracle/"
                      "celerite_ref.c, gcc -O3, 1 thread, %.1f s" % (S, B, N, t1),
            "all_cores": {"value": S2 / t2, "cores": cores,
                          "sample": "%d problems, one per thread over %d threads, %.1f s" % (S2, cores, t2)},
        },
        "parity": {
            "logdet_rel_max": float(np.max(np.abs(ld_gpu[:S][ok] - d0[ok]) / np.abs(d0[ok]))),
            "quad_rel_max": float(np.max(np.abs(q_gpu[:S][ok] - q0[ok]) / np.abs(q0[ok]))),
            "problems_checked": int(S), "tolerance": 1e-10,
        },
    }


if __name__ == "__main__":
    main()
