# -*- coding: utf-8 -*-
"""The N > 1 path on CPU: two `gloo` processes exercise exactly what bench.py
does across GPUs -- contiguous sharding of the batch axis, the timing barrier
and the max-over-ranks reduction -- with the oracle standing in for the device
kernels (no GPU here).  Because problems are independent, the sharded results
must be bit-identical to the unsharded ones."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    from oracle import ref
    from _cases import synthetic, coeffs_of

    d = bench.Dist()
    assert (d.rank, d.world) == (rank, world)
    lo, hi = bench.shard_bounds(total, d.rank, d.world)
    case = synthetic(total, 300, 2, 1, "bench", seed=77)  # same global batch on every rank
    mine = {k: v[lo:hi] for k, v in case.items()}
    d.barrier()
    ll, ld, q, st = ref.batch_log_likelihood(0.0, *coeffs_of(mine), mine["t"], mine["diag"], mine["y"])
    d.barrier()
    slowest = d.max(1.0 + rank)  # max over ranks, as for the step time
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lo=lo, hi=hi, ll=ll, ld=ld, q=q, st=st,
             slowest=slowest)
    d.close()


def test_shard_bounds_cover_the_batch():
    import bench

    for total in (1, 7, 8, 1024, 8191):
        for world in (1, 2, 3, 8):
            edges = [bench.shard_bounds(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharding(tmp_path):
    import torch.multiprocessing as mp

    from oracle import ref
    from _cases import synthetic, coeffs_of

    total, world = 11, 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    case = synthetic(total, 300, 2, 1, "bench", seed=77)
    l0, d0, q0, s0 = ref.batch_log_likelihood(0.0, *coeffs_of(case), case["t"], case["diag"], case["y"])
    seen = np.zeros(total, dtype=bool)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        lo, hi = int(z["lo"]), int(z["hi"])
        seen[lo:hi] = True
        assert np.array_equal(z["ll"], l0[lo:hi]) and np.array_equal(z["ld"], d0[lo:hi])
        assert np.array_equal(z["q"], q0[lo:hi]) and np.array_equal(z["st"], s0[lo:hi])
        assert float(z["slowest"]) == 2.0  # every rank sees the max over ranks
    assert seen.all()
