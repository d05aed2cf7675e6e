"""CPU, gloo, world_size 2: the marker-sharded driver (bwgr_amd/dist.py run_iterations) with a checker engine.

The product engine is the HIP one; here the per-rank sweep is the oracle's KMUP with pi = 0 and L = lambda, which is
BayesRR's sweep (src/Rcpp20260726ai.cpp:834-837), and the tail restates :839-843 with the oracle's variates.  What is
under test is the exchange logic: residual-delta all-reduce at block-range boundaries, global sums, identical tails.
"""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import synth_small
from oracle import oracle as O
from bwgr_amd import dist as bdist

from shard_checker import OracleRREngine, f32


def _problem():
    X, y = synth_small(120, 96, seed=12, causal=0.2)
    return X, y, float(O.stats(X)[2])


def test_world1_driver_is_the_exact_chain():
    X, y, msx = _problem()
    eng = OracleRREngine(X, y, 0, X.shape[1], msx, 5.0, 0.5, 31, block=16)
    bdist.run_iterations(eng, 8, blocks_per_sync=2, world=1)
    ref = O.bayes("BayesRR", y, X, it=8, bi=0, seed=31)["last"]
    assert np.max(np.abs(eng.b - ref["b"])) / np.max(np.abs(ref["b"])) < 2e-5
    assert np.max(np.abs(eng.e.numpy() - ref["e"])) / np.max(np.abs(ref["e"])) < 2e-5
    assert abs(float(eng.ve) - ref["ve"]) / ref["ve"] < 2e-5 and abs(float(eng.mu) - ref["mu"]) < 1e-6


def _worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, y, msx = _problem()
        n, p = X.shape
        lo, hi = bdist.shard_bounds(p, world, rank, 16)
        eng = OracleRREngine(X[:, lo:hi], y, lo, p, msx, 5.0, 0.5, 31, block=16)
        iters = 40
        bdist.run_iterations(eng, iters, blocks_per_sync=1, world=world)
        # (a) replicas of the residual and of the scalars agree bit for bit on every rank
        es = [torch.zeros_like(eng.e) for _ in range(world)]
        dist.all_gather(es, eng.e)
        assert all(torch.equal(es[0], t) for t in es)
        sc = torch.tensor([float(eng.ve), float(eng.mu), float(eng.vb)], dtype=torch.float64)
        scs = [torch.zeros_like(sc) for _ in range(world)]
        dist.all_gather(scs, sc)
        assert all(torch.equal(scs[0], t) for t in scs)
        # (b) the replicated residual is y - mu - X b over ALL shards
        xb = torch.from_numpy(X[:, lo:hi].astype(np.float64) @ eng.b.astype(np.float64))
        dist.all_reduce(xb)
        e_expect = y.astype(f32).astype(np.float64) - float(eng.mu) - xb.numpy()
        assert np.max(np.abs(eng.e.numpy() - e_expect)) / np.max(np.abs(e_expect)) < 1e-4
        # (c) statistical sanity against the exact (unsharded) chain: residual variance posterior mean
        ref = O.bayes("BayesRR", y, X, it=iters, bi=0, seed=31)
        assert abs(eng.VE / eng.nacc - ref["ve"] * (iters / (iters - 1))) / ref["ve"] < 0.25
    finally:
        dist.destroy_process_group()


def test_world2_gloo_sharded_exchange():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port), nprocs=2, join=True)


def test_shard_bounds_cover_all_markers():
    with pytest.raises(ValueError):
        bdist.shard_bounds(130, 4, 2, 128)          # more ranks than marker blocks
    for p, world, block in [(1000, 8, 128), (96, 2, 16), (1_000_000, 8, 128), (1_000_000, 4, 128)]:
        spans = [bdist.shard_bounds(p, world, r, block) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == p
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert all(lo % block == 0 for lo, _ in spans)
