"""Marker-sharded multi-GPU sampler: one process per GPU, torch.distributed over RCCL/xGMI.

Rank g owns the columns [g*p/G, (g+1)*p/G) of X and the matching b, d, vb, lambda slices; the residual e (n fp64
values, 80 KB at n=10k) is replicated.  Between synchronisations every rank sweeps `blocks_per_sync` of its own marker
blocks against its local copy of e (exactly, with the blocked sweep kernel); at the boundary ONE all-reduce(sum) of
the residual delta (e - e_at_boundary) makes the copies identical again.  Per iteration one more tiny all-reduce
carries {sum d, sum b^2}; the intercept / variance draws are then computed redundantly and identically on every rank
(same RNG counters).  The message is latency-bound (80 KB), so the number of boundaries per sweep, not the link
bandwidth, is what costs: keep p_local/blocks_per_sync small.

This is a partitioned ("independent-block") Gibbs sampler: for G > 1 markers on different ranks do not see each other's
updates until the next boundary, so it is NOT the reference's chain -- parity is statistical (posterior means within
Monte-Carlo error), and `blocks_per_sync` trades mixing fidelity for speed.  G = 1 takes no exchange path at all and
is the exact chain.

SOUND ONLY ON CENTRED COLUMNS.  bWGR never centres X, so all columns are collinear through the mean direction: every
shard corrects the same stale residual mean and the summed corrections overshoot (4 shards: ve 15 against 1.45; DESIGN.md
section 8).  On x_j - mean(x_j) the same driver follows the exact chain with 2, 4 and 8 shards (ve within 1-3 %, mean(d)
equal, cor(hat) 0.994: tools/centred_shard_probe.py, tests/test_gpu_parity3.py::test_partitioned_sampler_on_centred_columns);
centring is a reparametrisation under the samplers' flat intercept prior (/root/reference/src/Rcpp20260726ai.cpp:683-684) that an exact
Gibbs sampler would not notice; bWGR's own chain does a little (ve 1.47 uncentred against 1.56 centred on the probe panel, DESIGN.md
section 8), so "sound" means: follows the exact chain on the same centred panel.  bench_sharded therefore centres its shard unless told not to --
IMPLICITLY since round 4 (bwgr_panel_set_centred: the genotypes stay int8 in HBM, k_sweep3 sweeps them and its sequencer carries the scalar terms;
`--centre-explicit` keeps round 3's float copy of the centred columns) -- and `statistically_sound` is computed from the panel (bwgr_panel_centred),
not from the output.

The driver is engine-agnostic: an engine exposes sweep_blocks / residual / set_residual / sums / end_iteration over
torch tensors.  The product engine is HipShardEngine (bwgr_amd.Chain on the GPU).  tests/test_dist_gloo.py drives the
same driver on CPU over gloo with a checker engine.
"""
import math
import os
import time

import numpy as np
import torch  # noqa: F401  (before the library is loaded: torch bundles its own HIP runtime, see _lib.lib)


class HipShardEngine:
    def __init__(self, panel, model, y, it, bi, pi, df, R2, seed, marker0, p_total, msx_total):
        import torch
        from .api import Chain
        self.panel = panel
        self.e = torch.zeros(panel.ld, dtype=torch.float64, device="cuda:%d" % panel.device)
        self.chain = Chain(panel, model, y, it=it, bi=bi, pi=pi, df=df, R2=R2, seed=seed,
                           shard=(marker0, p_total, msx_total), e_ext=self.e)
        self.nblocks = self.chain.nblocks

    def sweep_blocks(self, lo, hi):
        self.chain.sweep_blocks(lo, hi)

    def residual(self):
        return self.e

    def set_residual(self, t):
        self.e.copy_(t)

    def round_sweep(self, lo, hi):
        """First half of an exchange round inside the library (two small launches instead of three tensor operations):
        returns the tensor to all-reduce."""
        if not hasattr(self, "_delta"):
            import torch
            self._delta = torch.empty_like(self.e)
        self.chain.round_sweep(lo, hi, self._delta)
        return self._delta

    def round_apply(self, delta):
        self.chain.round_apply(delta)

    def sums(self):
        import torch
        if not hasattr(self, "_sums"):
            self._sums = torch.zeros(2, dtype=torch.float64, device=self.e.device)
        self.chain.get_sums_dev(self._sums)   # stays on the device: the all-reduce and end_iteration take it from there
        return self._sums

    def end_iteration(self, sums_total):
        if sums_total is None:
            self.chain.end_iteration(None)
        elif sums_total.is_cuda:
            self.chain.end_iteration_dev(sums_total)
        else:
            self.chain.end_iteration(sums_total.detach().cpu().numpy())


def sync_rounds(nblocks_local, blocks_per_sync, world):
    """Every rank must enter the same number of all-reduces per sweep even when shards differ by a block."""
    import torch
    import torch.distributed as dist
    r = math.ceil(nblocks_local / blocks_per_sync)
    if world > 1:
        t = torch.tensor([r], dtype=torch.int64, device=_comm_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        r = int(t.item())
    return r


def _all_reduce(t, op=None):
    """dist.all_reduce; a CUDA tensor under gloo (the rehearsal with more ranks than GPUs, and the CPU tests) goes through the host."""
    import torch.distributed as dist
    kw = {} if op is None else {"op": op}
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.cpu()
        dist.all_reduce(h, **kw)
        t.copy_(h)
    else:
        dist.all_reduce(t, **kw)


def _comm_device():
    import torch
    import torch.distributed as dist
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def run_iterations(engine, iters, blocks_per_sync, world, rounds=None):
    """`iters` MCMC iterations of the sharded sampler.  world == 1: no exchange (the exact chain)."""
    import torch.distributed as dist
    nb = engine.nblocks
    if world == 1:
        for _ in range(iters):
            engine.sweep_blocks(0, nb)
            engine.end_iteration(None)
        return
    bps = max(1, int(blocks_per_sync))
    if rounds is None:
        rounds = sync_rounds(nb, bps, world)
    for _ in range(iters):
        for r in range(rounds):
            lo, hi = r * bps, min(nb, (r + 1) * bps)
            if hasattr(engine, "round_sweep"):      # the product engine keeps the round's vector arithmetic in the library
                delta = engine.round_sweep(lo, hi)  # (lo >= hi on a rank that has run out of blocks: nothing is swept)
                _all_reduce(delta)                  # RCCL over xGMI: n fp64 values
                engine.round_apply(delta)
                continue
            e0 = engine.residual().clone()
            if lo < hi:
                engine.sweep_blocks(lo, hi)
            delta = engine.residual() - e0
            _all_reduce(delta)                      # (gloo in the CPU tests)
            engine.set_residual(e0 + delta)
        s = engine.sums()
        _all_reduce(s)
        engine.end_iteration(s)


def shard_bounds(p, world, rank, block):
    """Contiguous column shards aligned to the marker block size."""
    nblk = (p + block - 1) // block
    per = (nblk + world - 1) // world
    lo = min(p, rank * per * block)
    hi = min(p, (rank + 1) * per * block)
    if hi <= lo:
        raise ValueError("rank %d of %d would own no markers: p=%d has only %d blocks of %d" % (rank, world, p, nblk, block))
    return lo, hi


def bench_sharded(args, n, p, model, pi, K, W, rank, world, dev):
    """bench.py's N > 1 leg: strong scaling of one panel over `world` GPUs."""
    import torch
    import torch.distributed as dist
    import bwgr_amd
    from . import synth
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        if world > torch.cuda.device_count(): dist.init_process_group("gloo")   # rehearsal: ranks share devices, collectives through the host
        else: dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    block = args.block if args.block > 0 else 128
    lo, hi = shard_bounds(p, world, rank, block)
    X = synth.genotypes(n, hi - lo, col0=lo, device=dev)
    g = synth.phenotype(X, n, col0=lo, p_total=p)
    _all_reduce(g)
    y = synth.scale_phenotype(g)
    # BWGR_FORCE_CENTRE=1: rehearse the centred panel with one rank (bench.py's BWGR_FORCE_DIST leg)
    centre = (world > 1 or bool(os.environ.get("BWGR_FORCE_CENTRE"))) and not getattr(args, "uncentred", False)
    # (implicit centring covers the selection models on both of their engines, k_sweep3 and k_sweep2; the affine models keep the float copy, which
    # k_sweep2w / k_sweep2 sweep)
    explicit = centre and (getattr(args, "centre_explicit", False) or model not in ("BayesB", "BayesC", "BayesCpi", "BayesDpi"))
    if explicit:   # x_j - mean(x_j) as a float panel (column-major n x p_local on the device, in column chunks to bound the temporaries)
        Xf = torch.empty((hi - lo, n), dtype=torch.float32, device=X.device)     # (p_local, n): row j = column j, no row padding
        for c0 in range(0, hi - lo, 8192):
            blk_ = X[c0:c0 + 8192, :n].to(torch.float32)
            Xf[c0:c0 + 8192] = blk_ - blk_.mean(dim=1, keepdim=True)
        del X
        X = Xf
        block = 64 if args.block <= 0 else min(args.block, 64)      # (float panels: 64-marker blocks)
    P = bwgr_amd.Panel(X, n=n, device=dev, block=block, nwg=args.nwg)
    del X
    torch.cuda.empty_cache()
    if centre and not explicit:
        P.set_centred(True)      # implicit: int8 in HBM, the shard's own column means (rows are not sharded)
    msx = torch.tensor([P.stats()[2]], dtype=torch.float64, device="cuda:%d" % dev)
    _all_reduce(msx)
    eng = HipShardEngine(P, model, y, W + K, W, pi, 5.0, 0.5, synth.SEED, lo, p, float(msx.item()))
    # default: one residual all-reduce per 131 072 markers swept over ALL ranks, so the staleness window of the partitioned
    # sampler -- how many markers are updated against a residual that has not seen the other ranks' updates yet -- does not
    # depend on the number of GPUs (8 GPUs: every 16 384 markers per rank)
    markers_per_sync = args.sync_every if args.sync_every > 0 else max(P.block, 131072 // world)
    bps = max(1, markers_per_sync // P.block)
    rounds = sync_rounds(eng.nblocks, bps, world)
    run_iterations(eng, W, bps, world, rounds)
    eng.chain.sync(); eng.chain.sweep_ms()
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_iterations(eng, K, bps, world, rounds)
    eng.chain.sync()
    dist.barrier(); torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda:%d" % dev)
    _all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    ms, launches = eng.chain.sweep_ms()
    sweep_ms_per_iter = ms * launches / K
    st = eng.chain.state()
    alg = float(n) * float(hi - lo) * (4.0 if explicit else 1.0)
    ach = alg / (sweep_ms_per_iter * 1e-3) / 1e9
    out = {
        "metric": "MCMC iter/sec (full marker sweep)", "value": K / elapsed, "unit": "iter/s", "n_gpus": world, "steps": K,
        "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32 scalars, f64 residual/accumulation, %s genotypes" % ("centred f32" if explicit else ("int8, implicitly centred" if centre else "int8")), "data": "synthetic",
        "config": {"workload": "%s: synthetic n=%d x p=%d %s, %s%s, markers sharded over %d GPUs (%d per rank), residual "
                               "all-reduce (n fp64) every %d markers per rank = %d per sweep; partitioned Gibbs (statistical "
                               "parity for N>1)" % (args.workload, n, p, "int8 centred to f32 columns" if explicit else ("int8 swept as implicitly centred columns" if centre else "int8"), model,
                                                    " pi=%.2f" % pi if pi else "", world, hi - lo, bps * P.block, rounds),
                   "n": n, "p": p, "model": model, "pi": pi, "df": 5, "R2": 0.5, "chains": 1, "block": P.block,
                   "slab_workgroups": P.nwg, "sync_rounds_per_sweep": rounds},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
                     "kernel": "%s (rank 0, all launches of one sweep summed)" % ("k_sweep2<float>" if explicit else ("k_sweep3 / k_sweep2<int8>, implicitly centred" if centre else "k_sweep2<int8> / k_sweep3")), "kernel_ms": sweep_ms_per_iter,
                     "launches": launches, "algorithmic_bytes_per_launch": alg},
        "chain_check": {"ve": st["ve"], "mu": st["mu"], "mean_d_rank0": float(st["d"].mean())},
    }
    # sound = one rank (the exact chain) or centred columns: a criterion on the panel (bwgr_panel_centred: every |mean_j| <= 1e-3 sd_j), not on
    # the output.  Measured: centred, 2 / 4 / 8 shards follow the exact chain; uncentred they overshoot
    cen = torch.tensor([1 if P.centred() else 0], dtype=torch.int64, device="cuda:%d" % dev)
    _all_reduce(cen, op=dist.ReduceOp.MIN)
    out["statistically_sound"] = bool(world == 1 or int(cen.item()) == 1)
    out["centred_columns"] = bool(int(cen.item()) == 1)
    out["note"] = ("marker-sharded partitioned Gibbs sampler: NOT the reference's chain for N > 1; statistically sound on centred columns "
                   "(tests/test_gpu_parity3.py::test_partitioned_sampler_on_centred_columns: 2, 4, 8 shards against the exact chain), unsound on "
                   "uncentred genotypes (tests/test_gpu_parity2.py::test_partitioned_sampler_characterisation, DESIGN.md section 8); "
                   + ("explicitly centred shards are float panels (4 bytes per genotype), so `roofline` here is priced on 4 n p bytes" if explicit else
                      "the shards are int8 panels swept as implicitly centred columns (bwgr_panel_set_centred): `roofline` is priced on n p bytes"))
    out["centring"] = "explicit-f32" if explicit else ("implicit-int8" if centre else "none")
    dist.destroy_process_group()
    return out
