#!/usr/bin/env python3
"""bench.py -- MCMC iterations/second of the bWGR Gibbs hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c4|c3|c2]

One "step" = one MCMC iteration of the fused sampler: the full marker sweep plus that model's intercept / variance
draws and posterior sums (everything inside for(i...) of src/Rcpp20260726ai.cpp:666-688), with X, y and all chain
state resident in HBM before the timed region.  The default workload is the configuration BASELINE.json's target is
quoted on: synthetic n=10,000 x p=1,000,000 int8 genotypes, BayesB with 1 % of markers in the model (bWGR pi=0.99).
It fits one GPU (10.2 GB of X).  For N > 1 every rank runs an exact replica chain on its own copy of the panel, no data-path
collective, `"scaling": "weak"` (the exact sweep is a recurrence and does not shard); `--sharded` selects
the marker-sharded partitioned sampler instead (one chain over N GPUs, RCCL residual all-reduce; strong scaling), which runs on
CENTRED columns -- where it is statistically sound -- unless `--uncentred` is given.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and, at N=1, `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n, p, model, bWGR pi)
    "c2": (5000, 50000, "BayesA", 0.0),
    "c3": (10000, 500000, "BayesB", 0.99),
    "c4": (10000, 1000000, "BayesB", 0.99),
    "c4a": (10000, 1000000, "BayesA", 0.0),    # (not a BASELINE config: C4's shape with an affine model, as wgr() runs by default)
    "c5": (50000, 1000000, "BayesCpi", 0.5),   # BASELINE config 5's panel and model on ONE GPU (50 GB of int8 genotypes)
    "c5b": (50000, 1000000, "BayesB", 0.99),   # (not a BASELINE config: config 5's panel under the headline model)
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(Xs_host, y_host, model, pi, p_total, budget_s=14.0):
    """Times the oracle's float-faithful restatement (1 thread) on a column slice of the same panel and scales the
    sweep time linearly in p (the sweep is O(n*p))."""
    import numpy as np
    from oracle import oracle as O
    n, ps = Xs_host.shape
    Xf = np.asfortranarray(Xs_host, dtype=np.float32)

    def run(it, fast=True):
        t = time.perf_counter()
        O.bayes(model, y_host, Xf, it=it, bi=0, pi=pi, seed=1, flavour="f", fast=fast)
        return time.perf_counter() - t

    t1 = run(1)
    t3 = run(3)
    per_sweep = max((t3 - t1) / 2.0, 1e-9)
    extra = int(max(0, min(40, (budget_s - t1 - t3) / per_sweep - 1)))
    if extra >= 2:
        tk = run(1 + extra)
        per_sweep = (tk - t1) / extra
    full_sweep = per_sweep * (p_total / ps)
    # the same restatement as CRAN would build it (-O2, no -march): two short runs
    o1 = run(1, fast=False); o3 = run(3, fast=False)
    per_sweep_o2 = max((o3 - o1) / 2.0, 1e-9)
    return {"value": 1.0 / full_sweep, "unit": "iter/s", "cores": 1, "kind": "port",
            "value_O2": 1.0 / (per_sweep_o2 * (p_total / ps)),
            "sample": "oracle float-faithful BayesX restatement (1 thread, fp32 X) on the first %d of %d markers at n=%d, scaled "
                      "linearly in p: value = gcc -O3 -march=native, %.3f s per slice sweep; value_O2 = gcc -O2 (CRAN's default "
                      "flags), %.3f s per slice sweep; host has %d cores" % (ps, p_total, n, per_sweep, per_sweep_o2, os.cpu_count())}


def paired_leg(P, model, y, pi, npairs, K, W, n, p):
    """2 x npairs independent chains, two to a set of streamer workgroups (bwgr_chain_run_pair, k_sweep3p: one pass over the
    genotypes serves a pair; every chain is bit-identical to the one it would be alone, tests/test_gpu_parity2.py)."""
    import torch
    import bwgr_amd
    from bwgr_amd import synth
    handles = [P] + [P.clone() for _ in range(2 * npairs - 1)]
    chains = [bwgr_amd.Chain(h, model, y, it=W + K, bi=W, pi=pi, df=5, R2=0.5, seed=synth.SEED + i) for i, h in enumerate(handles)]
    try:
        for _ in range(W):
            for i in range(npairs):
                chains[2 * i].run_pair(chains[2 * i + 1], 1)
        for c in chains:
            c.sync(); c.sweep_ms()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            for i in range(npairs):
                chains[2 * i].run_pair(chains[2 * i + 1], 1)
        for c in chains:
            c.sync()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        kms = [chains[2 * i].sweep_ms()[0] for i in range(npairs)]
        rate = 2 * npairs * K / el
        return {"chains": 2 * npairs, "pairs": npairs, "value": rate, "unit": "chain-iter/s", "ms_per_step_all_chains": 1e3 * el / K,
                "sweep_kernel_ms_per_pair": kms, "genotype_GBps": npairs * (K / el) * float(n) * float(p) / 1e9,
                "frac_of_hbm_peak": npairs * (K / el) * float(n) * float(p) / 1e9 / HBM_PEAK_GBS,
                "note": "two chains share one read of X: bytes moved = pairs x n x p per step"}
    finally:
        for c in chains:
            c.close()
        for h in handles[1:]:
            h.close()


def concurrent_leg(P, model, y, pi, nch, K, W, n, p):
    """nch independent chains (seeds SEED, SEED+1, ...) of the same model on the one resident panel, each on its own
    clone (private sweep scratch + stream, shared genotypes): K iterations of every chain, timed like the main leg.
    This is the throughput shape of the reference's multi-fit callers (mcmcCV, replicate chains); every chain is
    bit-identical to the one it would be alone (tests/test_gpu_parity.py)."""
    import torch
    import bwgr_amd
    from bwgr_amd import synth
    handles = [P] + [P.clone() for _ in range(nch - 1)]
    chains = [bwgr_amd.Chain(h, model, y, it=W + K, bi=W, pi=pi, df=5, R2=0.5, seed=synth.SEED + i) for i, h in enumerate(handles)]
    try:
        for _ in range(W):
            for c in chains:
                c.run(1)
        for c in chains:
            c.sync(); c.sweep_ms()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            for c in chains:
                c.run(1)
        for c in chains:
            c.sync()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        kms = [c.sweep_ms()[0] for c in chains]
        rate = nch * K / el
        return {"chains": nch, "value": rate, "unit": "chain-iter/s", "ms_per_step_all_chains": 1e3 * el / K,
                "sweep_kernel_ms_per_chain": kms, "genotype_GBps": rate * float(n) * float(p) / 1e9,
                "frac_of_hbm_peak": rate * float(n) * float(p) / 1e9 / HBM_PEAK_GBS,
                "note": "each chain reads X itself: bytes moved = chains x n x p per step"}
    finally:
        for c in chains:
            c.close()
        for h in handles[1:]:
            h.close()


def pmc_traffic(workload, n, p, kernel):
    """HBM bytes per launch from the newest committed rocprofv3 --pmc summary of this workload and kernel (profiles/rNN*_pmc_<workload>.json;
    it cannot be read inside this process).  Returns (bytes, source) or (None, None)."""
    import glob
    names = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % workload)), reverse=True)
    for path in names:
        try:
            pm = json.load(open(path))
            if pm["workload"] == workload and pm["n"] == n and pm["p"] == p and kernel.split("<")[0] in pm.get("kernel", "k_sweep2<int8>").split("<")[0].split("::")[-1].split():
                return pm["traffic_bytes_per_launch"], "profiles/%s (an earlier rocprofv3 --pmc pass of this kernel%s, not measured in this run)" % (
                    os.path.basename(path), ", commit %s" % pm["commit"] if pm.get("commit") else "")
        except Exception:
            pass
    return None, None


def single_leg(workload, K, W, dev, args, keep):
    """One exact chain of `workload` alone on the GPU: W untimed iterations, K timed between synchronizes, the sweep kernel's mean launch
    time from hipEvents on its stream.  keep: return the panel and inputs for the legs that follow (the caller closes them)."""
    import torch
    import bwgr_amd
    from bwgr_amd import synth
    n, p, model, pi = WORKLOADS[workload]
    t_setup = time.perf_counter()
    X = synth.genotypes(n, p, device=dev)
    g = synth.phenotype(X, n)
    y = synth.scale_phenotype(g)
    Xs_host = None
    if keep and not args.no_cpu:
        ps = min(args.cpu_slice, p)
        Xs_host = X[:ps, :n].cpu().numpy().T   # (n, ps) view, column-major
    P = bwgr_amd.Panel(X, n=n, device=dev, block=args.block, nwg=args.nwg)
    del X
    torch.cuda.empty_cache()
    ch = bwgr_amd.Chain(P, model, y, it=W + K, bi=W, pi=pi, df=5, R2=0.5, seed=synth.SEED)
    setup_s = time.perf_counter() - t_setup
    ch.run(W)
    ch.sync()
    ch.sweep_ms()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ch.run(K)
    ch.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    sweep_ms, launches = ch.sweep_ms()
    st = ch.state()
    elapsed = t1 - t0
    alg_bytes = float(n) * float(p) * 1.0      # every genotype byte read once per sweep
    pl = P.pipeline(bool(pi) or model in ("BayesCpi", "BayesDpi"))
    mean_d = float(st["d"].mean())
    if pl["generation"] == 3 and mean_d >= float(os.environ.get("BWGR_ENG3_THR", "0.03")):
        # the device picks the selection sweeps' engine from the chain's inclusion rate: above the threshold k_sweep2 ran them
        pl = {"generation": 2, "lag": int(os.environ.get("BWGR_LAG", "3")), "feeders": 0}
    kernel = {4: "k_sweep2w", 3: "k_sweep3<uint%d>" % pl.get("gram_bits", 16), 2: "k_sweep2<int8>", 1: "k_sweep<int8>"}[pl["generation"]]
    # (a chain that has the GPU to itself runs k_sweep3 with 128-row streamers, two to a slab: bwgr_hip.hip, launch_sweep3)
    solo3 = pl["generation"] == 3 and os.environ.get("BWGR_SOLO3", "1") != "0" and P.slab_rows == 256 and 2 * P.nwg + 1 <= 256 and not os.environ.get("BWGR_R3")
    traffic, traffic_source = pmc_traffic(workload, n, p, kernel)
    achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9
    out = {
        "metric": "MCMC iter/sec (full marker sweep)", "value": K / elapsed, "unit": "iter/s", "n_gpus": 1,
        "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 scalars, fixed-point / f64 residual, int8 genotypes", "data": "synthetic",
        "config": {"workload": "%s: synthetic n=%d x p=%d int8, %s%s, exact chain (blocked sweep, block=%d, %d slabs "
                               "x %d rows + 1 sequencer%s)" % (workload, n, p, model,
                                                          " pi=%.2f (%.1f%% of the markers in the model at the last sweep)" % (pi, 100.0 * mean_d) if pi else "", P.block, P.nwg, P.slab_rows,
                                                          (" + %d q feeders" % pl["feeders"] if pl["feeders"] else "")
                                                          + (", the chain alone on the GPU: %d streamers x 128 rows" % (2 * P.nwg) if solo3 else "")
                                                          + ", engine generation %d, lag %d blocks" % (pl["generation"], pl["lag"])),
                   "n": n, "p": p, "model": model, "pi": pi, "df": 5, "R2": 0.5, "chains": 1},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "kernel": kernel,
                     "kernel_ms": sweep_ms, "launches": launches, "algorithmic_bytes_per_launch": alg_bytes},
        "setup_s": setup_s,
        "chain_check": {"ve": st["ve"], "mu": st["mu"], "mean_d": mean_d},
    }
    ch.close()
    if keep:
        return out, P, y, Xs_host, pl
    P.close()
    del P, y
    torch.cuda.empty_cache()
    return {k: out[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline", "setup_s", "chain_check")}


def intra_gpu_shards_leg(model, pi, S, K, W, n, p, dev):
    """The partitioned sampler's shards SIDE BY SIDE ON THIS ONE GPU (Group(devices=[dev] * S), include/bwgr.h "shards side by side on one GPU"):
    S marker shards of the same panel, each an int8 panel swept as implicitly centred columns (bwgr_panel_set_centred) by its own k_sweep3 launch
    on its own stream and compute units, residual deltas summed at the exchange rounds by a kernel.  One exact chain is a latency-bound pipeline
    on a third of the chip; this is the north-star's marker-sharded independent-block sampler filling the rest.  NOT the reference's chain
    (statistical parity: tests/test_gpu_parity3.py::test_shards_side_by_side_on_one_gpu, ::test_partitioned_sampler_on_centred_columns), so it
    is reported beside the headline, never as it."""
    import torch
    import bwgr_amd
    from bwgr_amd import synth
    X = synth.genotypes(n, p, device=dev)
    y = synth.scale_phenotype(synth.phenotype(X, n))
    g = bwgr_amd.Group(model, y, X, devices=[dev] * S, it=W + K, bi=W, pi=pi, df=5, R2=0.5, seed=synth.SEED, centre=True, n=n)
    try:
        g.run(W); g.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.run(K); g.sync(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        info = g.info()
        r = g.result()
        rate = K / el
        return {"shards": S, "value": rate, "unit": "iter/s", "ms_per_step": 1e3 * el / K, "genotype_GBps": rate * float(n) * float(p) / 1e9,
                "frac_of_hbm_peak": rate * float(n) * float(p) / 1e9 / HBM_PEAK_GBS, "rounds_per_sweep": info["rounds_per_sweep"],
                "markers_per_round": info["markers_per_round"], "centring": "implicit-int8", "statistically_sound": bool(r["statistically_sound"]),
                "chain_check": {"ve": float(r["ve"]), "mean_d": float(r["d"].mean())},
                "note": "partitioned (independent-block) Gibbs sampler over %d marker shards of one panel on ONE GPU: a different chain from the reference's, "
                        "statistically sound on centred columns; every genotype byte is read once per iteration" % S}
    finally:
        g.close()
        del X
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--nwg", type=int, default=0)
    ap.add_argument("--sync-every", type=int, default=0, help="markers per rank between residual all-reduces (N>1)")
    ap.add_argument("--chains", type=int, default=0, help="extra leg at N=1: this many chains side by side on the one "
                    "resident panel (0 = as many as fit the chip, 1 = skip the leg); reported as concurrent_chains")
    ap.add_argument("--sharded", action="store_true", help="N > 1: the marker-sharded partitioned sampler (one chain over N GPUs, RCCL "
                    "residual all-reduce) instead of N replica chains; statistically unsound on uncentred genotypes, see DESIGN.md section 8")
    ap.add_argument("--uncentred", action="store_true", help="--sharded on the raw int8 genotypes (statistically UNSOUND for N > 1: DESIGN.md section 8; "
                    "kept for measurements)")
    ap.add_argument("--centre-explicit", action="store_true", help="--sharded: centre into a float32 copy of the shard (round 3's path, 4 bytes per genotype) instead of "
                    "sweeping the int8 shard as implicitly centred columns")
    ap.add_argument("--shards", type=int, default=3, help="extra leg at N=1: the partitioned sampler's marker shards side by side on the one GPU, implicitly centred "
                    "int8 (reported as intra_gpu_shards; 0 or 1 = skip)")
    ap.add_argument("--pairs", type=int, default=0, help="pairs of chains in the paired-chains leg (0: as many as fit)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the c2 / c3 legs that follow the headline workload (`workloads` in the JSON line)")
    ap.add_argument("--cpu-slice", type=int, default=20000)
    args = ap.parse_args()

    import numpy as np
    import torch
    import bwgr_amd
    from bwgr_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    n, p, model, pi = WORKLOADS[args.workload]
    K, W = args.steps, args.warmup
    if bwgr_amd.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: bwgr_amd has no CPU fallback")
    ndev = torch.cuda.device_count()
    shared = local_rank >= ndev     # rehearsal on a box with fewer GPUs than ranks: ranks share devices, gloo instead of RCCL
    dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev)

    if (world > 1 and args.sharded) or os.environ.get("BWGR_FORCE_DIST"):   # BWGR_FORCE_DIST=1: rehearse the sharded leg with one rank
        from bwgr_amd import dist as bdist
        out = bdist.bench_sharded(args, n, p, model, pi, K, W, rank, world, dev)
        if rank == 0:
            print(json.dumps(out))
        return
    if world > 1:
        # N replica chains: every rank stages the full panel (10 GB at C4) and runs the EXACT chain with its own seed; no data-path
        # collective.  The sweep is a recurrence over the markers, so one exact chain does not shard (the
        # partitioned sampler the north-star sketches overshoots on uncentred genotypes at every exchange window that would scale).
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        shared_any = world > ndev
        if not dist.is_initialized():
            if shared_any: dist.init_process_group("gloo")
            else: dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        X = synth.genotypes(n, p, device=dev)
        y = synth.scale_phenotype(synth.phenotype(X, n))
        P = bwgr_amd.Panel(X, n=n, device=dev, block=args.block, nwg=args.nwg)
        del X
        torch.cuda.empty_cache()
        ch = bwgr_amd.Chain(P, model, y, it=W + K, bi=W, pi=pi, df=5, R2=0.5, seed=synth.SEED + rank)
        ch.run(W); ch.sync(); ch.sweep_ms()
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ch.run(K); ch.sync()
        dist.barrier(); torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if shared_any else "cuda:%d" % dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
        sweep_ms, launches = ch.sweep_ms()
        st = ch.state()
        mean_d = float(st["d"].mean())
        pl = P.pipeline(bool(pi))
        kernel = {4: "k_sweep2w", 3: "k_sweep3<uint16>", 2: "k_sweep2<int8>", 1: "k_sweep<int8>"}[pl["generation"]]
        alg_bytes = float(n) * float(p)
        achieved = alg_bytes / (sweep_ms * 1e-3) / 1e9
        out = {
            "metric": "MCMC iter/sec (full marker sweep)", "value": world * K / elapsed, "unit": "iter/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 scalars, fixed-point / f64 residual, int8 genotypes", "data": "synthetic",
            "config": {"workload": "%s: synthetic n=%d x p=%d int8, %s%s: %d replica chains, one exact chain per GPU on its own copy "
                                   "of the panel (replicas only: no data-path collective; value = chain-iterations/s over all GPUs)"
                                   % (args.workload, n, p, model, " pi=%.2f" % pi if pi else "", world),
                       "n": n, "p": p, "model": model, "pi": pi, "df": 5, "R2": 0.5, "chains": world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "traffic_source": None, "kernel": kernel + " (rank 0)", "kernel_ms": sweep_ms,
                         "launches": launches, "algorithmic_bytes_per_launch": alg_bytes},
            "chain_check": {"ve": st["ve"], "mu": st["mu"], "mean_d": mean_d},
        }
        ch.close(); P.close()
        dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
        return

    # ---- single GPU ----
    out, P, y, Xs_host, pl = single_leg(args.workload, K, W, dev, args, keep=True)
    # BASELINE configs 2 and 3 on the same box, same timing discipline (a barrier-free single GPU: synchronize on both sides, K steps
    # timed exactly, kernel time by hipEvents on the kernel's stream): driver-visible beside the headline, each with its own roofline.
    # (Before the legs that open further streams: c2 is about 1 ms and fifteen launches a step, and once the process holds several hardware
    # queues the gaps between its launches double -- 1.84 against 1.07 ms a step at the same kernel time when this ran last.)
    if not args.no_extra and args.workload == "c4":
        out["workloads"] = {}
        for wl in ("c2", "c3"):
            try:
                out["workloads"][wl] = single_leg(wl, max(K, 10) * (5 if wl == "c2" else 1), max(W, 2), dev, args, keep=False)   # (c2 is 1 ms per step: more of them)
            except Exception as ex:
                out["workloads"][wl] = {"error": str(ex)}
    nch = args.chains if args.chains > 0 else P.max_concurrent(bool(pi))
    nch = min(nch, P.max_concurrent(bool(pi)))
    if nch > 1:
        try:
            out["concurrent_chains"] = concurrent_leg(P, model, y, pi, nch, K, W, n, p)
        except Exception as ex:   # the headline leg above stands on its own
            out["concurrent_chains"] = {"chains": nch, "error": str(ex)}
    if model in ("BayesB", "BayesC") and pi >= 0.95 and pl["generation"] == 3 and args.chains != 1:   # (fit_many's rule for pairing)
        npairs = args.pairs if args.pairs > 0 else max(1, P.max_pairs())   # (bwgr_panel_max_pairs: a pair holds K3 + 2 CUs for the sweep; ~40 CUs stay free for the iterations' small kernels: six pairs at C4 measured slower than five)
        try:
            os.environ["BWGR_ENG3_THR"] = os.environ.get("BWGR_ENG3_THR", "0.03")
            out["paired_chains"] = paired_leg(P, model, y, pi, npairs, K, W, n, p)
        except Exception as ex:
            out["paired_chains"] = {"pairs": npairs, "error": str(ex)}
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(Xs_host, y.cpu().numpy(), model, pi, p)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    P.close()
    del P, y
    torch.cuda.empty_cache()
    if args.shards > 1 and args.workload in ("c3", "c4") and model in ("BayesB", "BayesC"):
        try:
            out["intra_gpu_shards"] = intra_gpu_shards_leg(model, pi, args.shards, K, W, n, p, dev)
        except Exception as ex:
            out["intra_gpu_shards"] = {"shards": args.shards, "error": str(ex)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
