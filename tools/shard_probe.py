"""One rank's share of the marker-sharded sampler on one GPU: p_local markers (C4 split over G ranks), the exchange loop of
bwgr_amd/dist.py forced on (a one-rank RCCL group), to see what the rounds cost on top of the sweeps.
python tools/shard_probe.py [G] [implicit|none]   (implicit: the shard swept as implicitly centred columns, bwgr_panel_set_centred -- what bench.py --sharded runs)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
import bwgr_amd
from bwgr_amd import synth, dist as bdist

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
MODE = sys.argv[2] if len(sys.argv) > 2 else "implicit"
n, p = int(os.environ.get("SP_N", "10000")), 1000000           # SP_N=50000 SP_MODEL=BayesCpi SP_PI=0.5: config 5's panel and model
MODEL, PI = os.environ.get("SP_MODEL", "BayesB"), float(os.environ.get("SP_PI", "0.99"))
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
lo, hi = bdist.shard_bounds(p, G, 0, 128)
X = synth.genotypes(n, hi - lo, col0=lo, device=0)
y = synth.scale_phenotype(synth.phenotype(X, n, col0=lo, p_total=p))
P = bwgr_amd.Panel(X, n=n, device=0, block=128); del X
if MODE == "implicit":
    P.set_centred(True)
msx = P.stats()[2] * G
K, W = int(os.environ.get("SP_K", "30")), 5
eng = bdist.HipShardEngine(P, MODEL, y, 2 * (W + K), 0, PI, 5.0, 0.5, synth.SEED, lo, p, msx)
bps = max(1, max(P.block, 131072 // G) // P.block)
rounds = (eng.nblocks + bps - 1) // bps
out = {"G": G, "n": n, "model": MODEL, "p_local": hi - lo, "rounds_per_sweep": rounds, "centring": MODE}
def exchange(iters):
    nb = eng.nblocks
    for _ in range(iters):
        for r in range(rounds):
            lo_b, hi_b = r * bps, min(nb, (r + 1) * bps)
            delta = eng.round_sweep(lo_b, hi_b)
            dist.all_reduce(delta)
            eng.round_apply(delta)
        s = eng.sums(); dist.all_reduce(s); eng.end_iteration(s)
exchange(W); eng.chain.sync(); torch.cuda.synchronize(); t0 = time.perf_counter(); exchange(K); eng.chain.sync(); torch.cuda.synchronize()
out["ms_per_iteration_with_rounds"] = round(1e3 * (time.perf_counter() - t0) / K, 3)
ms, launches = eng.chain.sweep_ms(); out["sweep_kernel_ms_per_iteration"] = round(ms * launches / (W + K), 3)
out["chain_check"] = {k: (float(v.mean()) if k == "d" else float(v)) for k, v in eng.chain.state().items() if k in ("ve", "mu", "d")}
eng.chain.close()
ch = bwgr_amd.Chain(P, MODEL, y, it=W + K, bi=0, pi=PI, seed=synth.SEED)
ch.run(W); ch.sync(); torch.cuda.synchronize(); t0 = time.perf_counter(); ch.run(K); ch.sync(); torch.cuda.synchronize()
out["ms_per_iteration_plain_chain"] = round(1e3 * (time.perf_counter() - t0) / K, 3)
print(json.dumps(out)); dist.destroy_process_group()
