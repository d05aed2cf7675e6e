"""Experiment: the marker-sharded partitioned sampler on CENTRED columns.  bWGR never centres X, so every pair of
columns is collinear through the mean direction and G > 1 shards overshoot.  Here the same in-process shards run
on x_j - mean(x_j) (a float panel) and are compared with the exact chain on the same centred panel, beside the uncentred numbers.
Usage: centred_shard_probe.py [G:markers_per_round ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bwgr_amd
from bwgr_amd.dist import HipShardEngine, shard_bounds
from conftest import synth_small
from oracle import oracle as O
n, p, it, bi, pi = 800, 16384, int(os.environ.get("IT", "160")), int(os.environ.get("BI", "40")), 0.95
X, y = synth_small(n, p, seed=23, causal=0.01)
y = y.astype(np.float32)
msx = float(O.stats(X)[2])
Xc = np.asfortranarray((X.astype(np.float64) - X.astype(np.float64).mean(0)).astype(np.float32))

def sharded(XX, G, markers_per_round, seed, align):
    spans = [shard_bounds(p, G, r, align) for r in range(G)]
    panels = [bwgr_amd.Panel(np.asfortranarray(XX[:, lo:hi])) for lo, hi in spans]
    engs = [HipShardEngine(panels[r], "BayesB", y, it, bi, pi, 5.0, 0.5, seed, spans[r][0], p, msx) for r in range(G)]
    blk = panels[0].block
    bps = max(1, markers_per_round // blk)
    rounds = max((e.nblocks + bps - 1) // bps for e in engs)
    for _ in range(it):
        for r in range(rounds):
            ds = [e.round_sweep(min(e.nblocks, r * bps), min(e.nblocks, (r + 1) * bps)) for e in engs]
            total = torch.stack(ds).sum(0)
            for e, dlt in zip(engs, ds):
                dlt.copy_(total); e.round_apply(dlt)
        s_ = torch.stack([e.sums() for e in engs]).sum(0)
        for e in engs:
            e.sums().copy_(s_); e.end_iteration(e.sums())
    res = [e.chain.result() for e in engs]
    out = {"ve": res[0]["ve"], "mu": res[0]["mu"], "d": np.concatenate([r_["d"] for r_ in res]),
           "b": np.concatenate([r_["b"] for r_ in res]), "hat": res[0]["mu"] + sum(r_["hat"] - r_["mu"] for r_ in res)}
    for e in engs: e.chain.close()
    for P in panels: P.close()
    return out

cases = [(2, 2048), (4, 1024), (8, 512), (8, 2048)]
if len(sys.argv) > 1: cases = [tuple(int(v) for v in c.split(":")) for c in sys.argv[1:]]
for name, XX, align in (("centred (float panel)", Xc, 128), ("uncentred (int8 panel)", X, 128)):
    a = bwgr_amd.BayesB(y, XX, it=it, bi=bi, pi=pi, seed=31)
    a2 = bwgr_amd.BayesB(y, XX, it=it, bi=bi, pi=pi, seed=77)    # a second exact chain: the Monte-Carlo distance between two exact chains
    print("%s: exact chain ve %.4f mean d %.4f h2 %.3f | second exact chain (other seed): ve %.4f mean d %.4f cor(hat) %.5f cor(b) %.4f" % (
        name, a["ve"], a["d"].mean(), a["h2"], a2["ve"], a2["d"].mean(), np.corrcoef(a["hat"], a2["hat"])[0, 1], np.corrcoef(a["b"], a2["b"])[0, 1]), flush=True)
    for G, mpr in cases:
        s_ = sharded(XX, G, mpr, 31, align)
        print("   %d shards x %5d markers per round: ve %.4f  mean d %.4f  cor(hat) %.5f  cor(b) %.4f  rmse(hat)/sd %.3f" % (
            G, mpr, s_["ve"], s_["d"].mean(), np.corrcoef(s_["hat"], a["hat"])[0, 1], np.corrcoef(s_["b"], a["b"])[0, 1],
            float(np.sqrt(np.mean((s_["hat"] - a["hat"]) ** 2)) / np.std(a["hat"]))), flush=True)
