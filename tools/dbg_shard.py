import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bwgr_amd
from bwgr_amd.dist import HipShardEngine, shard_bounds
from oracle import oracle as O
from shard_checker import OracleRREngine
from conftest import synth_small, scaled_err
X, y = synth_small(300, 160, seed=14, causal=0.2)
n, p = X.shape
msx = float(O.stats(X)[2])
lo, hi = 0, 80
for ranges in ([(0, 2), (2, 5)], [(0, 5)], [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)], [(0,3),(3,5)]):
    P = bwgr_amd.Panel(np.asfortranarray(X[:, lo:hi]), block=16)
    g = HipShardEngine(P, "BayesRR", y, 6, 0, 0.0, 5.0, 0.5, 41, lo, p, msx)
    c = OracleRREngine(X[:, lo:hi], y, lo, p, msx, 5.0, 0.5, 41, block=16)
    for (a, z) in ranges:
        g.sweep_blocks(a, z); c.sweep_blocks(a, z)
        st = g.chain.state()
        errs = [scaled_err(st["b"][k*16:(k+1)*16], c.b[k*16:(k+1)*16]) if np.abs(c.b[k*16:(k+1)*16]).max() > 0 else float(np.abs(st["b"][k*16:(k+1)*16]).max()) for k in range(5)]
        print(ranges, (a, z), "per-block b err", ["%.1e" % e for e in errs], "e err %.1e" % scaled_err(g.e[:n].cpu().numpy(), c.e.numpy()))
    g.chain.close(); P.close()
