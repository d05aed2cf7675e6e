"""wgr(bag = 0.5) at C4 size: ms per iteration (row subsample + Gram rebuild + KMUP2 sweep each iteration)."""
import sys, time, json
import torch
import bwgr_amd
from bwgr_amd import synth
n, p = 10000, 1000000
X = synth.genotypes(n, p, device=0); y = synth.scale_phenotype(synth.phenotype(X, n)).cpu().numpy().astype("float64")
P = bwgr_amd.Panel(X, n=n, device=0); del X
ts = []
for it in (4, 14):
    torch.cuda.synchronize(); t0 = time.perf_counter(); bwgr_amd.wgr(y, P, it=it, bi=1, seed=3, iv=True, pi=0.99, bag=0.5); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(json.dumps({"wgr bag=0.5 BayesB setting": {"ms_per_iteration": round(1e3 * (ts[1] - ts[0]) / 10, 1)}}))
