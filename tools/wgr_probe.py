"""wgr() at C4 size (n = 10 000 x p = 1 000 000): seconds per iteration of the device-resident loop in the reference's
settings (BRR, BayesA-like iv, BayesB-like iv + pi, BayesC-like pi), from two run lengths."""
import sys, time, json
import torch
import bwgr_amd
from bwgr_amd import synth
n, p = (int(v) for v in (sys.argv[1:3] + ["10000", "1000000"][len(sys.argv) - 1:]))
X = synth.genotypes(n, p, device=0); y = synth.scale_phenotype(synth.phenotype(X, n)).cpu().numpy().astype("float64")
P = bwgr_amd.Panel(X, n=n, device=0); del X
out = {}
for name, kw in (("BRR", {}), ("BayesA (iv)", {"iv": True}), ("BayesB (iv, pi=0.99)", {"iv": True, "pi": 0.99}), ("BayesC (pi=0.99)", {"pi": 0.99})):
    ts = []
    for it in (6, 26):
        torch.cuda.synchronize(); t0 = time.perf_counter(); bwgr_amd.wgr(y, P, it=it, bi=2, seed=3, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    out[name] = {"ms_per_iteration": round(1e3 * (ts[1] - ts[0]) / 20, 2), "iter_per_s": round(20 / (ts[1] - ts[0]), 1)}
print(json.dumps(out))
