"""BayesCpi / BayesDpi on the C4-size panel: iterations/s per window while pi adapts (dense start, sparse steady state)."""
import sys, time, json
import torch
import bwgr_amd
from bwgr_amd import synth
n, p = (int(v) for v in (sys.argv[1:3] + ["10000", "1000000"][len(sys.argv) - 1:]))
X = synth.genotypes(n, p, device=0); y = synth.scale_phenotype(synth.phenotype(X, n))
P = bwgr_amd.Panel(X, n=n, device=0); del X
out = {}
for model in ("BayesCpi", "BayesDpi", "BayesC"):
    ch = bwgr_amd.Chain(P, model, y, it=120, bi=0, pi=0.99, df=5, R2=0.5, seed=7)
    rates = []
    for w in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ch.run(20); ch.sync(); t1 = time.perf_counter()
        st = ch.state(); rates.append((round(20 / (t1 - t0), 1), round(float(st["d"].mean()), 4)))
    out[model] = rates; ch.close()
print(json.dumps(out))
