"""Statistical check of the shards-side-by-side sampler at bench scale: the exact implicitly centred chain (one shard) against S shards on one GPU,
same data, same seed, posterior means over IT - BI iterations.  python tools/intra_shard_soak.py [S ...]   (AB_P markers, default 300 000)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bwgr_amd
from bwgr_amd import synth

n, p = 10000, int(os.environ.get("AB_P", "300000"))
IT, BI = int(os.environ.get("AB_IT", "400")), int(os.environ.get("AB_BI", "100"))
X = synth.genotypes(n, p)
y = synth.scale_phenotype(synth.phenotype(X, n))
res = {}
for S in [1] + ([int(a) for a in sys.argv[1:]] or [3, 6]):
    t0 = time.perf_counter()
    g = bwgr_amd.Group("BayesB", y, X, devices=[0] * S, it=IT, bi=BI, pi=0.99, seed=synth.SEED, centre=True, n=n, markers_per_sync=int(os.environ.get("AB_MPS", "0")))
    info = g.info()
    g.run(IT); g.sync()
    r = g.result(); g.close()
    res[S] = r
    line = {"shards": S, "rounds_per_sweep": info["rounds_per_sweep"], "ve": float(r["ve"]), "h2": float(r["h2"]), "mean_d": float(r["d"].mean()), "seconds": round(time.perf_counter() - t0, 1)}
    if S > 1:
        a = res[1]
        line.update({"cor_hat_vs_exact": float(np.corrcoef(r["hat"], a["hat"])[0, 1]), "cor_b_vs_exact": float(np.corrcoef(r["b"], a["b"])[0, 1]),
                     "ve_rel_diff": float(abs(r["ve"] - a["ve"]) / a["ve"]), "cor_d_vs_exact": float(np.corrcoef(r["d"], a["d"])[0, 1])})
    print(json.dumps(line), flush=True)
# two exact chains with different seeds: the Monte-Carlo floor of the comparison
g = bwgr_amd.Group("BayesB", y, X, devices=[0], it=IT, bi=BI, pi=0.99, seed=synth.SEED + 1, centre=True, n=n)
g.run(IT); g.sync(); r2 = g.result(); g.close()
a = res[1]
print(json.dumps({"exact_other_seed": True, "ve": float(r2["ve"]), "mean_d": float(r2["d"].mean()), "cor_hat_vs_exact": float(np.corrcoef(r2["hat"], a["hat"])[0, 1]),
                  "cor_b_vs_exact": float(np.corrcoef(r2["b"], a["b"])[0, 1]), "ve_rel_diff": float(abs(r2["ve"] - a["ve"]) / a["ve"])}), flush=True)
