"""Shards of the partitioned sampler side by side on ONE GPU (Group(devices=[0] * S), implicitly centred int8 shards): iterations per second by
the number of shards, at C4's shape.  Usage: python tools/intra_shard_probe.py [S ...]   (AB_P = markers, default 1 000 000)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bwgr_amd
from bwgr_amd import synth

n, p = 10000, int(os.environ.get("AB_P", "1000000"))
K, W = int(os.environ.get("AB_K", "10")), 3
X = synth.genotypes(n, p)
y = synth.scale_phenotype(synth.phenotype(X, n))
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5]:
    try:
        g = bwgr_amd.Group("BayesB", y, X, devices=[0] * S, it=W + K, bi=W, pi=0.99, seed=synth.SEED, centre=True, n=n, markers_per_sync=int(os.environ.get("AB_MPS", "0")))
        g.run(W); g.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.run(K); g.sync(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        r = g.result()
        info = g.info()
        print(json.dumps({"shards": S, "iter_per_s": K / el, "ms_per_iteration": 1e3 * el / K, "frac_of_hbm_peak": (K / el) * n * p / 8e12, "rounds_per_sweep": info["rounds_per_sweep"],
                          "markers_per_round": info["markers_per_round"], "statistically_sound": r["statistically_sound"], "ve": r["ve"], "mean_d": float(r["d"].mean())}), flush=True)
        g.close()
    except Exception as ex:
        print(json.dumps({"shards": S, "error": str(ex)}), flush=True)
