import sys, time, torch
sys.path.insert(0, '.')
import bwgr_amd
from bwgr_amd import synth
n, p = 10000, 1000000
X = synth.genotypes(n, p, device=0); y = synth.scale_phenotype(synth.phenotype(X, n))
P = bwgr_amd.Panel(X, n=n, device=0); del X
for model, its, pi in (("BayesB", 600, 0.99), ("BayesB", 200, 0.95), ("BayesCpi", 120, 0.5), ("BayesA", 150, 0.0)):
    ch = bwgr_amd.Chain(P, model, y, it=its, bi=its // 3, pi=pi, df=5, R2=0.5, seed=11)
    t0 = time.perf_counter(); ch.run(its); ch.sync(); t1 = time.perf_counter()
    st = ch.state(); r = ch.result()
    print(model, pi, "%d it in %.1f s = %.1f it/s" % (its, t1 - t0, its / (t1 - t0)), "ve %.4f mean_d %.4f h2 %.3f redo %d" % (st["ve"], float(st["d"].mean()), float(r.get("h2", float("nan"))), ch.redo_count()), flush=True)
    ch.close()
P.close()
