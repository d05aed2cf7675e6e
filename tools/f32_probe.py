"""Diagnostic: sweep time of an fp32 (non-integer) panel next to the int8 one of the same shape."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bwgr_amd
from bwgr_amd import synth
n, p = 10000, 100000
X = synth.genotypes(n, p); y = synth.scale_phenotype(synth.phenotype(X, n))
Xf = X[:, :n].float(); Xf = Xf - Xf.mean(dim=1, keepdim=True)          # centred genotypes: not int8 any more
for name, src in (("int8", X), ("fp32 centred", Xf)):
    P = bwgr_amd.Panel(src, n=n)
    for model, pi in (("BayesB", 0.99), ("BayesA", 0.0)):
        ch = bwgr_amd.Chain(P, model, y, it=5, bi=0, pi=pi, seed=1)
        ch.run(2); ch.sync(); ch.sweep_ms(); ch.run(3); ch.sync()
        ms, nl = ch.sweep_ms()
        print("%-13s block=%3d K=%3d R=%3d  %-6s sweep %8.3f ms  %6.1f ns/marker" % (name, P.block, P.nwg, P.slab_rows, model, ms, 1e6 * ms / p), flush=True)
        ch.close()
    P.close()
