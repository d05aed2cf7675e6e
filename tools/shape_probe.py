"""Diagnostic: sweep time for a few panel shapes (BayesB pi = 0.99 and BayesA), to catch pathologies away from the C4 geometry."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bwgr_amd
from bwgr_amd import synth
shapes = [(2000, 200000), (10000, 200000), (25000, 200000), (50000, 100000), (10000, 200000, 64)]
for sh in shapes:
    n, p = sh[0], sh[1]
    blk = sh[2] if len(sh) > 2 else 0
    X = synth.genotypes(n, p); y = synth.scale_phenotype(synth.phenotype(X, n))
    P = bwgr_amd.Panel(X, n=n, block=blk); del X
    for model, pi in (("BayesB", 0.99), ("BayesA", 0.0)):
        ch = bwgr_amd.Chain(P, model, y, it=6, bi=0, pi=pi, seed=1)
        ch.run(2); ch.sync(); ch.sweep_ms()
        ch.run(4); ch.sync()
        ms, nl = ch.sweep_ms()
        nb = (p + P.block - 1) // P.block
        print("n=%6d p=%7d block=%3d K=%3d R=%3d  %-6s sweep %8.3f ms  %6.2f us/block  %6.1f GB/s" % (n, p, P.block, P.nwg, P.slab_rows, model, ms, 1e3 * ms / nb, n * p / ms / 1e6), flush=True)
        ch.close()
    P.close()
