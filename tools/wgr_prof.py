"""wgr() BRR at C4 size under rocprofv3 --kernel-trace --stats: which kernels make up an iteration."""
import sys, time
import torch
import bwgr_amd
from bwgr_amd import synth
n, p = 10000, 1000000
X = synth.genotypes(n, p, device=0); y = synth.scale_phenotype(synth.phenotype(X, n)).cpu().numpy().astype("float64")
P = bwgr_amd.Panel(X, n=n, device=0); del X
bwgr_amd.wgr(y, P, it=22, bi=2, seed=3)
torch.cuda.synchronize()
print("done")
