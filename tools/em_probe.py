"""EM family timing probe (GPU box): sweeps per second of every built member on a synthetic int8 panel next to the
oracle's float-faithful restatement on one host core.  python tools/em_probe.py [n p gpu_sweeps cpu_sweeps]"""
import json
import sys
import time

import numpy as np
import torch

import bwgr_amd
from bwgr_amd import synth
from oracle import oracle as O

n, p, gs, cs = (int(v) for v in (sys.argv[1:5] + ["5000", "50000", "20", "2"][len(sys.argv) - 1:]))
X = synth.genotypes(n, p, device=0)
y = synth.scale_phenotype(synth.phenotype(X, n)).cpu().numpy()
Xh = X[:, :n].cpu().numpy().T
P = bwgr_amd.Panel(X, n=n, device=0)
out = {"n": n, "p": p}
for model in ("emRR", "emBA", "emBB", "emBC", "emBCpi", "emDE", "emBL", "emEN", "emML"):
    f = getattr(bwgr_amd, model)
    f(y, P, maxit=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); a = f(y, P, maxit=2 + gs); torch.cuda.synchronize(); t1 = time.perf_counter()
    t2 = time.perf_counter(); f(y, P, maxit=2); torch.cuda.synchronize(); t3 = time.perf_counter()
    gpu = gs / ((t1 - t0) - (t3 - t2))
    t0 = time.perf_counter(); O.em(model, y, Xh, maxit=1, flavour="f", fast=True); t1 = time.perf_counter()
    r = O.em(model, y, Xh, maxit=1 + cs, flavour="f", fast=True); t2 = time.perf_counter()
    cpu = cs / ((t2 - t1) - (t1 - t0))
    out[model] = {"gpu_sweeps_per_s": gpu, "cpu_sweeps_per_s_1core": cpu, "ratio": gpu / cpu}
P.close()
print(json.dumps(out))
