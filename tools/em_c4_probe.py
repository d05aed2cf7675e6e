"""EM family at C4 size (GPU only): ms per sweep of an affine, a soft-selection and a soft-threshold member (slope over three
run lengths; the per-call constant -- an 11 GB scratch panel allocated and freed -- is reported too)."""
import time, json
import torch
import bwgr_amd
from bwgr_amd import synth
n, p = 10000, 1000000
X = synth.genotypes(n, p, device=0); y = synth.scale_phenotype(synth.phenotype(X, n)).cpu().numpy()
P = bwgr_amd.Panel(X, n=n, device=0); del X
out = {}
for model in ("emRR", "emBB", "emEN", "emBCpi", "lasso"):
    f = getattr(bwgr_amd, model); ts = {}
    for it in (3, 8, 13, 3, 8, 13):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(y, P, maxit=it); torch.cuda.synchronize(); ts.setdefault(it, []).append(time.perf_counter() - t0)
    t3, t8, t13 = (min(ts[k]) for k in (3, 8, 13))
    out[model] = {"ms_per_sweep": round(1e3 * (t13 - t3) / 10, 1), "ms_per_call_constant": round(1e3 * (t3 - 3 * (t13 - t3) / 10), 1)}
print(json.dumps({"em_at_c4": out}))
