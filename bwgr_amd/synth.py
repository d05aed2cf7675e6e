"""Synthetic genotype panels of BASELINE.md section 3, generated on the GPU (torch is used for device memory only).

f_j ~ U(0.05,0.5); X_ij ~ Binomial(2,f_j) int8, column-major with ld = n rounded up to 128, uncentred; 1 % causal
markers with beta ~ N(0,1); y scaled so var(y) ~ 1 with h2 = 0.5."""
import ctypes as C
import numpy as np
import torch  # noqa: F401  (before the library is loaded: torch bundles its own HIP runtime, see _lib.lib)

from . import _lib

SEED = 20260803


def genotypes(n, p, col0=0, seed=SEED, device=0):
    """Returns a torch int8 tensor of shape (p, ld): row j = column col0+j of the panel (column-major storage)."""
    import torch
    ld = (n + 127) // 128 * 128
    X = torch.empty((p, ld), dtype=torch.int8, device="cuda:%d" % device)
    s = torch.cuda.current_stream(X.device)
    _lib.check(_lib.lib().bwgr_synth_genotypes(C.c_void_p(X.data_ptr()), n, p, ld, col0, C.c_uint64(seed), None, device,
                                                C.c_void_p(s.cuda_stream)))
    s.synchronize()
    return X


def phenotype(X, n, causal_frac=0.01, h2=0.5, seed=SEED, col0=0, p_total=None):
    """y (torch float32, n) from the (p, ld) device panel X; with marker shards pass col0/p_total and sum the returned
    genetic values across ranks before scaling (see bwgr_amd/dist.py)."""
    import torch
    p = X.shape[0]
    p_total = p if p_total is None else p_total
    rs = np.random.RandomState(seed % (2 ** 31))
    ncausal = max(1, int(p_total * causal_frac))
    idx = np.sort(rs.choice(p_total, ncausal, replace=False))
    beta = rs.normal(size=ncausal)
    mine = (idx >= col0) & (idx < col0 + p)
    g = torch.zeros(n, dtype=torch.float64, device=X.device)
    if mine.any():
        li = torch.as_tensor(idx[mine] - col0, device=X.device)
        bl = torch.as_tensor(beta[mine], dtype=torch.float64, device=X.device)
        for a in range(0, li.numel(), 2048):   # bounded temporary
            g += (X[li[a:a + 2048]][:, :n].to(torch.float64).T @ bl[a:a + 2048])
    return g


def scale_phenotype(g, h2=0.5, seed=SEED):
    import torch
    n = g.numel()
    gen = torch.Generator(device=g.device); gen.manual_seed(seed)
    gs = (g - g.mean()) / g.std()
    y = gs * (h2 ** 0.5) + torch.randn(n, dtype=torch.float64, device=g.device, generator=gen) * ((1 - h2) ** 0.5)
    return y.to(torch.float32).contiguous()
