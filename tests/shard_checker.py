"""CPU checker engine for the marker-sharded driver (test infrastructure): BayesRR shard sweeps via the oracle's KMUP."""
import numpy as np
import torch
from oracle import oracle as O

f32 = np.float32
GM = O.GLOBAL_MARKER


class OracleRREngine:
    def __init__(self, Xs, y, marker0, p_total, msx_total, df, R2, seed, block):
        self.X = np.asfortranarray(Xs); self.n, self.pl = Xs.shape
        self.marker0, self.p_total, self.seed, self.block = marker0, p_total, seed, block
        self.nblocks = (self.pl + block - 1) // block
        self.xx = (self.X.astype(np.float64) ** 2).sum(0).astype(f32)
        yf = y.astype(f32)
        vy = f32(O.fvar(yf)); self.df = f32(df)
        self.Sb = f32(f32(R2) * self.df * vy / f32(msx_total)); self.Se = f32((f32(1) - f32(R2)) * self.df * vy)
        self.mu = f32(yf.astype(np.float64).mean()); self.ve = vy; self.vb = self.Sb; self.lam = f32(self.ve / self.vb)
        self.e = torch.from_numpy((yf - self.mu).astype(np.float64))
        self.b = np.zeros(self.pl, f32); self.it = 0; self.B = np.zeros(self.pl); self.VE = 0.0; self.nacc = 0

    def sweep_blocks(self, lo, hi):
        a, z = lo * self.block, min(self.pl, hi * self.block)
        o = O.kmup(self.X[:, a:z], self.b[a:z], np.ones(z - a), self.xx[a:z], self.e.numpy(), np.full(z - a, self.lam), self.ve, 0.0,
                   seed=self.seed, it=self.it, marker0=self.marker0 + a)
        self.b[a:z] = o["b"]; self.e = torch.from_numpy(o["e"].astype(np.float64))

    def residual(self):
        return self.e

    def set_residual(self, t):
        self.e = t.clone()

    def sums(self):
        return torch.tensor([float(self.pl), float((self.b.astype(np.float64) ** 2).sum())], dtype=torch.float64)

    def end_iteration(self, s):
        b2 = f32(s[1].item()) if s is not None else f32((self.b.astype(np.float64) ** 2).sum())
        e = self.e.numpy(); n = self.n
        eM = f32(np.float64(f32(e.mean())) + np.float64(f32(np.sqrt(f32(self.ve / f32(n))))) * O.variate(self.seed, "normal", GM, self.it, O.PURPOSE["G_MU"]))
        self.mu = f32(self.mu + eM); e = e - np.float64(eM)
        ss = f32((e ** 2).sum())
        self.ve = f32(np.float64(f32(ss + self.Se)) / O.variate(self.seed, "chisq", GM, self.it, O.PURPOSE["G_VE"], nu=float(f32(n + self.df))))
        self.vb = f32(np.float64(f32(b2 + self.Sb)) / O.variate(self.seed, "chisq", GM, self.it, O.PURPOSE["G_VB"], nu=float(f32(self.p_total + self.df))))
        self.lam = f32(self.ve / self.vb)
        self.e = torch.from_numpy(e); self.it += 1
        self.B += self.b; self.VE += float(self.ve); self.nacc += 1
