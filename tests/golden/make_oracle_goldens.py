#!/usr/bin/env python3
"""Generates tests/golden/oracle_goldens.npz: outputs of the CPU oracle (oracle/bwgr_oracle.c, the -O2 / -ffp-contract=off build) frozen as
fixtures, so that an edit to the oracle cannot silently move the target of the GPU parity tests (tests/test_oracle_goldens.py compares the
live oracle with this file bit for bit; the GPU tests compare the HIP path with it at the parity tolerance).

What is frozen, both flavours ("w" wide accumulators -- the GPU's target --, "f" the
reference's float types):
  * single sweeps: KMUP (src/Rcpp20260726ai.cpp:12-38) at pi = 0 and pi = 0.3, KMUP2 (:41-77) on a row subsample, on tpod and on a small
    synthetic panel with several blocks;
  * 20-iteration chains of each fused sampler BayesA / B / C / L / RR / Cpi / Dpi (:589-987) on tpod: the return list and the last state;
  * wgr() (R/wgr.R:2-169) in the five settings of man/wgr.Rd:82 plus thinning, 25 iterations on tpod;
  * the two-effect samplers BayesA2 / B2 / RR2 (:990-1218), 12 iterations;
  * the EM / Gauss-Seidel family (:80-586, :1463-1547), 6 sweeps each.

THESE ARE THE ORACLE'S NUMBERS, NOT bWGR's: the reference ships no expected outputs and cannot be built here (no R); the oracle is a
line-cited restatement on a Philox stream ("parity unpinned").  Real-bWGR fixtures come from tools/make_r_fixtures.R, run by
someone who has R.

Usage: python tests/golden/make_oracle_goldens.py   (rewrites the .npz; commit it together with any deliberate oracle change)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OUT = os.path.join(HERE, "oracle_goldens.npz")
SAMPLERS = ["BayesA", "BayesB", "BayesC", "BayesL", "BayesRR", "BayesCpi", "BayesDpi"]
WGR_SETTINGS = [("BRR", {}), ("BayesA", {"iv": True}), ("BayesB", {"iv": True, "pi": 0.5}), ("BayesC", {"pi": 0.5}), ("BayesL", {"de": True}),
                ("thin", {"th": 3, "bi": 4})]
EM = ["emRR", "emBA", "emDE", "emML", "emBB", "emBC", "emBCpi", "emBL", "emEN", "lasso"]


def synth_small(n, p, seed, h2=0.5, causal=0.05):
    """(the same generator as tests/conftest.py::synth_small, restated so that this script stands alone)"""
    rs = np.random.RandomState(seed)
    f = rs.uniform(0.05, 0.5, p)
    X = (rs.uniform(size=(n, p)) < f).astype(np.int8) + (rs.uniform(size=(n, p)) < f).astype(np.int8)
    nc = max(1, int(p * causal))
    idx = rs.choice(p, nc, replace=False)
    g = X[:, idx].astype(np.float64) @ rs.normal(size=nc)
    g = (g - g.mean()) / (g.std() + 1e-12)
    y = g * np.sqrt(h2) + rs.normal(size=n) * np.sqrt(1 - h2)
    return np.asfortranarray(X), y


def kmup_inputs(X, y, seed):
    n, p = X.shape
    rs = np.random.RandomState(seed)
    xx = (X.astype(np.float64) ** 2).sum(0)
    b = rs.normal(size=p) * 0.01
    d = np.ones(p)
    e = y - y.mean() - X.astype(np.float64) @ b
    L = np.full(p, 120.0) * rs.uniform(0.5, 2.0, p)
    return xx, b, d, e, L


def flatten(prefix, obj, out):
    if isinstance(obj, dict):
        for k, v in obj.items():
            flatten(prefix + "/" + k, v, out)
    else:
        out[prefix] = np.asarray(obj)


def cases():
    """Every frozen case as (name, thunk): the thunk calls the live oracle.  tests/test_oracle_goldens.py iterates the same list."""
    from oracle import oracle as O
    d = np.load(os.path.join(HERE, "tpod.npz"))
    ty, tX = d["y"].astype(np.float64), np.asfortranarray(d["gen"])
    sX, sy = synth_small(300, 330, seed=41)
    out = []
    for fl in ("w", "f"):
        for data, (X, y) in (("tpod", (tX, ty)), ("synth", (sX, sy))):
            xx, b, dd, e, L = kmup_inputs(X, y, 5)
            for pi in (0.0, 0.3):
                out.append(("kmup/%s/pi%.1f/%s" % (data, pi, fl),
                            lambda X=X, b=b, dd=dd, xx=xx, e=e, L=L, pi=pi, fl=fl: O.kmup(X, b, dd, xx, e, L, 0.03, pi, seed=77, it=3, flavour=fl)))
            use = O.bag_rows(123, 2, X.shape[0], int(0.6 * X.shape[0]))
            out.append(("kmup2/%s/%s" % (data, fl),
                        lambda X=X, use=use, b=b, dd=dd, xx=xx, e=e, L=L, fl=fl: dict(O.kmup2(X, use, b, dd, xx * 0.6, e, L, 0.03, 0.3, seed=78, it=2, flavour=fl), use=use)))
        for m in SAMPLERS:
            out.append(("bayes/%s/%s" % (m, fl), lambda m=m, fl=fl: O.bayes(m, ty, tX, it=20, bi=5, pi=0.9, df=5, R2=0.5, seed=11, flavour=fl)))
        for name, kw in WGR_SETTINGS:
            kw2 = dict(it=25, bi=5, seed=21); kw2.update(kw)
            out.append(("wgr/%s/%s" % (name, fl), lambda kw2=kw2, fl=fl: O.wgr(ty, tX, flavour=fl, **kw2)))
        X2 = np.asfortranarray(tX[:, ::-1][:, :200])
        for m in ("BayesA2", "BayesB2", "BayesRR2"):
            out.append(("bayes2/%s/%s" % (m, fl), lambda m=m, fl=fl, X2=X2: O.bayes2(m, ty, tX, X2, it=12, bi=3, pi=0.8, seed=31, flavour=fl)))
        for m in EM:
            out.append(("em/%s/%s" % (m, fl), lambda m=m, fl=fl: O.em(m, ty, tX, maxit=6, flavour=fl)))
    return out


def main():
    flat = {}
    for name, thunk in cases():
        flatten(name, thunk(), flat)
    np.savez_compressed(OUT, **flat)
    print("%s: %d arrays, %.1f KB" % (OUT, len(flat), os.path.getsize(OUT) / 1024.0))


if __name__ == "__main__":
    main()
