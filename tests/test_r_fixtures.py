"""Real-bWGR fixtures, when someone with R has produced them (tools/make_r_fixtures.R -> tests/golden/r_fixtures/): the CPU oracle in its R-stream
mode (oracle/bwgr_rstream.h: set.seed + Mersenne-Twister + inversion normals + nmath's rgamma / rbinom, drawn in the reference's own order --
the extra normal only on rejection, src/Rcpp20260726ai.cpp:678) must reproduce what the reference printed.  This is the test that would turn
"parity unpinned" into "pinned"; without the directory it is skipped (there is no R in the build image).

Tolerances: the faithful flavour ("f") keeps the reference's float types, but Eigen's summation order inside dot / squaredNorm is not the oracle's:
single sweeps are compared at 1e-5 of the vector's scale, 20-iteration chains at 1e-3 (a flipped inclusion decision would show as a gross
difference and fails)."""
import os

import numpy as np
import pytest

from conftest import scaled_err

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "r_fixtures")
pytestmark = pytest.mark.skipif(not os.path.isdir(FIX), reason="no real-bWGR fixtures (run tools/make_r_fixtures.R where R and bWGR are installed)")


def _get(case, name):
    return np.loadtxt(os.path.join(FIX, case, name + ".txt"), ndmin=1)


def _kmup_inputs(tpod):
    X, y = tpod["gen"], tpod["y"]
    p = X.shape[1]
    j = np.arange(1, p + 1, dtype=np.float64)
    b0 = 0.01 * np.sin(j); d0 = np.ones(p); xx = (X.astype(np.float64) ** 2).sum(0)
    e0 = y - y.mean() - X.astype(np.float64) @ b0
    L = 120.0 * (1.0 + 0.5 * np.cos(j))
    return X, b0, d0, xx, e0, L


@pytest.mark.parametrize("case,pi", [("kmup_pi0", 0.0), ("kmup_pi03", 0.3)])
def test_kmup_against_real_bwgr(tpod, case, pi):
    from oracle import oracle as O
    X, b0, d0, xx, e0, L = _kmup_inputs(tpod)
    O.rstream_seed(77, "f")
    # stable=0: the literal cj / (cj + dj) of src/Rcpp20260726ai.cpp:25-27, as the reference evaluates it
    o = O.kmup(X, b0, d0, xx, e0, L, 0.03, pi, seed=0, rng_mode=O.RSTREAM, stable=0, flavour="f")
    assert np.array_equal(o["d"], _get(case, "d"))
    assert scaled_err(o["b"], _get(case, "b")) < 1e-5 and scaled_err(o["e"], _get(case, "e")) < 1e-5


@pytest.mark.parametrize("model", ["BayesA", "BayesB", "BayesC", "BayesL", "BayesRR", "BayesCpi", "BayesDpi"])
def test_samplers_against_real_bwgr(tpod, model):
    from oracle import oracle as O
    O.rstream_seed(11, "f")
    o = O.bayes(model, tpod["y"], tpod["gen"], it=20, bi=5, pi=0.9, df=5, R2=0.5, seed=0, rng_mode=O.RSTREAM, flavour="f")
    for f in ("b", "hat"):
        assert scaled_err(o[f], _get(model, f)) < 1e-3, f
    for f in ("mu", "ve", "h2"):
        assert abs(float(o[f]) - float(_get(model, f)[0])) < 1e-3 * max(abs(float(_get(model, f)[0])), 1e-3), f
    if "d" in o:
        assert scaled_err(o["d"], _get(model, "d")) < 1e-3


@pytest.mark.parametrize("name,kw", [("BRR", {}), ("BayesA", {"iv": True}), ("BayesB", {"iv": True, "pi": 0.5}), ("BayesC", {"pi": 0.5}),
                                     ("BayesL", {"de": True}), ("thin", {"th": 3, "bi": 4})])
def test_wgr_against_real_bwgr(tpod, name, kw):
    from oracle import oracle as O
    a = dict(it=25, bi=5); a.update(kw)
    O.rstream_seed(21, "f")
    o = O.wgr(tpod["y"], tpod["gen"], seed=0, rng_mode=O.RSTREAM, stable=0, flavour="f", **a)
    case = "wgr_" + name
    assert scaled_err(o["b"], _get(case, "b")) < 1e-3 and scaled_err(o["hat"], _get(case, "hat")) < 1e-3
    assert abs(o["Ve"] - float(_get(case, "Ve")[0])) < 1e-3 * abs(float(_get(case, "Ve")[0]))
