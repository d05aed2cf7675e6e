"""The oracle's outputs frozen as fixtures (tests/golden/oracle_goldens.npz, written by tests/golden/make_oracle_goldens.py).

CPU: the live oracle (oracle/bwgr_oracle.c at -O2 with -ffp-contract=off: no FMA contraction, no -march) reproduces every frozen array
BIT FOR BIT -- an edit to the oracle that moves a number fails here first, instead of silently moving the target of every GPU parity test.
GPU: the HIP path is compared with the FROZEN arrays (not the live oracle) at the parity tolerance, for the single sweeps
(src/Rcpp20260726ai.cpp:12-38, :41-77), the seven samplers' 20-iteration chains (:589-987) and the wgr() settings (R/wgr.R:2-169,
man/wgr.Rd:82).

These are the oracle's numbers, not bWGR's (the reference ships no expected outputs and cannot be built without R): parity stays "unpinned"
until tools/make_r_fixtures.R has been run by someone with R (tests/test_r_fixtures.py picks its files up when present).
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_oracle_goldens as G   # noqa: E402

from conftest import scaled_err   # noqa: E402

TOL = 1e-6


@pytest.fixture(scope="module")
def frozen():
    return np.load(G.OUT)


def test_goldens_cover_every_case(frozen):
    names = [n for n, _ in G.cases()]
    assert len(names) == len(set(names)) == 2 * (2 * 3 + 7 + 6 + 3 + 10)
    have = {k.rsplit("/", 1)[0] if not k.count("/last/") else k.split("/last/")[0] for k in frozen.files}
    assert set(names) <= have


def test_live_oracle_reproduces_the_goldens_bit_for_bit(frozen):
    checked = 0
    for name, thunk in G.cases():
        flat = {}
        G.flatten(name, thunk(), flat)
        for k, v in flat.items():
            assert k in frozen.files, k
            a, b = np.asarray(v), frozen[k]
            assert a.dtype == b.dtype and a.shape == b.shape, k
            assert np.array_equal(a, b, equal_nan=True), "%s: the live oracle no longer reproduces its frozen output (max |diff| %.3g)" % (
                k, float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0)
            checked += 1
    assert checked == len(frozen.files)


def _rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("data", ["tpod", "synth"])
@pytest.mark.parametrize("pi", [0.0, 0.3])
def test_gpu_kmup_against_the_frozen_sweeps(frozen, tpod, data, pi):
    import bwgr_amd
    if data == "tpod":
        X, y = tpod["gen"], tpod["y"]
    else:
        X, y = G.synth_small(300, 330, seed=41)
    xx, b, d, e, L = G.kmup_inputs(X, y, 5)
    g = bwgr_amd.KMUP(X, b, d, xx, e, L, 0.03, pi, seed=77, it=3)
    key = "kmup/%s/pi%.1f/w/" % (data, pi)
    assert scaled_err(g["b"], frozen[key + "b"]) < TOL and scaled_err(g["e"], frozen[key + "e"]) < TOL
    assert np.array_equal(g["d"], frozen[key + "d"])


@pytest.mark.gpu
@pytest.mark.parametrize("data", ["tpod", "synth"])
def test_gpu_kmup2_against_the_frozen_sweeps(frozen, tpod, data):
    import bwgr_amd
    if data == "tpod":
        X, y = tpod["gen"], tpod["y"]
    else:
        X, y = G.synth_small(300, 330, seed=41)
    xx, b, d, e, L = G.kmup_inputs(X, y, 5)
    key = "kmup2/%s/w/" % data
    use = frozen[key + "use"]
    g = bwgr_amd.KMUP2(X, use, b, d, xx * 0.6, e, L, 0.03, 0.3, seed=78, it=2)
    assert scaled_err(g["b"], frozen[key + "b"]) < TOL and scaled_err(g["e"], frozen[key + "e"]) < TOL
    assert np.array_equal(g["d"], frozen[key + "d"])


@pytest.mark.gpu
@pytest.mark.parametrize("model", G.SAMPLERS)
def test_gpu_chains_against_the_frozen_chains(frozen, tpod, model):
    import bwgr_amd
    X, y = tpod["gen"], tpod["y"]
    P = bwgr_amd.Panel(X)
    ch = bwgr_amd.Chain(P, model, y, it=20, bi=5, pi=0.9, df=5, R2=0.5, seed=11)
    ch.run(20)
    g = ch.result(); st = ch.state()
    ch.close(); P.close()
    key = "bayes/%s/w/" % model
    assert scaled_err(g["b"], frozen[key + "b"]) < TOL and scaled_err(g["hat"], frozen[key + "hat"]) < TOL
    assert _rel(g["ve"], frozen[key + "ve"]) < TOL and _rel(g["mu"], frozen[key + "mu"]) < TOL
    assert scaled_err(np.atleast_1d(g["vb"]), np.atleast_1d(frozen[key + "vb"])) < 5 * TOL
    if key + "d" in frozen.files:
        assert np.array_equal(g["d"], frozen[key + "d"])
    assert scaled_err(st["e"], frozen[key + "last/e"]) < TOL and scaled_err(st["b"], frozen[key + "last/b"]) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", G.WGR_SETTINGS)
def test_gpu_wgr_against_the_frozen_runs(frozen, tpod, name, kw):
    import bwgr_amd
    kw2 = dict(it=25, bi=5, seed=21); kw2.update(kw)
    g = bwgr_amd.wgr(tpod["y"], tpod["gen"], **kw2)
    key = "wgr/%s/w/" % name
    assert scaled_err(g["b"], frozen[key + "b"]) < TOL and scaled_err(g["hat"], frozen[key + "hat"]) < TOL
    assert _rel(g["Ve"], frozen[key + "Ve"]) < TOL and _rel(g["mu"], frozen[key + "mu"]) < TOL
    assert scaled_err(np.atleast_1d(g["Vb"]), np.atleast_1d(frozen[key + "Vb"])) < 5 * TOL
    assert scaled_err(g["d"], frozen[key + "d"]) < TOL
