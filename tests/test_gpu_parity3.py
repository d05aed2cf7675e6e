"""GPU parity tests, third file (round 3).

* the fixed-point engines leaving their range: the sweep is redone on the fp64 residual and the chain stays the oracle's
  (the reference's update, src/Rcpp20260726ai.cpp:681, cannot fail)
* KMUP calls whose residual is zero or tiny against the steps (the affine engine's scale used to come from max|e| alone)
* k_sweep4 (opt-in), both sequencer forms, against the oracle
* the parity matrix at the SHIPPED engine threshold (the suite otherwise forces k_sweep3 everywhere)
"""
import numpy as np
import pytest

from conftest import scaled_err, synth_small

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-300)


@pytest.mark.parametrize("model,pi", [("BayesB", 0.9), ("BayesCpi", 0.0), ("BayesA", 0.0), ("BayesRR", 0.0)])
def test_a_sweep_that_leaves_the_fixed_point_range_is_redone(tpod, model, pi, monkeypatch):
    """BWGR_DEBUG_SH_ADD takes fourteen bits of headroom off the fixed-point grid, so EVERY sweep of the fixed-point engines
    (k_sweep3 for the selection models, k_sweep2w's fixed-point streamers for the affine ones) raises its range flag; each is
    then redone from the state it started with on the fp64 residual.  The chain must be the oracle's, and no error surfaces."""
    import bwgr_amd
    from oracle import oracle as O
    monkeypatch.setenv("BWGR_DEBUG_SH_ADD", "14")
    X, y = tpod["gen"], tpod["y"]
    P = bwgr_amd.Panel(X)
    ch = bwgr_amd.Chain(P, model, y, it=10, bi=2, pi=pi, seed=21)
    ch.run(10)
    st = ch.state()
    res = ch.result()
    nredo = ch.redo_count()
    ch.close(); P.close()
    assert nredo == 10, "every sweep was meant to leave the range and be redone (%d of 10 were)" % nredo
    o = O.bayes(model, y, X, it=10, bi=2, pi=pi, seed=21)
    ol = o["last"]
    assert pi == 0.0 and model in ("BayesA", "BayesRR") or np.array_equal(st["d"], ol["d"])
    assert scaled_err(st["b"], ol["b"]) < TOL and scaled_err(st["e"], ol["e"]) < TOL and _rel(st["ve"], ol["ve"]) < TOL
    assert scaled_err(res["b"], o["b"]) < TOL and scaled_err(res["hat"], o["hat"]) < TOL


def test_range_recovery_on_a_larger_panel(monkeypatch):
    """The same on a panel with several slab workgroups and a ragged last block (600 x 1 100), BayesB at 5 % inclusion."""
    import bwgr_amd
    from oracle import oracle as O
    monkeypatch.setenv("BWGR_DEBUG_SH_ADD", "14")
    X, y = synth_small(600, 1100, seed=12)
    P = bwgr_amd.Panel(X)
    ch = bwgr_amd.Chain(P, "BayesB", y, it=6, bi=1, pi=0.95, seed=3)
    ch.run(6)
    st = ch.state()
    nredo = ch.redo_count()
    ch.close(); P.close()
    assert nredo == 6
    o = O.bayes("BayesB", y, X, it=6, bi=1, pi=0.95, seed=3)["last"]
    assert np.array_equal(st["d"], o["d"])
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL


@pytest.mark.parametrize("pi", [0.0, 0.3])
@pytest.mark.parametrize("escale", [0.0, 1e-12])
def test_kmup_with_a_zero_or_tiny_residual(tpod, pi, escale):
    """KMUP(X, b, d, xx, e, L, Ve, pi) is a legal call with e = 0 (or e far below sqrt(Ve / (xx + L)) |x|): the reference returns a
    result (src/Rcpp20260726ai.cpp:12-38).  The fixed-point grid is sized by the steps as well as by the residual, and where
    that is still not enough the sweep is redone on the fp64 residual."""
    import bwgr_amd
    from oracle import oracle as O
    X = tpod["gen"]
    n, p = X.shape
    rs = np.random.RandomState(4)
    xx = (X.astype(np.float64) ** 2).sum(0)
    b = rs.normal(size=p) * 0.02
    d = np.ones(p)
    e = rs.normal(size=n) * escale
    L = np.full(p, 200.0) * rs.uniform(0.5, 2.0, p)
    g = bwgr_amd.KMUP(X, b, d, xx, e, L, 0.04, pi, seed=17, it=2)
    o = O.kmup(X, b, d, xx, e, L, 0.04, pi, seed=17, it=2)
    assert scaled_err(g["b"], o["b"]) < TOL and scaled_err(g["e"], o["e"]) < TOL
    assert np.array_equal(g["d"], o["d"])


@pytest.mark.parametrize("seq", ["1", "2"])
@pytest.mark.parametrize("model,pi,data", [("BayesB", 0.9, "tpod"), ("BayesCpi", 0.0, "tpod"), ("BayesB", 0.97, "synth"), ("BayesDpi", 0.0, "synth")])
def test_sweep4_against_the_oracle(tpod, model, pi, data, seq, monkeypatch):
    """k_sweep4 (opt-in, BWGR_SWEEP4=1): the super-block streamers with either sequencer form -- the token walk over eight waves
    (BWGR_SEQ4=1) and the chain wave with helpers (BWGR_SEQ4=2) -- run the oracle's chain: sparse and dense inclusion, a ragged last
    quad (tpod: three blocks; synth: 1 000 x 1 700, fourteen blocks, eight slab streamers)."""
    import bwgr_amd
    from oracle import oracle as O
    monkeypatch.setenv("BWGR_SWEEP4", "1")
    monkeypatch.setenv("BWGR_SEQ4", seq)
    if data == "tpod":
        X, y = tpod["gen"], tpod["y"]
    else:
        X, y = synth_small(1000, 1700, seed=31)
    P = bwgr_amd.Panel(X)
    ch = bwgr_amd.Chain(P, model, y, it=8, bi=2, pi=pi, seed=9)
    ch.run(8)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes(model, y, X, it=8, bi=2, pi=pi, seed=9)["last"]
    assert np.array_equal(st["d"], o["d"])
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and _rel(st["ve"], o["ve"]) < TOL


@pytest.mark.parametrize("model,pi", [("BayesB", 0.9), ("BayesB", 0.99), ("BayesC", 0.95), ("BayesCpi", 0.0), ("BayesDpi", 0.0)])
def test_short_chains_under_the_shipped_engine_threshold(tpod, model, pi, monkeypatch):
    """The suite forces k_sweep3 for every selection sweep (conftest: BWGR_ENG3_THR=1).  Here the variable is removed, so the device
    picks the engine per sweep exactly as a user's run does (k_sweep3 below 3 % of the markers in the model, k_sweep2 above): the chain
    is the oracle's whichever engine takes which sweep."""
    import bwgr_amd
    from oracle import oracle as O
    monkeypatch.delenv("BWGR_ENG3_THR", raising=False)
    X, y = tpod["gen"], tpod["y"]
    P = bwgr_amd.Panel(X)
    ch = bwgr_amd.Chain(P, model, y, it=12, bi=2, pi=pi, seed=33)
    ch.run(12)
    st = ch.state()
    nredo = ch.redo_count()
    ch.close(); P.close()
    assert nredo == 0      # (an ordinary chain never leaves the range)
    o = O.bayes(model, y, X, it=12, bi=2, pi=pi, seed=33)["last"]
    assert np.array_equal(st["d"], o["d"])
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and _rel(st["ve"], o["ve"]) < TOL
