"""GPU parity tests, second file: the configurations and boundary entries round 1 left uncovered.

* KMUP2 as a standalone entry (src/Rcpp20260726ai.cpp:41-77)
* n > 16 383 (the 16-bit Gram copies can no longer be taken for granted), many slab workgroups, > 2 q feeders: the shape of
  BASELINE config 5 (n = 50 000, BayesCpi) at a p the oracle finishes in seconds
* full-size BASELINE config 2 (5 000 x 50 000 BayesA) against the oracle
* the GPU against the oracle's FLOAT-FAITHFUL flavour (the restatement of the reference's own types), with the bound
  that is actually measured
* the abort path (bounded spins -> BWGR_ETIMEOUT -> the panel stays usable)
"""
import os
import numpy as np
import pytest

from conftest import scaled_err, synth_small

pytestmark = pytest.mark.gpu
TOL = 1e-6
ALL_MODELS = ["BayesA", "BayesB", "BayesC", "BayesL", "BayesRR", "BayesCpi", "BayesDpi"]


def _rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-300)


@pytest.mark.parametrize("pi", [0.0, 0.3])
@pytest.mark.parametrize("repeats", [False, True])
def test_kmup2_tpod(tpod, pi, repeats):
    """KMUP2(X,Use,b,d,xx,E,L,Ve,pi): the sweep on a row subsample, with the reference's `+ b0` numerator and
    `xx*bg + L` denominator (:47, :59); Use as wgr passes it (sorted, 0-based; with rp = TRUE rows repeat)."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = tpod["gen"], tpod["y"]
    n, p = X.shape
    rs = np.random.RandomState(9)
    use = np.sort(rs.choice(n, 120, replace=repeats)).astype(np.int32)
    xx = (X.astype(np.float64) ** 2).sum(0) * (120.0 / n)        # wgr passes colSums(X^2) * bag
    b = rs.normal(size=p) * 0.01
    d = np.ones(p)
    E = y - y.mean() - X.astype(np.float64) @ b
    L = np.full(p, 120.0) * rs.uniform(0.5, 2.0, p)
    g = bwgr_amd.KMUP2(X, use, b, d, xx, E, L, 0.03, pi, seed=78, it=4)
    o = O.kmup2(X, use, b, d, xx, E, L, 0.03, pi, seed=78, it=4)
    assert g["e"].shape == (120,)
    assert scaled_err(g["b"], o["b"]) < TOL and scaled_err(g["e"], o["e"]) < TOL
    assert np.array_equal(g["d"], o["d"])


def test_kmup2_rejects_rows_outside_the_panel(tpod):
    import bwgr_amd
    X = tpod["gen"]; n, p = X.shape
    z = np.zeros(p)
    with pytest.raises(bwgr_amd.BwgrError):
        bwgr_amd.KMUP2(X, [0, 1, n], z, z + 1, z + 1, np.zeros(n), z + 1, 1.0, 0.0, seed=1)


def test_wgr_bag_above_one_needs_replacement(tpod):
    """sample(n, n*bag, FALSE) with bag > 1 is an error in R (R/wgr.R:68); the library must refuse it before touching memory."""
    import bwgr_amd
    with pytest.raises(bwgr_amd.BwgrError):
        bwgr_amd.wgr(tpod["y"], tpod["gen"], it=3, bi=1, bag=1.5, rp=False, seed=1)
    g = bwgr_amd.wgr(tpod["y"], tpod["gen"], it=3, bi=1, bag=1.5, rp=True, seed=1)    # with replacement it is legal
    assert np.isfinite(g["hat"]).all()


def test_panel_refuses_integers_that_do_not_fit_int8(tpod):
    import bwgr_amd
    X = tpod["gen"].astype(np.int32).copy(); X[3, 5] = 200
    with pytest.raises(ValueError):
        bwgr_amd.Panel(X, as_int8=True)
    P = bwgr_amd.Panel(X)                       # default: staged as float32, value kept
    xx, _, _ = P.stats()
    assert xx[5] == np.float32((X[:, 5].astype(np.float64) ** 2).sum())
    P.close()


def test_panel_destroy_waits_for_its_chains(tpod):
    import bwgr_amd
    P = bwgr_amd.Panel(tpod["gen"])
    ch = bwgr_amd.Chain(P, "BayesRR", tpod["y"], it=2, bi=0, seed=1)
    with pytest.raises(bwgr_amd.BwgrError):
        P.close()                                # a chain is alive: refused, handle still valid
    ch.run(2); ch.sync(); ch.close()
    P.close()


@pytest.mark.parametrize("model", ["BayesCpi", "BayesB"])
@pytest.mark.parametrize("n,p,env", [(20000, 640, {}), (50000, 384, {}), (50000, 384, {"BWGR_GRAM16": "0"}),
                                     (20000, 640, {"BWGR_SWEEP": "2"}), (50000, 384, {"BWGR_SWEEP": "2"}),
                                     (50000, 384, {"BWGR_SWEEP": "2", "BWGR_GRAM16": "0"})])
def test_large_n_against_oracle(model, n, p, env, monkeypatch):
    """BASELINE config 5's shape (n = 50 000, BayesCpi: dense inclusion) and n = 20 000, at a p the oracle finishes in seconds:
    79-196 slab workgroups (more than nine for the first time); the trajectory engine (k_sweep3, default) and k_sweep2 with its
    3-6 q feeders; 16-bit and 32-bit Gram entries."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(n, p, seed=n // 100 + p)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    P = bwgr_amd.Panel(X)
    pipe = P.pipeline(True)
    assert P.nwg >= 79, (P.nwg, pipe)
    if env.get("BWGR_SWEEP") == "2":
        assert pipe["generation"] == 2 and pipe["feeders"] >= 3, pipe
    else:
        assert pipe["generation"] == 3 and pipe["lag"] >= 2, pipe
    if "BWGR_GRAM16" in env:
        assert pipe["gram_bits"] == 32
    ch = bwgr_amd.Chain(P, model, y, it=3, bi=0, pi=0.9, seed=5)
    ch.run(3)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes(model, y, X, it=3, bi=0, pi=0.9, seed=5)["last"]
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and _rel(st["ve"], o["ve"]) < TOL
    assert np.array_equal(st["d"], o["d"])


def test_gram_entries_beyond_16_bits_fall_back_naturally():
    """Markers fixed at 2 over 20 000 rows: X_j'X_k = 80 000 > 65 535, so the device-side range check must fail and the
    32-bit staging take over by itself (no environment switch)."""
    import bwgr_amd
    from oracle import oracle as O
    n, p = 20000, 256
    X, y = synth_small(n, p, seed=31)
    X = np.array(X); X[:, 10] = 2; X[:, 11] = 2; X[:, 140] = 2
    X = np.asfortranarray(X)
    P = bwgr_amd.Panel(X)
    assert P.pipeline(True)["gram_bits"] == 32
    ch = bwgr_amd.Chain(P, "BayesB", y, it=3, bi=0, pi=0.8, seed=6)
    ch.run(3)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes("BayesB", y, X, it=3, bi=0, pi=0.8, seed=6)["last"]
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and np.array_equal(st["d"], o["d"])


def test_full_size_c2_against_oracle():
    """BASELINE config 2 at full size: synthetic 5 000 x 50 000 int8, BayesA, two iterations against the oracle
    (src/Rcpp20260726ai.cpp:589-635); the oracle takes about a second per iteration."""
    import bwgr_amd
    from bwgr_amd import synth
    from oracle import oracle as O
    n, p = 5000, 50000
    Xd = synth.genotypes(n, p)
    y = synth.scale_phenotype(synth.phenotype(Xd, n))
    P = bwgr_amd.Panel(Xd, n=n)
    ch = bwgr_amd.Chain(P, "BayesA", y, it=2, bi=0, seed=synth.SEED)
    ch.run(2)
    st = ch.state()
    ch.close(); P.close()
    Xh = np.asfortranarray(Xd[:, :n].cpu().numpy().T.astype(np.float32))
    o = O.bayes("BayesA", y.cpu().numpy(), Xh, it=2, bi=0, seed=synth.SEED)["last"]
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and _rel(st["ve"], o["ve"]) < TOL


def test_c3_size_properties():
    """BASELINE config 3 (n = 10 000 x p = 500 000, BayesB pi = 0.99): residual identity against an independent fp64 product
    and the inclusion rate, after three iterations (the oracle cannot run this size)."""
    import torch
    import bwgr_amd
    from bwgr_amd import synth
    n, p = 10000, 500000
    X = synth.genotypes(n, p)
    y = synth.scale_phenotype(synth.phenotype(X, n))
    P = bwgr_amd.Panel(X, n=n)
    ch = bwgr_amd.Chain(P, "BayesB", y, it=3, bi=0, pi=0.99, seed=synth.SEED)
    ch.run(3)
    st = ch.state()
    ch.close(); P.close()
    b = torch.from_numpy(st["b"]).to(X.device).double()
    xb = torch.zeros(n, dtype=torch.float64, device=X.device)
    for j0 in range(0, p, 50000):
        xb += X[j0:j0 + 50000, :n].double().T @ b[j0:j0 + 50000]
    e_ref = (y.double() - st["mu"] - xb).cpu().numpy()
    assert np.abs(e_ref - st["e"]).max() < 2e-5 * np.abs(e_ref).max()
    assert 0.003 < st["d"].mean() < 0.05


def test_c5_size_properties(monkeypatch):
    """BASELINE config 5's panel on one GPU (n = 50 000 x p = 1 000 000 int8, 50 GB; 196 slab workgroups), its own model BayesCpi (dense
    inclusion: k_sweep2 by the device's choice at the shipped threshold) and the headline model BayesB pi = 0.99 on the same genotypes under
    both engines (BWGR_ENG3_THR=1: k_sweep3 takes every sweep; BWGR_SWEEP=2: k_sweep2 does): after two iterations the residual identity
    e == y - mu - X b  against an independent fp64 torch product, the inclusion rate, no range redo, and the two engines' agreement:
    decisions equal, effects to 1e-6.  (The oracle cannot run this size; the samplers follow src/Rcpp20260726ai.cpp:858-921 / :638-699.)"""
    import torch
    import bwgr_amd
    from bwgr_amd import synth
    n, p = 50000, 1000000
    X = synth.genotypes(n, p)
    y = synth.scale_phenotype(synth.phenotype(X, n))

    def xb_of(b_host):
        b = torch.from_numpy(b_host).to(X.device).double()
        xb = torch.zeros(n, dtype=torch.float64, device=X.device)
        for j0 in range(0, p, 20000):
            xb += X[j0:j0 + 20000, :n].double().T @ b[j0:j0 + 20000]
        return xb

    states = {}
    for tag, model, pi, env, gen in (("cpi", "BayesCpi", 0.5, {"BWGR_ENG3_THR": "0.03"}, 3), ("b3", "BayesB", 0.99, {"BWGR_ENG3_THR": "1"}, 3),
                                     ("b2", "BayesB", 0.99, {"BWGR_SWEEP": "2"}, 2)):
        monkeypatch.delenv("BWGR_SWEEP", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)                    # (read when the panel is made)
        P = bwgr_amd.Panel(X, n=n)
        assert P.pipeline(True)["generation"] == gen
        ch = bwgr_amd.Chain(P, model, y, it=2, bi=0, pi=pi, seed=synth.SEED)
        ch.run(2)
        st = ch.state()
        nredo = ch.redo_count()
        ch.close(); P.close()
        assert nredo == 0, tag
        e_ref = (y.double() - st["mu"] - xb_of(st["b"])).cpu().numpy()
        assert np.abs(e_ref - st["e"]).max() < 2e-5 * np.abs(e_ref).max(), tag
        states[tag] = st
    assert 0.3 < states["cpi"]["d"].mean() < 0.7
    assert 0.003 < states["b3"]["d"].mean() < 0.05
    assert np.array_equal(states["b3"]["d"], states["b2"]["d"])
    assert scaled_err(states["b3"]["b"], states["b2"]["b"]) < 1e-6 and scaled_err(states["b3"]["e"], states["b2"]["e"]) < 1e-6


# Distance of the GPU chain from the oracle's FLOAT-FAITHFUL flavour ("f": float residual, float accumulators in eight
# interleaved partial sums like Eigen's packet reduction, norms rounded to float before subtraction -- the reference's own
# types).  That flavour's results depend on the summation order at the 1e-6 level, which is why the parity target is the wide
# flavour; the bounds below are the measured distances with a margin of about two, and they are what "identical to the
# reference" can mean for this path: the north-star's 1e-6 holds against the widened restatement only.
# measured on MI355X (round 2, ten iterations, tpod and a 600 x 500 synthetic panel, all seven samplers and KMUP):
# b <= 8.0e-7, e <= 1.9e-6, ve <= 5.6e-7, inclusion decisions equal everywhere
FAITHFUL_BOUND = {"b": 2e-6, "e": 4e-6, "scalar": 2e-6}
_FLIPS = {}   # (model, data) -> (flipped inclusion decisions against the float flavour, markers)


@pytest.mark.parametrize("model", ALL_MODELS)
@pytest.mark.parametrize("data", ["tpod", "synth"])
def test_distance_to_the_float_faithful_flavour(tpod, model, data):
    import bwgr_amd
    from oracle import oracle as O
    if data == "tpod":
        X, y = tpod["gen"], tpod["y"]
    else:
        X, y = synth_small(600, 500, seed=3)
    P = bwgr_amd.Panel(X)
    ch = bwgr_amd.Chain(P, model, y, it=10, bi=2, pi=0.9, seed=21)
    ch.run(10)
    st = ch.state()
    ch.close(); P.close()
    f = O.bayes(model, y, X, it=10, bi=2, pi=0.9, seed=21, flavour="f")["last"]
    w = O.bayes(model, y, X, it=10, bi=2, pi=0.9, seed=21, flavour="w")["last"]
    # inclusion decisions of the GPU chain's last sweep against the float flavour's: a flip (the float flavour's norm cancellation lands on
    # the other side of a uniform) forks the chains, after which the values have nothing to bound -- so flips are COUNTED, and
    # test_flip_rate_against_the_float_faithful_flavour fails when they are more than rare
    selection = model in ("BayesB", "BayesC", "BayesCpi", "BayesDpi")   # (the affine samplers have no decisions: the oracle leaves d = 0)
    flips = int(np.sum(st["d"] != f["d"])) if selection else 0
    _FLIPS[(model, data)] = (flips, int(st["d"].size))
    eb, ee, ev = scaled_err(st["b"], f["b"]), scaled_err(st["e"], f["e"]), _rel(st["ve"], f["ve"])
    print("faithful-distance %s/%s: b %.2e e %.2e ve %.2e (wide: b %.2e e %.2e) flipped decisions %d of %d" % (
        model, data, eb, ee, ev, scaled_err(st["b"], w["b"]), scaled_err(st["e"], w["e"]), flips, st["d"].size))
    assert scaled_err(st["b"], w["b"]) < TOL and scaled_err(st["e"], w["e"]) < TOL
    assert not selection or np.array_equal(st["d"], w["d"])
    if flips == 0:
        assert eb < FAITHFUL_BOUND["b"] and ee < FAITHFUL_BOUND["e"] and ev < FAITHFUL_BOUND["scalar"]


def test_flip_rate_against_the_float_faithful_flavour():
    """Stated rate: at most 2 of the 14 (model, data) cases above may contain a flipped inclusion decision at all, and none may
    have flipped more than 1 % of its markers (measured in rounds 2 and 3: no flip in any case)."""
    if not _FLIPS:
        pytest.skip("the per-case tests did not run in this session")
    bad = {k: v for k, v in _FLIPS.items() if v[0] > 0}
    print("cases with flipped decisions: %d of %d %s" % (len(bad), len(_FLIPS), bad))
    assert len(bad) <= 2, bad
    assert all(v[0] <= 0.01 * v[1] for v in bad.values()), bad


@pytest.mark.parametrize("pi", [0.0, 0.3])
def test_kmup_distance_to_the_float_faithful_flavour(tpod, pi):
    import bwgr_amd
    from oracle import oracle as O
    X, y = tpod["gen"], tpod["y"]
    n, p = X.shape
    rs = np.random.RandomState(5)
    xx = (X.astype(np.float64) ** 2).sum(0)
    b = rs.normal(size=p) * 0.01
    e = y - y.mean() - X.astype(np.float64) @ b
    L = np.full(p, 120.0) * rs.uniform(0.5, 2.0, p)
    g = bwgr_amd.KMUP(X, b, np.ones(p), xx, e, L, 0.03, pi, seed=77, it=3)
    f = O.kmup(X, b, np.ones(p), xx, e, L, 0.03, pi, seed=77, it=3, flavour="f")
    eb, ee = scaled_err(g["b"], f["b"]), scaled_err(g["e"], f["e"])
    flips = int(np.sum(np.asarray(g["d"]) != np.asarray(f["d"])))
    print("faithful-distance KMUP pi=%.1f: b %.2e e %.2e flipped decisions %d of %d" % (pi, eb, ee, flips, p))
    assert flips <= 0.01 * p          # (stated rate: a flip is a uniform landing inside the float flavour's cancellation error)
    if flips == 0:
        assert eb < FAITHFUL_BOUND["b"] and ee < FAITHFUL_BOUND["e"]


def test_abort_path_reports_a_timeout_and_the_panel_survives():
    """One slab workgroup withheld (bwgr_debug_withhold): every workgroup that waits for it must reach its wall-clock bound,
    the shared abort word must end the launch, bwgr_chain_sync must return BWGR_ETIMEOUT, and the next launch on the same
    panel must succeed and give the oracle's chain."""
    import ctypes as C
    import time
    import bwgr_amd
    from bwgr_amd import _lib
    from oracle import oracle as O
    X, y = synth_small(700, 600, seed=4)
    P = bwgr_amd.Panel(X, block=64, nwg=3)
    _lib.check(_lib.lib().bwgr_debug_withhold(P._h, 1))
    ch = bwgr_amd.Chain(P, "BayesB", y, it=2, bi=0, pi=0.8, seed=2)
    t0 = time.time()
    ch.run(1)
    with pytest.raises(bwgr_amd.BwgrError) as ei:
        ch.sync()
    assert ei.value.code == 4 and time.time() - t0 < 30.0          # BWGR_ETIMEOUT, within the 4 s bound (+ slack)
    ch.close()
    _lib.check(_lib.lib().bwgr_debug_withhold(P._h, 0))
    ch = bwgr_amd.Chain(P, "BayesB", y, it=4, bi=0, pi=0.8, seed=2)
    ch.run(4)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes("BayesB", y, X, it=4, bi=0, pi=0.8, seed=2)["last"]
    assert scaled_err(st["b"], o["b"]) < TOL and np.array_equal(st["d"], o["d"])


@pytest.mark.parametrize("model,pi", [("BayesB", 0.99), ("BayesB", 0.9), ("BayesC", 0.985), ("BayesCpi", 0.0)])
def test_engine_choice_follows_the_inclusion_rate(model, pi, monkeypatch):
    """With the default threshold (2 % of markers in the model) the device picks k_sweep3 or k_sweep2 sweep by sweep from the
    chain's own inclusion rate; whichever runs, the chain is the oracle's.  BayesB pi = 0.99 stays on k_sweep3, pi = 0.9 and
    BayesCpi on k_sweep2, BayesC pi = 0.985 wanders across the threshold."""
    import bwgr_amd
    from oracle import oracle as O
    monkeypatch.setenv("BWGR_ENG3_THR", "0.02")
    X, y = synth_small(500, 3000, seed=17, causal=0.01)
    P = bwgr_amd.Panel(X)
    assert P.pipeline(True)["generation"] == 3
    ch = bwgr_amd.Chain(P, model, y, it=12, bi=2, pi=pi, seed=9)
    ch.run(12)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes(model, y, X, it=12, bi=2, pi=pi, seed=9)["last"]
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and _rel(st["ve"], o["ve"]) < TOL
    assert np.array_equal(st["d"], o["d"])


@pytest.mark.parametrize("model,pi", [("BayesB", 0.9), ("BayesRR", 0.0)])
def test_group_on_one_device_is_the_plain_chain(tpod, model, pi, monkeypatch):
    """bwgr_group_* (multi-GPU inside the library) with one device: without RCCL it IS bwgr_chain_run; with the RCCL path forced
    (a one-rank communicator, the exchange rounds and the in-place all-reduces of the residual delta and the iteration's sums)
    it must give the same chain (same decisions, effects to 1e-9; the rounds start blocks from already folded slab dots)."""
    import bwgr_amd
    X, y = tpod["gen"], tpod["y"]
    ref = getattr(bwgr_amd, model)(y, X, it=8, bi=2, **({"pi": pi} if model == "BayesB" else {}), seed=5, block=32)
    for force in ("0", "1"):
        monkeypatch.setenv("BWGR_GROUP_FORCE_COMM", force)
        g = bwgr_amd.Group(model, y, X, devices=[0], it=8, bi=2, pi=pi, seed=5, block=32, markers_per_sync=96)
        info = g.info()
        assert info["devices"] == 1 and info["rccl"] == int(force) and (force == "0" or info["rounds_per_sweep"] == 4)
        g.run(8); g.sync()
        out = g.result(); g.close()
        assert out.pop("statistically_sound") is True          # one shard: the exact chain
        assert list(out) == list(ref)
        for k in ref:
            if k == "d":
                assert np.array_equal(out[k], ref[k])
            elif np.ndim(ref[k]):
                assert scaled_err(out[k], ref[k]) < 1e-6, k
            else:
                assert _rel(out[k], ref[k]) < 1e-6, k


def test_partitioned_sampler_characterisation():
    """The marker-sharded sampler is a different chain from the reference for more than one shard (every shard sweeps against a
    residual that has not seen the other shards' updates since the last exchange round), so its parity can only be statistical.
    This test pins what IS guaranteed and records what is not:
      * one shard driven through the round API is the exact chain (asserted);
      * several shards of a p >> n panel of uncentred genotypes are NOT a sound approximation at any exchange window: all
        markers share the mean direction, every shard corrects the same stale residual mean, and the summed corrections
        overshoot (measured on MI355X, n = 800 x p = 16 384, BayesB pi = 0.95, 150 kept iterations, exact chain ve = 1.45:
        2 shards x 128 markers per round ve = 2.39, cor(hat) = 0.66; 2 x 512: 9.19, 0.50; 4 x 1024: 16.1).  The runs must
        complete and stay finite; their statistics are printed, not asserted (the asserting test is the one on CENTRED columns, where
        the sampler is sound: test_gpu_parity3.py::test_partitioned_sampler_on_centred_columns).  bench.py therefore scales over GPUs
        with replica chains, its --sharded leg centres its shards, and the library refuses several shards on uncentred columns unless
        BWGR_GROUP_ALLOW_UNCENTRED=1 (bwgr_group_create)."""
    import os
    import torch
    import bwgr_amd
    from bwgr_amd.dist import HipShardEngine, shard_bounds
    from oracle import oracle as O
    n, p, it, bi, pi = 800, 16384, 60, 10, 0.95
    X, y = synth_small(n, p, seed=23, causal=0.01)
    y = y.astype(np.float32)
    msx = float(O.stats(X)[2])

    def sharded(G, markers_per_round, seed):
        spans = [shard_bounds(p, G, r, 128) for r in range(G)]
        panels = [bwgr_amd.Panel(np.asfortranarray(X[:, lo:hi])) for lo, hi in spans]
        engs = [HipShardEngine(panels[r], "BayesB", y, it, bi, pi, 5.0, 0.5, seed, spans[r][0], p, msx) for r in range(G)]
        bps = max(1, markers_per_round // 128)
        rounds = max((e.nblocks + bps - 1) // bps for e in engs)
        for _ in range(it):
            for r in range(rounds):
                ds = [e.round_sweep(min(e.nblocks, r * bps), min(e.nblocks, (r + 1) * bps)) for e in engs]
                total = torch.stack(ds).sum(0)
                for e, dlt in zip(engs, ds):
                    dlt.copy_(total); e.round_apply(dlt)
            s_ = torch.stack([e.sums() for e in engs]).sum(0)
            for e in engs:
                e.sums().copy_(s_); e.end_iteration(e.sums())
        res = [e.chain.result() for e in engs]
        out = {"ve": res[0]["ve"], "mu": res[0]["mu"], "d": np.concatenate([r_["d"] for r_ in res]),
               "b": np.concatenate([r_["b"] for r_ in res]), "hat": res[0]["mu"] + sum(r_["hat"] - r_["mu"] for r_ in res)}
        for e in engs:
            e.chain.close()
        for P in panels:
            P.close()
        return out

    a = bwgr_amd.BayesB(y, X, it=it, bi=bi, pi=pi, seed=31)
    one = sharded(1, 1024, 31)
    assert np.array_equal(one["d"], a["d"]) and scaled_err(one["b"], a["b"]) < 1e-6 and _rel(one["ve"], a["ve"]) < 1e-6
    cases = ((2, 2048), (4, 1024))
    if os.environ.get("SHARD_SCAN"):
        cases = [(int(a_), int(b_)) for a_, b_ in (c.split(":") for c in os.environ["SHARD_SCAN"].split(","))]
    for G, mpr in cases:
        s_ = sharded(G, mpr, 31)
        print("shards %d x %d markers per round: ve %.4f (exact %.4f), mean d %.4f (exact %.4f), cor(hat) %.5f" % (
            G, mpr, s_["ve"], a["ve"], s_["d"].mean(), a["d"].mean(), np.corrcoef(s_["hat"], a["hat"])[0, 1]))
        assert np.isfinite(s_["ve"]) and np.isfinite(s_["hat"]).all() and np.isfinite(s_["b"]).all()


# ---- affine sweeps as a triangular product (k_affine_inv + k_sweep2w, bwgr_amd/csrc/sweep2w.hip.h) ----
@pytest.mark.parametrize("model", ["BayesA", "BayesRR"])
def test_affine_winv_long_chain_stays_on_the_oracle(model):
    """The product sequencer feeds the UN-rounded draws of a block forward (the serial recurrence and the oracle feed the
    float-rounded ones): a rounding-level difference per sweep.  No accept / reject behind it, the sweep is a contraction:
    200 iterations later the chain still sits on the oracle's to the same 1e-6 (reference recurrences:
    /root/reference/src/Rcpp20260726ai.cpp:612-619 BayesA, :833-838 BayesRR)."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(700, 1000, seed=23)
    P = bwgr_amd.Panel(X)
    assert P.pipeline(False)["generation"] == 4 and P.pipeline(False)["lag"] == int(os.environ.get("BWGR_WLAG", "4"))
    ch = bwgr_amd.Chain(P, model, y, it=200, bi=50, seed=4)
    ch.run(200)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes(model, y, X, it=200, bi=50, seed=4)["last"]
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and _rel(st["ve"], o["ve"]) < TOL


@pytest.mark.parametrize("env", [{"BWGR_WINV": "0"}, {"BWGR_WLAG": "2"}, {"BWGR_WLAG": "3"}, {"BWGR_WPF": "0"}, {"BWGR_WPF": "1", "BWGR_WAHEAD": "2"},
                                 {"BWGR_WFX": "0"}, {"BWGR_WFX": "0", "BWGR_WLAG": "2"}, {"BWGR_WNQ": "2"}, {"BWGR_WNQ": "4"}])
def test_affine_winv_variants_agree(env, monkeypatch):
    """The serial sequencer (BWGR_WINV=0), k_sweep2's streamers under the product sequencer (BWGR_WFX=0: fp64 residual and one
    word per streamer instead of the fixed-point residual and atomic sums), shallower pipelines (cross terms of one or two blocks back instead of three) and the
    L2 prefetch workgroups switched off or down: the oracle's chain every time, and the default's to 2e-7 (bit for bit where only
    the prefetch changed)."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(900, 1300, seed=29)
    out = {}
    for name, e in (("default", {}), ("variant", env)):
        for k in ("BWGR_WINV", "BWGR_WLAG", "BWGR_WPF", "BWGR_WAHEAD", "BWGR_WFX", "BWGR_WNQ"):
            monkeypatch.delenv(k, raising=False)
        for k, v in e.items():
            monkeypatch.setenv(k, v)
        P = bwgr_amd.Panel(X)
        gen = P.pipeline(False)["generation"]
        assert gen == (2 if e.get("BWGR_WINV") == "0" else 4)
        ch = bwgr_amd.Chain(P, "BayesA", y, it=10, bi=2, seed=8)
        ch.run(10)
        out[name] = ch.state()
        ch.close(); P.close()
    o = O.bayes("BayesA", y, X, it=10, bi=2, seed=8)["last"]
    for name in out:
        assert scaled_err(out[name]["b"], o["b"]) < TOL and scaled_err(out[name]["e"], o["e"]) < TOL
    # (another depth means other partial sums of the same numbers, and a float rounding of a draw flips here and there)
    # (the streamers' slab-dot sums are exact integers: any number of copies of the sums gives the same bits)
    tol = 0.0 if set(env) <= {"BWGR_WPF", "BWGR_WAHEAD", "BWGR_WNQ"} else 2e-7
    assert scaled_err(out["default"]["b"], out["variant"]["b"]) <= tol


def test_affine_winv_falls_back_on_signed_genotypes():
    """Centred genotype codes {-1, 0, 1} give negative Gram entries: no 16-bit planes, the serial sequencer takes the affine
    sweeps (and says so)."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(400, 600, seed=31)
    Xc = (X.astype(np.int16) - 1).astype(np.int8)
    P = bwgr_amd.Panel(Xc)
    assert P.pipeline(False)["generation"] == 2
    ch = bwgr_amd.Chain(P, "BayesRR", y, it=6, bi=1, seed=3)
    ch.run(6)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes("BayesRR", y, Xc, it=6, bi=1, seed=3)["last"]
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL


def test_affine_winv_forty_slabs():
    """n = 10 000 rows are 40 slab workgroups: more slab dots per marker than the four polling waves request early (2 x 16), the
    rest are the product waves' (S2WPollX)."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(10000, 700, seed=37)
    P = bwgr_amd.Panel(X)
    assert P.nwg == 40 and P.pipeline(False)["generation"] == 4
    ch = bwgr_amd.Chain(P, "BayesA", y, it=5, bi=1, seed=12)
    ch.run(5)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes("BayesA", y, X, it=5, bi=1, seed=12)["last"]
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and _rel(st["ve"], o["ve"]) < TOL


@pytest.mark.parametrize("lag", ["2", "3", "4"])
@pytest.mark.parametrize("model,pi", [("BayesCpi", 0.0), ("BayesDpi", 0.0), ("BayesB", 0.6), ("BayesC", 0.7)])
def test_dense_inclusion_rounds_are_the_same_chain(model, pi, lag, monkeypatch):
    """Selection sweeps with 30-50 % of the markers in the model (k_sweep2; a round per included marker, decided by lane_quick's two
    compares on the residual dot with lane_accept behind them for the sliver between the radii), at every pipeline depth: the
    oracle's chain, identical decisions (/root/reference/src/Rcpp20260726ai.cpp:884-909 BayesCpi, :950-975 BayesDpi)."""
    import bwgr_amd
    from oracle import oracle as O
    monkeypatch.setenv("BWGR_LAG", lag)
    monkeypatch.setenv("BWGR_ENG3_THR", "0.02")
    X, y = synth_small(600, 1500, seed=41)
    P = bwgr_amd.Panel(X)
    ch = bwgr_amd.Chain(P, model, y, it=10, bi=2, pi=pi, seed=6)
    ch.run(10)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes(model, y, X, it=10, bi=2, pi=pi, seed=6)["last"]
    assert np.array_equal(st["d"], o["d"])
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and _rel(st["ve"], o["ve"]) < TOL


@pytest.mark.parametrize("model,pi", [("BayesB", 0.97), ("BayesCpi", 0.0)])
def test_long_chains_on_a_mid_size_panel(model, pi, monkeypatch):
    """150 iterations at n = 2000 x p = 16 000 (125 blocks, 8 slabs): k_sweep3 with its LDS-staged Gram blocks and deferred row
    requests (BayesB) and the dense selection sweeps of k_sweep2 (BayesCpi), hundreds of thousands of block hand-offs each, against
    the oracle: identical decisions, effects and residual to 1e-6."""
    import bwgr_amd
    from oracle import oracle as O
    monkeypatch.setenv("BWGR_ENG3_THR", "0.05")
    X, y = synth_small(2000, 16000, seed=43, causal=0.01)
    P = bwgr_amd.Panel(X)
    ch = bwgr_amd.Chain(P, model, y, it=150, bi=50, pi=pi, seed=21)
    ch.run(150)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes(model, y, X, it=150, bi=50, pi=pi, seed=21)["last"]
    assert np.array_equal(st["d"], o["d"])
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and _rel(st["ve"], o["ve"]) < TOL


@pytest.mark.parametrize("model,pi", [("BayesA", 0.0), ("BayesB", 0.9)])
@pytest.mark.parametrize("scale", [1e-6, 1e5])
def test_fixed_point_engines_follow_the_phenotype_scale(model, pi, scale):
    """k_sweep3 and the affine engine's streamers keep the residual on a per-sweep fixed-point grid chosen from the residual's own
    largest exponent: phenotypes a million times smaller or a hundred thousand times larger than unit variance give the same chain,
    scaled (no range flag, same parity)."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(500, 900, seed=47)
    ys = (y * scale).astype(y.dtype)
    P = bwgr_amd.Panel(X)
    ch = bwgr_amd.Chain(P, model, ys, it=12, bi=2, pi=pi, seed=5)
    ch.run(12)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes(model, ys, X, it=12, bi=2, pi=pi, seed=5)["last"]
    assert pi == 0.0 or np.array_equal(st["d"], o["d"])
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and _rel(st["ve"], o["ve"]) < TOL


@pytest.mark.parametrize("models", [("BayesB", "BayesB"), ("BayesB", "BayesC"), ("BayesDpi", "BayesB")])
def test_chain_pairs_share_the_streamers_and_nothing_else(models, monkeypatch):
    """bwgr_chain_run_pair (k_sweep3p): two chains of one resident panel on one set of streamer workgroups -- chain 1's digits ride in
    the idle columns of chain 0's MFMA products.  Each chain is bit for bit the chain it is alone (same fixed-point arithmetic, exact
    integer slab-dot sums), whatever the partner's model or seed; mixing paired and single iterations changes nothing either."""
    import bwgr_amd
    monkeypatch.setenv("BWGR_ENG3_THR", "1")
    X, y = synth_small(700, 2600, seed=53, causal=0.02)
    P = bwgr_amd.Panel(X)
    Q = P.clone()
    solo = []
    for h, mdl, seed in ((P, models[0], 3), (Q, models[1], 4)):
        ch = bwgr_amd.Chain(h, mdl, y, it=9, bi=2, pi=0.95, seed=seed)
        ch.run(9)
        solo.append(ch.state()); ch.close()
    c0 = bwgr_amd.Chain(P, models[0], y, it=9, bi=2, pi=0.95, seed=3)
    c1 = bwgr_amd.Chain(Q, models[1], y, it=9, bi=2, pi=0.95, seed=4)
    c0.run_pair(c1, 4)
    c0.run(1); c1.run(1)
    c0.run_pair(c1, 4)
    s0, s1 = c0.state(), c1.state()
    c0.close(); c1.close(); Q.close(); P.close()
    for got, want in ((s0, solo[0]), (s1, solo[1])):
        assert np.array_equal(got["d"], want["d"]) and np.array_equal(got["b"], want["b"]) and np.array_equal(got["e"], want["e"])
        assert got["ve"] == want["ve"]


def test_chain_pairs_abort_path(monkeypatch):
    """A streamer workgroup of a pair's launch withheld: both chains' sequencers and the remaining streamers must reach their
    wall-clock bound, both chains report BWGR_ETIMEOUT, and the next paired launch on the same handles succeeds and gives the
    chains they are alone."""
    import time
    import bwgr_amd
    from bwgr_amd import _lib
    monkeypatch.setenv("BWGR_ENG3_THR", "1")
    X, y = synth_small(700, 1500, seed=59, causal=0.02)
    P = bwgr_amd.Panel(X)
    Q = P.clone()
    _lib.check(_lib.lib().bwgr_debug_withhold(P._h, 1))
    c0 = bwgr_amd.Chain(P, "BayesB", y, it=2, bi=0, pi=0.95, seed=2)
    c1 = bwgr_amd.Chain(Q, "BayesB", y, it=2, bi=0, pi=0.95, seed=3)
    t0 = time.time()
    c0.run_pair(c1, 1)
    for c in (c0, c1):
        with pytest.raises(bwgr_amd.BwgrError) as ei:
            c.sync()
        assert ei.value.code == 4                                   # BWGR_ETIMEOUT
    assert time.time() - t0 < 30.0
    c0.close(); c1.close()
    _lib.check(_lib.lib().bwgr_debug_withhold(P._h, 0))
    solo = []
    for h, seed in ((P, 2), (Q, 3)):
        ch = bwgr_amd.Chain(h, "BayesB", y, it=5, bi=0, pi=0.95, seed=seed); ch.run(5); solo.append(ch.state()); ch.close()
    c0 = bwgr_amd.Chain(P, "BayesB", y, it=5, bi=0, pi=0.95, seed=2)
    c1 = bwgr_amd.Chain(Q, "BayesB", y, it=5, bi=0, pi=0.95, seed=3)
    c0.run_pair(c1, 5)
    s0, s1 = c0.state(), c1.state()
    c0.close(); c1.close(); Q.close(); P.close()
    for got, want in ((s0, solo[0]), (s1, solo[1])):
        assert np.array_equal(got["d"], want["d"]) and np.array_equal(got["b"], want["b"])


def test_a_chain_changes_streamer_height_with_its_company(monkeypatch):
    """k_sweep3 gives a chain that has the GPU to itself 128-row streamers (two to a slab) and the L2 prefetcher workgroup, and 256-row
    streamers as soon as the root panel has a live clone (launch_sweep3).  The slab dots are integer sums, so the chain is the same bit
    for bit whichever geometry a sweep ran on -- alone throughout, beside a clone throughout, or changing in mid-chain with a clone's
    chain running at the same time -- and equal to the oracle's."""
    import bwgr_amd
    from oracle import oracle as O
    monkeypatch.setenv("BWGR_ENG3_THR", "1")
    X, y = synth_small(1100, 2600, seed=61, causal=0.02)
    P = bwgr_amd.Panel(X)
    assert P.pipeline(True)["generation"] == 3
    ch = bwgr_amd.Chain(P, "BayesB", y, it=8, bi=2, pi=0.95, seed=9)
    ch.run(8)
    alone = ch.state(); ch.close()
    ch = bwgr_amd.Chain(P, "BayesB", y, it=8, bi=2, pi=0.95, seed=9)
    ch.run(3)                                   # alone: 128-row streamers (not waited for)
    Q = P.clone()
    other = bwgr_amd.Chain(Q, "BayesC", y, it=8, bi=2, pi=0.95, seed=10)
    other.run(5); ch.run(3)                     # side by side: 256-row streamers
    other.sync(); other.close(); Q.close()
    ch.run(2)                                   # alone again
    mixed = ch.state(); ch.close()
    monkeypatch.setenv("BWGR_SOLO3", "0")
    P2 = bwgr_amd.Panel(X)
    ch = bwgr_amd.Chain(P2, "BayesB", y, it=8, bi=2, pi=0.95, seed=9)
    ch.run(8)
    tall = ch.state(); ch.close(); P2.close(); P.close()
    for got in (mixed, tall):
        assert np.array_equal(got["d"], alone["d"]) and np.array_equal(got["b"], alone["b"]) and np.array_equal(got["e"], alone["e"]) and got["ve"] == alone["ve"]
    o = O.bayes("BayesB", y, X, it=8, bi=2, pi=0.95, seed=9)["last"]
    assert np.array_equal(alone["d"], o["d"]) and scaled_err(alone["b"], o["b"]) < TOL and scaled_err(alone["e"], o["e"]) < TOL
