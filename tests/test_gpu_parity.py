"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerance: north_star asks for 1e-6 relative on effects / variance components with a fixed RNG seed.  Vectors are
compared as max|gpu-oracle| / max|oracle| (TOL below); scalars as plain relative error.  The oracle flavour is "w"
(double accumulators in the Eigen reductions), see oracle/bwgr_oracle.c header.
"""
import numpy as np
import pytest

from conftest import scaled_err, synth_small

pytestmark = pytest.mark.gpu
TOL = 1e-6

ALL_MODELS = ["BayesA", "BayesB", "BayesC", "BayesL", "BayesRR", "BayesCpi", "BayesDpi"]


def _rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-300)


def test_rng_contract_matches_oracle():
    import bwgr_amd
    from oracle import oracle as O
    seed = 0x1234567887654321
    for kind, purpose, nu in [("normal", 0, 0.0), ("normal", 1, 0.0), ("uniform", 2, 0.0), ("chisq", 3, 6.0),
                              ("chisq", 3, 1.5), ("chisq", 17, 201.0), ("chisq", 3, 0.7)]:
        dev = bwgr_amd.debug_variates(seed, kind, 5, 512, it=9, purpose=purpose, nu=nu)
        ref = np.array([O.variate(seed, kind, 5 + i, 9, purpose, nu=nu) for i in range(512)])
        assert np.max(np.abs(dev - ref) / np.maximum(np.abs(ref), 1e-300)) < 1e-12, (kind, purpose, nu)


def test_panel_stats_tpod(tpod):
    import bwgr_amd
    from oracle import oracle as O
    P = bwgr_amd.Panel(tpod["gen"])
    xx, vx, msx = P.stats()
    oxx, ovx, omsx = O.stats(tpod["gen"])
    assert np.array_equal(xx, oxx)             # integer sums of squares: exact
    assert scaled_err(vx, ovx) < 1e-7
    assert _rel(msx, omsx) < 1e-7
    P.close()


@pytest.mark.parametrize("pi", [0.0, 0.3])
@pytest.mark.parametrize("block,nwg", [(0, 0), (16, 1), (64, 1)])
def test_kmup_sweep_tpod(tpod, pi, block, nwg, engine_threshold):
    import bwgr_amd
    from oracle import oracle as O
    X, y = tpod["gen"], tpod["y"]
    n, p = X.shape
    rs = np.random.RandomState(5)
    xx = (X.astype(np.float64) ** 2).sum(0)
    b = rs.normal(size=p) * 0.01
    d = np.ones(p)
    e = (y - y.mean() - X.astype(np.float64) @ b)
    L = np.full(p, 120.0) * rs.uniform(0.5, 2.0, p)
    Ve = 0.03
    g = bwgr_amd.KMUP(X, b, d, xx, e, L, Ve, pi, seed=77, it=3, block=block, nwg=nwg)
    o = O.kmup(X, b, d, xx, e, L, Ve, pi, seed=77, it=3)
    assert scaled_err(g["b"], o["b"]) < TOL
    assert scaled_err(g["e"], o["e"]) < TOL
    assert np.array_equal(g["d"], o["d"])


@pytest.mark.parametrize("model", ALL_MODELS)
def test_short_chain_tpod(tpod, model, engine_threshold):
    import bwgr_amd
    from oracle import oracle as O
    X, y = tpod["gen"], tpod["y"]
    P = bwgr_amd.Panel(X)
    ch = bwgr_amd.Chain(P, model, y, it=20, bi=5, pi=0.9, df=5, R2=0.5, seed=11)
    ch.run(20)
    g = ch.result(); st = ch.state()
    o = O.bayes(model, y, X, it=20, bi=5, pi=0.9, df=5, R2=0.5, seed=11)
    assert scaled_err(g["b"], o["b"]) < TOL
    assert scaled_err(g["hat"], o["hat"]) < TOL
    assert _rel(g["ve"], o["ve"]) < TOL and _rel(g["mu"], o["mu"]) < TOL and _rel(g["h2"], o["h2"]) < 5 * TOL
    assert scaled_err(np.atleast_1d(g["vb"]), np.atleast_1d(o["vb"])) < 5 * TOL
    if "d" in o:
        assert np.array_equal(g["d"], o["d"])
    assert scaled_err(st["e"], o["last"]["e"]) < TOL
    assert scaled_err(st["b"], o["last"]["b"]) < TOL
    assert _rel(st["ve"], o["last"]["ve"]) < TOL
    ch.close(); P.close()


@pytest.mark.parametrize("model,nwg,block", [("BayesA", 2, 32), ("BayesB", 3, 64), ("BayesRR", 4, 128), ("BayesCpi", 2, 48)])
def test_short_chain_multi_workgroup(model, nwg, block, engine_threshold):
    """Row slabs across several workgroups: exercises the in-kernel all-gather of slab partials."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(700, 900, seed=3)
    P = bwgr_amd.Panel(X, nwg=nwg, block=block)
    assert P.nwg == nwg
    ch = bwgr_amd.Chain(P, model, y, it=12, bi=3, pi=0.9, seed=5)
    ch.run(12)
    g = ch.result(); st = ch.state()
    o = O.bayes(model, y, X, it=12, bi=3, pi=0.9, seed=5)
    assert scaled_err(g["b"], o["b"]) < TOL
    assert scaled_err(st["e"], o["last"]["e"]) < TOL
    assert _rel(g["ve"], o["ve"]) < TOL
    ch.close(); P.close()


def test_float_panel_centered():
    """Non-integer X (centred genotypes) goes through the fp32 panel path."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(300, 200, seed=9)
    Xc = (X - X.mean(0)).astype(np.float32)
    g = bwgr_amd.BayesRR(y, Xc, it=10, bi=2, seed=4)
    o = O.bayes("BayesRR", y, Xc, it=10, bi=2, seed=4)
    assert scaled_err(g["b"], o["b"]) < TOL
    assert scaled_err(g["hat"], o["hat"]) < TOL


@pytest.mark.parametrize("model", ["BayesB", "BayesCpi", "BayesA"])
def test_float_panel_chains(model):
    """fp32 panels (block <= 64, fp64 Gram blocks, fp64-FMA slab loops) through the generic streamer / sequencer and the q
    feeders: selection models at lag 3 (sparse cross terms from the fp64 distance-2 blocks), affine at lag 2; several
    blocks with a ragged last one and several slab workgroups."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(500, 330, seed=19)
    Xc = ((X - X.mean(0)) / (X.std(0) + 0.5)).astype(np.float32)
    kw = dict(it=8, bi=2, seed=14)
    if model == "BayesB":
        kw["pi"] = 0.8
    g = getattr(bwgr_amd, model)(y, Xc, block=48, **kw)
    o = O.bayes(model, y, Xc, **kw)
    assert scaled_err(g["b"], o["b"]) < TOL and scaled_err(g["hat"], o["hat"]) < TOL
    if "d" in o:
        assert scaled_err(g["d"], o["d"]) < TOL


# wgr() settings of man/wgr.Rd:82: BRR (defaults), BayesA (iv), BayesB (iv, pi>0), BayesC (pi>0), BayesL (de)
@pytest.mark.parametrize("name,kw", [("BRR", {}), ("BayesA", {"iv": True}), ("BayesB", {"iv": True, "pi": 0.5}),
                                     ("BayesC", {"pi": 0.5}), ("BayesL", {"de": True}), ("thin", {"th": 3, "bi": 4})])
def test_wgr_tpod(tpod, name, kw, engine_threshold):
    import bwgr_amd
    from oracle import oracle as O
    X, y = tpod["gen"], tpod["y"]
    args = dict(it=25, bi=5, th=1, df=5, R2=0.5)
    args.update(kw)
    g = bwgr_amd.wgr(y, X, seed=21, **args)
    o = O.wgr(y, X, seed=21, **args)
    assert list(g.keys()) == ["mu", "b", "Vb", "d", "Ve", "hat", "cxx"]
    assert scaled_err(g["b"], o["b"]) < TOL and scaled_err(g["hat"], o["hat"]) < TOL
    assert scaled_err(np.atleast_1d(g["Vb"]), np.atleast_1d(o["Vb"])) < 5 * TOL
    assert _rel(g["Ve"], o["Ve"]) < TOL and _rel(g["mu"], o["mu"]) < TOL and _rel(g["cxx"], o["cxx"]) < 1e-12
    assert scaled_err(g["d"], o["d"]) < 1e-12


def test_wgr_bag_with_eigk_is_refused(tpod):
    import bwgr_amd
    with pytest.raises(NotImplementedError):
        bwgr_amd.wgr(tpod["y"], tpod["gen"], it=5, bi=1, bag=0.5, eigK={"values": [1.0], "vectors": [[1.0]]})


def test_sharded_entry_points_world1_equals_run(tpod):
    """sweep_blocks over ranges + end_iteration == bwgr_chain_run (the G = 1 path of bwgr_amd/dist.py)."""
    import bwgr_amd
    X, y = tpod["gen"], tpod["y"]
    P = bwgr_amd.Panel(X, block=32)
    a = bwgr_amd.Chain(P, "BayesB", y, it=6, bi=1, pi=0.8, seed=3); a.run(6); sa = a.state()
    b = bwgr_amd.Chain(P, "BayesB", y, it=6, bi=1, pi=0.8, seed=3)
    for _ in range(6):
        for lo in range(0, b.nblocks, 5):
            b.sweep_blocks(lo, min(b.nblocks, lo + 5))
        b.end_iteration(None)
    sb = b.state()
    # Same chain; not always the same bits: k_sweep3 takes a launch's first blocks from slab dots that already hold the previous
    # range's included markers, where the one-launch sweep subtracts their Gram rows in fp64 -- the same numbers to 1e-16
    assert np.array_equal(sa["d"], sb["d"])
    for k in ("b", "e", "vb"):
        assert scaled_err(sa[k], sb[k]) < 1e-9, k
    assert _rel(sa["ve"], sb["ve"]) < 1e-9 and _rel(sa["mu"], sb["mu"]) < 1e-9
    a.close(); b.close(); P.close()


@pytest.mark.parametrize("model,pi", [("BayesB", 0.8), ("BayesCpi", 0.0), ("BayesRR", 0.0)])
def test_ranged_sweeps_are_bit_for_bit_on_the_fp64_engine(tpod, monkeypatch, model, pi):
    """The same comparison on k_sweep2 alone (BWGR_SWEEP=2: no k_sweep3; BWGR_WINV=0: the affine models on k_sweep2's sequencer too):
    its fp64 arithmetic does not depend on where a launch starts, so ranged sweeps and exchange rounds are the one-launch chain BIT FOR BIT."""
    import bwgr_amd
    from bwgr_amd.dist import HipShardEngine
    monkeypatch.setenv("BWGR_SWEEP", "2"); monkeypatch.setenv("BWGR_WINV", "0")
    X, y = tpod["gen"], tpod["y"].astype(np.float32)
    P = bwgr_amd.Panel(X, block=32)
    a = bwgr_amd.Chain(P, model, y, it=6, bi=1, pi=pi, seed=3); a.run(6); sa = a.state(); a.close()
    b = bwgr_amd.Chain(P, model, y, it=6, bi=1, pi=pi, seed=3)
    for _ in range(6):
        for lo in range(0, b.nblocks, 5):
            b.sweep_blocks(lo, min(b.nblocks, lo + 5))
        b.end_iteration(None)
    sb = b.state(); b.close()
    eng = HipShardEngine(P, model, y, 6, 1, pi, 5.0, 0.5, 3, 0, P.p, P.stats()[2])
    for _ in range(6):
        for lo in range(0, eng.nblocks, 7):
            eng.round_apply(eng.round_sweep(lo, min(eng.nblocks, lo + 7)))
        eng.end_iteration(eng.sums())
    sc = eng.chain.state(); eng.chain.close(); P.close()
    for other, what in ((sb, "sweep_blocks"), (sc, "exchange rounds")):
        for k in ("d", "b", "vb"):
            assert np.array_equal(sa[k], other[k]), (what, k)
        assert sa["ve"] == other["ve"] and sa["mu"] == other["mu"], what
    assert np.array_equal(sa["e"], sb["e"])
    assert scaled_err(sc["e"], sa["e"]) < 1e-12   # (round_apply forms e = e0 + (e - e0) in fp64: the last bit may move)


def test_two_shards_on_one_gpu_match_cpu_checker():
    """Two marker shards ('ranks') in one process with a hand-rolled residual exchange, against the CPU checker engine
    (tests/shard_checker.py) doing the same partitioned sampler: global marker ids, MSx_total, p_total, external e."""
    import torch
    import bwgr_amd
    from bwgr_amd.dist import HipShardEngine, shard_bounds
    from oracle import oracle as O
    from shard_checker import OracleRREngine
    X, y = synth_small(300, 160, seed=14, causal=0.2)
    n, p = X.shape
    msx = float(O.stats(X)[2])
    spans = [shard_bounds(p, 2, r, 16) for r in range(2)]
    panels = [bwgr_amd.Panel(np.asfortranarray(X[:, lo:hi]), block=16) for lo, hi in spans]
    gpu = [HipShardEngine(panels[r], "BayesRR", y, 6, 0, 0.0, 5.0, 0.5, 41, spans[r][0], p, msx) for r in range(2)]
    cpu = [OracleRREngine(X[:, lo:hi], y, lo, p, msx, 5.0, 0.5, 41, block=16) for lo, hi in spans]

    def iteration(engs, bps):
        nb = max(e.nblocks for e in engs)
        for r in range(0, nb, bps):
            if hasattr(engs[0], "round_sweep"):   # the product engine: the round's vector arithmetic lives in the library
                ds = [e.round_sweep(r, min(e.nblocks, r + bps)) for e in engs]   # (an empty range sweeps nothing)
                total = sum(d.cpu() for d in ds)
                for e, d in zip(engs, ds):
                    d.copy_(total.to(d.device)); e.round_apply(d)
                continue
            e0 = [e.residual().clone() for e in engs]
            for e in engs:
                if r < e.nblocks:
                    e.sweep_blocks(r, min(e.nblocks, r + bps))
            delta = sum((e.residual() - z).cpu() for e, z in zip(engs, e0))
            for e, z in zip(engs, e0):
                e.set_residual(z + delta.to(z.device))
        s = sum(e.sums().cpu() for e in engs)
        for e in engs:
            e.end_iteration(s)

    for _ in range(6):
        iteration(gpu, 2); iteration(cpu, 2)
    for r in range(2):
        st = gpu[r].chain.state()
        assert scaled_err(st["b"], cpu[r].b) < 2e-5
        assert scaled_err(gpu[r].e[:n].cpu().numpy(), cpu[r].e.numpy()) < 2e-5
        assert _rel(st["ve"], cpu[r].ve) < 2e-5 and _rel(st["mu"], cpu[r].mu) < 2e-5
    for g in gpu:
        g.chain.close()
    for P in panels:
        P.close()


@pytest.mark.parametrize("n,p,block,nwg", [(130, 9, 16, 0), (130, 20, 16, 0), (130, 40, 16, 0), (130, 70, 16, 0), (300, 200, 16, 0), (300, 200, 128, 0), (500, 150, 64, 0),
                                           (700, 260, 128, 0), (700, 100, 48, 2), (1100, 300, 128, 0), (1100, 130, 32, 9)])
@pytest.mark.parametrize("model", ["BayesA", "BayesB"])
def test_geometry_sweep(model, n, p, block, nwg):
    """Slab rows R in {128, 256, 384, 512}, 1..9 workgroups, ragged last block, panels of one, two and three blocks (fewer
    than the pipeline is deep): every launch geometry the panel heuristics can pick must give the oracle's chain."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(n, p, seed=n + p)
    P = bwgr_amd.Panel(X, block=block, nwg=nwg)
    ch = bwgr_amd.Chain(P, model, y, it=5, bi=1, pi=0.8, seed=8)
    ch.run(5)
    st = ch.state()
    o = O.bayes(model, y, X, it=5, bi=1, pi=0.8, seed=8)["last"]
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL and _rel(st["ve"], o["ve"]) < TOL, (P.nwg, P.slab_rows)
    ch.close(); P.close()


@pytest.mark.parametrize("model", ["BayesRR", "BayesB", "BayesCpi", "BayesDpi"])
def test_both_sweep_engines_give_the_same_chain(model, monkeypatch):
    """k_sweep (replicated recurrence behind an all-gather) and k_sweep2 (streamer/sequencer pipeline, the default) are
    two schedules of the same blocked algebra: same chain to round-off of the fp64 partial-sum order."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(900, 700, seed=77)
    out = {}
    for v in ("1", "2"):
        monkeypatch.setenv("BWGR_SWEEP", v)
        P = bwgr_amd.Panel(X, block=64, nwg=4)
        ch = bwgr_amd.Chain(P, model, y, it=8, bi=2, pi=0.9, seed=6)
        ch.run(8)
        out[v] = ch.state()
        ch.close(); P.close()
    o = O.bayes(model, y, X, it=8, bi=2, pi=0.9, seed=6)["last"]
    for v in ("1", "2"):
        assert scaled_err(out[v]["b"], o["b"]) < TOL and scaled_err(out[v]["e"], o["e"]) < TOL
    # (the affine models' default engine solves each block with a precomputed inverse, sweep2w.hip.h: it feeds the un-rounded
    # draws forward, the replicated engine the float-rounded ones -- 1e-7, not 1e-9)
    assert scaled_err(out["1"]["b"], out["2"]["b"]) < (2e-7 if model == "BayesRR" else 1e-9) and np.array_equal(out["1"]["d"], out["2"]["d"])


@pytest.mark.parametrize("model", ["BayesB", "BayesDpi", "BayesC"])
@pytest.mark.parametrize("env", [{"BWGR_SWEEP": "2"}, {"BWGR_SWEEP": "2", "BWGR_LAG": "4"}, {"BWGR_SWEEP": "2", "BWGR_LAG": "2"},
                                 {"BWGR_SWEEP": "2", "BWGR_GRAM16": "0"}, {"BWGR_SWEEP": "2", "BWGR_GRAM16": "0", "BWGR_LAG": "3"},
                                 {}, {"BWGR_D3": "2"}, {"BWGR_D3": "3"}, {"BWGR_D3": "5"}, {"BWGR_R3": "64"}, {"BWGR_R3": "128", "BWGR_D3": "3"},
                                 {"BWGR_GRAM16": "0"}, {"BWGR_GRAM16": "0", "BWGR_D3": "4"}])
def test_selection_pipeline_variants_give_the_same_chain(model, env, monkeypatch):
    """Selection models on int8 panels run the trajectory engine k_sweep3 by default (sweep3.hip.h: streamers on the all-rejected
    trajectory in fixed point, included markers folded in D blocks later, Gram rows on demand; D = 2 .. 12, 64 / 128 / 256 rows per
    streamer, 16- or 32-bit Gram entries), or -- BWGR_SWEEP=2 -- k_sweep2 in one of five schedules: 16-bit Gram staging with the single-barrier sequencer and
    the q feeders (default when every Gram entry fits 16 bits) at lag 3 (default), 4 or 2, or 32-bit staging with the
    generic sequencer at lag 2 (default) or 3.  All are the same blocked algebra; several blocks and a ragged last one so that the distance-1
    and distance-2 cross terms, the first blocks of a launch and the tail are all exercised."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = synth_small(700, 1400, seed=91)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    P = bwgr_amd.Panel(X, block=128, nwg=3)
    ch = bwgr_amd.Chain(P, model, y, it=6, bi=1, pi=0.8, seed=12)
    ch.run(6)
    st = ch.state()
    ch.close(); P.close()
    o = O.bayes(model, y, X, it=6, bi=1, pi=0.8, seed=12)["last"]
    assert scaled_err(st["b"], o["b"]) < TOL and scaled_err(st["e"], o["e"]) < TOL
    assert np.array_equal(st["d"], o["d"])


def _tpod_eigk(tpod):
    Z = tpod["gen"].astype(np.float64); Z = Z - Z.mean(0)
    K = Z @ Z.T; K = K / np.mean(np.diag(K))
    w, v = np.linalg.eigh(K); o = np.argsort(-w)
    return {"values": w[o], "vectors": v[:, o]}


@pytest.mark.parametrize("kw", [{}, {"iv": True, "pi": 0.5}])
def test_wgr_polygenic_term_tpod(tpod, kw):
    """wgr(eigK=eigen(K)): second KMUP sweep over the kernel's eigenvectors each iteration (R/wgr.R:70-78),
    Vk draw (:116-119), list(mu,b,Vb,d,Ve,hat,u,Vk,cxx)."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = tpod["gen"], tpod["y"]
    eig = _tpod_eigk(tpod)
    g = bwgr_amd.wgr(y, X, it=25, bi=5, eigK=eig, VarK=0.9, seed=13, **kw)
    o = O.wgr(y, X, it=25, bi=5, eigK=eig, VarK=0.9, seed=13, **kw)
    assert list(g.keys()) == ["mu", "b", "Vb", "d", "Ve", "hat", "u", "Vk", "cxx"] == list(o.keys())
    assert scaled_err(g["b"], o["b"]) < TOL and scaled_err(g["hat"], o["hat"]) < TOL and scaled_err(g["u"], o["u"]) < 5 * TOL
    assert _rel(g["Ve"], o["Ve"]) < TOL and _rel(g["Vk"], o["Vk"]) < TOL and _rel(g["mu"], o["mu"]) < TOL
    assert scaled_err(g["d"], o["d"]) < 1e-12


@pytest.mark.parametrize("kw", [{"bag": 0.5}, {"bag": 0.8, "rp": True, "iv": True, "pi": 0.5}, {"bag": 0.7, "pi": 0.3}])
def test_wgr_bagging_tpod(tpod, kw):
    """wgr(bag != 1): KMUP2 on sort(sample(n, n*bag, rp)) rows each iteration (R/wgr.R:68,85;
    src/Rcpp20260726ai.cpp:41-77 incl. its '+ b0' numerator and xx*bg + L denominator), df/bag^2, xx*bag."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = tpod["gen"], tpod["y"]
    g = bwgr_amd.wgr(y, X, it=20, bi=5, seed=17, **kw)
    o = O.wgr(y, X, it=20, bi=5, seed=17, **kw)
    assert scaled_err(g["b"], o["b"]) < TOL and scaled_err(g["hat"], o["hat"]) < TOL
    assert _rel(g["Ve"], o["Ve"]) < TOL and _rel(g["mu"], o["mu"]) < TOL and _rel(g["cxx"], o["cxx"]) < 1e-12
    assert scaled_err(np.atleast_1d(g["Vb"]), np.atleast_1d(o["Vb"])) < 5 * TOL and scaled_err(g["d"], o["d"]) < 1e-12


@pytest.mark.parametrize("model", ["BayesA2", "BayesB2", "BayesRR2"])
def test_two_effect_samplers(model):
    """BayesA2 / BayesB2 / BayesRR2 (src/Rcpp20260726ai.cpp:990-1218): two resident panels sharing one residual, swept one
    after the other each iteration; an int8 panel and an fp32 panel, several blocks each."""
    import bwgr_amd
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    n, p1, p2 = 600, 300, 90
    X1 = rng.binomial(2, 0.3, size=(n, p1)).astype(np.float32)
    X2 = rng.normal(size=(n, p2)).astype(np.float32)
    y = (X1[:, :5] @ np.array([1.0, -0.8, 0.6, 0.5, -0.4]) + 0.7 * X2[:, 0] + rng.normal(size=n)).astype(np.float32)
    kw = dict(it=30, bi=5, seed=21)
    if model == "BayesB2":
        kw["pi"] = 0.7
    g = getattr(bwgr_amd, model)(y, X1, X2, block=64, **kw)
    o = O.bayes2(model, y, X1, X2, **kw)
    assert list(g.keys()) == [k for k in o.keys() if k != "last"]
    for k in ("b1", "b2", "hat"):
        assert scaled_err(g[k], o[k]) < TOL, k
    for k in ("mu", "ve", "h2"):
        assert abs(g[k] - o[k]) <= TOL * max(1.0, abs(o[k])), k
    if model == "BayesRR2":
        assert abs(g["vb1"] - o["vb1"]) <= TOL * abs(o["vb1"]) and abs(g["vb2"] - o["vb2"]) <= TOL * abs(o["vb2"])
    else:
        assert scaled_err(g["vb1"], o["vb1"]) < TOL and scaled_err(g["vb2"], o["vb2"]) < TOL
    if model == "BayesB2":
        assert scaled_err(g["d1"], o["d1"]) < TOL and scaled_err(g["d2"], o["d2"]) < TOL


def test_mcmccv_matches_a_checker_built_on_the_oracle(tpod):
    """mcmcCV (R/cv.R:113-216): same folds, same seven fits per fold, same correlations as a checker that runs the CPU
    restatement of every sampler on the fold's training rows."""
    import bwgr_amd
    from oracle import oracle as O
    from bwgr_amd.api import _CV_FITS, _CV_NAMES, _cor_last
    gen, y = tpod["gen"].astype(np.float32), tpod["y"].astype(np.float64)
    kw = dict(k=4, n=2, it=25, bi=5, pi=0.8, seed=77)
    g = bwgr_amd.mcmcCV(y, gen, ReturnGebv=True, **kw)
    N, p = gen.shape
    Ms, Bs = [], []
    for c in range(2):
        w = O.bag_rows(77, c + 1, N, int(round(N / 4)))
        assert np.array_equal(w, bwgr_amd.sample_rows(77, c + 1, N, int(round(N / 4))))
        keep = np.ones(N, bool); keep[w] = False
        B = np.zeros((p, 7))
        for i, model in enumerate(_CV_FITS):
            mk = {"pi": 0.8} if model in ("BayesB", "BayesC") else {}
            B[:, i] = O.bayes(model, y[keep].astype(np.float32), gen[keep], it=25, bi=5, seed=77 + 1000003 * (c + 1) + i, **mk)["b"]
        M = np.empty((len(w), 8)); M[:, :7] = gen[w].astype(np.float64) @ B; M[:, 7] = y[w]
        Ms.append(M); Bs.append(B)
    beta = sum(Bs) / 2
    assert scaled_err(g["beta"], beta) < TOL
    assert scaled_err(g["hat"], gen.astype(np.float64) @ beta + y.mean()) < TOL
    pa = _cor_last(np.vstack(Ms))
    assert list(g["cv"].values()) == sorted(g["cv"].values(), reverse=True)
    for i, nm in enumerate(_CV_NAMES):
        assert abs(g["cv"][nm] - round(float(pa[i]), 4)) <= 1.01e-4


def test_wgr_missing_phenotypes_are_dropped_and_predicted(tpod):
    """R/wgr.R:34-39, 146-152: rows with missing y are left out of the chain and still get a fitted value
    HAT = B0 + gen0 %*% B; missing genotypes are mean-imputed first (R/wgr.R:12-18)."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = tpod["gen"].astype(np.float64).copy(), tpod["y"].astype(np.float64).copy()
    miss = np.array([3, 50, 121, 190])
    y[miss] = np.nan
    X[7, 11] = np.nan; X[100, 200] = np.nan
    g = bwgr_amd.wgr(y, X, it=30, bi=5, seed=9)
    Xi = X.copy(); cm = np.nanmean(Xi, axis=0); idx = np.where(np.isnan(Xi)); Xi[idx] = cm[idx[1]]
    keep = ~np.isnan(y)
    o = O.wgr(y[keep], Xi[keep], it=30, bi=5, seed=9)
    assert scaled_err(g["b"], o["b"]) < TOL and abs(g["mu"] - o["mu"]) <= TOL * max(1.0, abs(o["mu"]))
    assert g["hat"].shape == (y.size,)
    assert scaled_err(g["hat"][keep], o["hat"]) < TOL
    assert scaled_err(g["hat"][~keep], o["mu"] + Xi[~keep] @ o["b"]) < TOL


def test_full_size_properties_c4(monkeypatch):
    """BASELINE config 4 (n = 10 000 x p = 1 000 000 int8, BayesB pi = 0.99; 40 slabs, 7 813 blocks, ragged last block) is
    far too large for the oracle; checked through size-independent properties instead:
    (1) residual identity  e == y - mu - X b  after three iterations, X b formed by an independent fp64 torch product;
    (2) the three sweep engines -- the trajectory engine k_sweep3 (default: fixed-point residual, included markers folded in
        twelve blocks later, Gram rows on demand), the pipelined k_sweep2 (MFMA streamers in fp64, lag 4, feeders) and the
        replicated-recurrence k_sweep (fp64-FMA slab loops) -- give the same chain: inclusion indicators bit-equal, effects to
        1e-9 between the two fp64 engines and to 1e-6 for the fixed-point one;
    (3) the inclusion rate sits near 1 - pi."""
    import torch
    import bwgr_amd
    from bwgr_amd import synth
    n, p = 10000, 1000000
    X = synth.genotypes(n, p)                     # (p, ld) int8 on the GPU, marker-major
    y = synth.scale_phenotype(synth.phenotype(X, n))
    out = {}
    for v in ("3", "2", "1"):
        monkeypatch.setenv("BWGR_SWEEP", v)
        P = bwgr_amd.Panel(X, n=n)
        assert P.pipeline(True)["generation"] == int(v)
        ch = bwgr_amd.Chain(P, "BayesB", y, it=3, bi=0, pi=0.99, seed=synth.SEED)
        ch.run(3)
        out[v] = ch.state()
        ch.close(); P.close()
    for v in ("3", "2"):
        st = out[v]
        b = torch.from_numpy(st["b"]).to(X.device).double()
        xb = torch.zeros(n, dtype=torch.float64, device=X.device)
        step = 50000
        for j0 in range(0, p, step):
            xb += X[j0:j0 + step, :n].double().T @ b[j0:j0 + step]
        e_ref = (y.double() - st["mu"] - xb).cpu().numpy()
        assert np.abs(e_ref - st["e"]).max() < 2e-5 * np.abs(e_ref).max(), v     # st["e"] is the fp64 residual narrowed to float
    assert np.array_equal(out["1"]["d"], out["2"]["d"]) and np.array_equal(out["3"]["d"], out["2"]["d"])
    assert scaled_err(out["1"]["b"], out["2"]["b"]) < 1e-9
    assert scaled_err(out["3"]["b"], out["2"]["b"]) < 1e-6 and scaled_err(out["3"]["e"], out["2"]["e"]) < 1e-6
    assert 0.003 < out["3"]["d"].mean() < 0.05


def test_fit_many_runs_the_same_chains_side_by_side():
    """fit_many: seven samplers on one resident panel, side by side on clones (private scratch + stream).  Every fit must be
    the chain the sampler runs alone -- bit for bit: a chain's arithmetic does not depend on what else is on the chip."""
    import bwgr_amd
    rng = np.random.default_rng(11)
    n, p = 300, 700
    X = rng.integers(0, 3, size=(n, p)).astype(np.int8)
    y = (X[:, :10].astype(np.float64) @ rng.normal(size=10) + rng.normal(size=n)).astype(np.float32)
    models = ["BayesA", "BayesB", "BayesC", "BayesL", "BayesRR", "BayesCpi", "BayesDpi"]
    P = bwgr_amd.Panel(X)
    try:
        assert P.max_concurrent(True) >= 2
        jobs = [dict(model=m, y=y, it=30, bi=10, seed=100 + i) for i, m in enumerate(models)]
        many = bwgr_amd.fit_many(P, jobs)
        few = bwgr_amd.fit_many(P, jobs, concurrent=3, chunk=7)
        # ... and with the selection jobs two to a set of streamer workgroups (bwgr_chain_run_pair); a longer partner, an odd one out
        jobs2 = [dict(model=m, y=y, it=30, bi=10, seed=100 + i) for i, m in enumerate(models)] + [dict(model="BayesB", y=y, it=41, bi=10, pi=0.97, seed=77)]
        pairs = bwgr_amd.fit_many(P, jobs2, pair=True, chunk=5)
        alone77 = bwgr_amd.BayesB(y, P, it=41, bi=10, pi=0.97, seed=77)
        for k in alone77:
            np.testing.assert_array_equal(np.asarray(pairs[-1][k]), np.asarray(alone77[k]), err_msg="paired BayesB %s" % k)
        for i, m in enumerate(models):
            alone = getattr(bwgr_amd, m)(y, P, it=30, bi=10, seed=100 + i)
            for got in (many[i], few[i], pairs[i]):
                assert list(got) == list(alone)
                for k in alone:
                    np.testing.assert_array_equal(np.asarray(got[k]), np.asarray(alone[k]), err_msg="%s %s" % (m, k))
        # a clone cannot outlive its parent silently: the parent's close() closes it
        q = P.clone()
        assert q.n == P.n and q.p == P.p
    finally:
        P.close()


EM_MODELS = ["emRR", "emBA", "emBB", "emBC", "emBCpi", "emDE", "emBL", "emEN", "emML", "lasso"]


def _em_check(model, got, ref, tol=TOL):
    assert list(got) == [k for k in ref if k != "iters"], (model, list(got))
    for k in got:
        g, r = np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64)
        if g.ndim:
            assert scaled_err(g, r) < tol, (model, k, scaled_err(g, r))
        else:
            assert _rel(g, r) < 10 * tol, (model, k, float(g), float(r))


@pytest.mark.parametrize("model", EM_MODELS)
def test_em_family_matches_oracle(model):
    """f4: emRR / emBA / emDE / emML in the reference's shuffled marker order (src/Rcpp20260726ai.cpp:80-128, :250-354,
    :463-521) against the oracle's wide flavour, 1e-6; the full default run (200 sweeps, or up to 300 with the
    convergence test) and a short one."""
    import bwgr_amd
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    n, p = 260, 700
    X = rng.integers(0, 3, size=(n, p)).astype(np.int8)
    if model != "lasso":
        X[:, 17] = 0                                  # a monomorphic marker: xx = 0 (emDE replaces it by 0.1, :261; lasso divides by it)
    y = (X[:, :12].astype(np.float64) @ rng.normal(size=12) + 2.0 * rng.normal(size=n)).astype(np.float32)
    P = bwgr_amd.Panel(X)
    try:
        for maxit in (7, 0):
            got = getattr(bwgr_amd, model)(y, P, maxit=maxit)
            ref = O.em(model, y, X, maxit=maxit)
            _em_check(model, got, ref)
    finally:
        P.close()


def test_em_family_other_shapes():
    """Marker weights for emML, non-default df / R2 / Pi / alpha, a float (centred) panel, a panel smaller than one block, and a
    p above 65535 (libstdc++'s shuffle switches from two swap positions per draw to one)."""
    import bwgr_amd
    from oracle import oracle as O
    rng = np.random.default_rng(4)
    n, p = 150, 400
    X = rng.integers(0, 3, size=(n, p)).astype(np.int8)
    y = (X[:, :6].astype(np.float64) @ rng.normal(size=6) + rng.normal(size=n)).astype(np.float32)
    D = rng.uniform(0.5, 2.0, size=p).astype(np.float32)
    _em_check("emML", bwgr_amd.emML(y, X, D=D, maxit=40), O.em("emML", y, X, D=D, maxit=40))
    _em_check("emRR", bwgr_amd.emRR(y, X, df=4, R2=0.3, maxit=30), O.em("emRR", y, X, df=4, R2=0.3, maxit=30))
    _em_check("emBA", bwgr_amd.emBA(y, X, df=6, R2=0.7, maxit=30), O.em("emBA", y, X, df=6, R2=0.7, maxit=30))
    Xc = (X - X.mean(axis=0)).astype(np.float32)
    _em_check("emDE", bwgr_amd.emDE(y, Xc, R2=0.4, maxit=30, as_int8=False), O.em("emDE", y, Xc, R2=0.4, maxit=30), tol=5e-6)
    _em_check("emRR", bwgr_amd.emRR(y, X[:, :11], maxit=30), O.em("emRR", y, X[:, :11], maxit=30))
    _em_check("emBB", bwgr_amd.emBB(y, X, df=5, R2=0.4, Pi=0.9, maxit=30), O.em("emBB", y, X, df=5, R2=0.4, Pi=0.9, maxit=30))
    _em_check("emBC", bwgr_amd.emBC(y, X, Pi=0.3, maxit=30), O.em("emBC", y, X, Pi=0.3, maxit=30))
    _em_check("emBCpi", bwgr_amd.emBCpi(y, Xc, Pi=0.6, maxit=30, as_int8=False), O.em("emBCpi", y, Xc, Pi=0.6, maxit=30), tol=5e-6)
    _em_check("emBL", bwgr_amd.emBL(y, X, R2=0.3, alpha=0.2, maxit=30), O.em("emBL", y, X, R2=0.3, alpha=0.2, maxit=30))
    _em_check("emEN", bwgr_amd.emEN(y, X, R2=0.6, alpha=0.5, maxit=30), O.em("emEN", y, X, R2=0.6, alpha=0.5, maxit=30))
    n2, p2 = 64, 66000
    X2 = rng.integers(0, 3, size=(n2, p2)).astype(np.int8)
    y2 = (X2[:, :5].astype(np.float64) @ rng.normal(size=5) + rng.normal(size=n2)).astype(np.float32)
    _em_check("emML", bwgr_amd.emML(y2, X2, maxit=3), O.em("emML", y2, X2, maxit=3))


@pytest.mark.parametrize("model", EM_MODELS)
def test_em_family_tpod_defaults(tpod, model):
    """emXX(y, gen) with the reference's defaults on its own example data (data/tpod.RData)."""
    import bwgr_amd
    from oracle import oracle as O
    X, y = tpod["gen"], tpod["y"]
    _em_check(model, getattr(bwgr_amd, model)(y, X), O.em(model, y, X), tol=2e-6 if model in ("emBC", "emBCpi") else TOL)


def test_exchange_rounds_with_one_shard_are_the_plain_chain(tpod):
    """The sharded sampler's device-side round API (bwgr_chain_round_sweep / round_apply, get_sums_dev / end_iteration_dev)
    with a single shard and an identity 'all-reduce' must reproduce bwgr_chain_run (same decisions, effects to 1e-9)."""
    import bwgr_amd
    from bwgr_amd.dist import HipShardEngine
    X, y = tpod["gen"], tpod["y"].astype(np.float32)
    for model, pi in (("BayesB", 0.9), ("BayesRR", 0.0), ("BayesCpi", 0.0)):
        P = bwgr_amd.Panel(X, block=16)
        msx = P.stats()[2]
        eng = HipShardEngine(P, model, y, 5, 1, pi, 5.0, 0.5, 77, 0, P.p, msx)
        for _ in range(5):
            for lo in range(0, eng.nblocks, 7):
                d = eng.round_sweep(lo, min(eng.nblocks, lo + 7))
                eng.round_apply(d)                       # one rank: the all-reduce is the identity
            d = eng.round_sweep(eng.nblocks, eng.nblocks)  # a rank that has run out of blocks still takes part in the round
            eng.round_apply(d)
            s = eng.sums()
            assert s.is_cuda
            eng.end_iteration(s)
        got = eng.chain.state(); eng.chain.close()
        ref_chain = bwgr_amd.Chain(P, model, y, it=5, bi=1, pi=pi, seed=77)
        ref_chain.run(5); ref = ref_chain.state(); ref_chain.close(); P.close()
        np.testing.assert_array_equal(got["d"], ref["d"], err_msg=model)
        for k in ("b", "e", "vb"):   # (to 1e-9, not bit for bit: see test_sharded_entry_points_world1_equals_run)
            assert scaled_err(got[k], ref[k]) < 1e-9, (model, k)
        assert _rel(got["ve"], ref["ve"]) < 1e-9 and _rel(got["mu"], ref["mu"]) < 1e-9
