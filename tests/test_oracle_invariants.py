"""CPU: analytic invariants that pin the oracle to the mathematics of the reference's samplers.
The reference ships no golden vectors, so these replace them."""
import numpy as np
import pytest
from oracle import oracle as O
from conftest import scaled_err, synth_small

ALL_MODELS = ["BayesA", "BayesB", "BayesC", "BayesL", "BayesRR", "BayesCpi", "BayesDpi"]


def test_tpod_fixture_facts(tpod):
    """var(y)=0.03743, MSx=351.944, mean xx=379.128, genotype counts."""
    y, X = tpod["y"], tpod["gen"]
    assert X.shape == (196, 376) and y.shape == (196,)
    assert abs(y.var(ddof=1) - 0.03743) < 5e-6
    xx, vx, msx = O.stats(X)
    assert abs(msx - 351.944) < 2e-3 and abs(xx.mean() - 379.128) < 1e-3
    assert list(np.bincount(X.ravel())) == [34470, 4784, 34442]
    assert np.array_equal(xx, (X.astype(np.float64) ** 2).sum(0).astype(np.float32))
    assert scaled_err(vx, X.astype(np.float64).var(axis=0, ddof=1)) < 1e-6


@pytest.mark.parametrize("flavour", ["w", "f"])
def test_kmup_residual_identity(tpod, flavour):
    """After any sweep e == e_in - X (b_out - b_in): the residual is maintained, never recomputed."""
    X, y = tpod["gen"], tpod["y"]
    Xd = X.astype(np.float64)
    rs = np.random.RandomState(1)
    p = X.shape[1]
    b = rs.normal(size=p) * 0.01
    e = y - y.mean() - Xd @ b
    xx = (Xd ** 2).sum(0)
    for pi in (0.0, 0.4):
        o = O.kmup(X, b, np.ones(p), xx, e, np.full(p, 100.0), 0.03, pi, seed=3, flavour=flavour)
        e_expect = e.astype(np.float32).astype(np.float64) - Xd @ (o["b"].astype(np.float64) - b.astype(np.float32))
        assert scaled_err(o["e"], e_expect) < (1e-6 if flavour == "w" else 2e-5)
        assert set(np.unique(o["d"])) <= {0.0, 1.0}
        if pi == 0:
            assert np.all(o["d"] == 1)


def test_kmup_degenerate_is_gauss_seidel(tpod):
    """With z = 0 the sweep is one Gauss-Seidel pass on (X'X + diag L) b = X'(y - mu); iterated, it converges to the
    ridge solution (numpy.linalg.solve)."""
    X, y = tpod["gen"][:, :30].astype(np.float32), tpod["y"]
    X = np.asfortranarray(X - X.mean(0))      # centred columns: a well-conditioned system, fast Gauss-Seidel
    Xd = X.astype(np.float64)
    p = X.shape[1]
    lam = np.full(p, 60.0)
    b = np.zeros(p); e = y - y.mean()
    xx = (Xd ** 2).sum(0)
    for it in range(400):
        o = O.kmup(X, b, np.ones(p), xx, e, lam, 0.03, 0.0, rng_mode=1)
        b, e = o["b"].astype(np.float64), o["e"].astype(np.float64)
    ridge = np.linalg.solve(Xd.T @ Xd + np.diag(lam), Xd.T @ (y - y.mean()))
    assert scaled_err(b, ridge) < 1e-5


def test_kmup_literal_and_stable_inclusion_agree_where_literal_is_finite(tpod):
    """cj/(cj+dj) (src/Rcpp20260726ai.cpp:25-27) == 1/(1+pi/(1-pi) exp(C(|e2|^2-|e1|^2))) on tpod, where the
    literal form does not underflow; on a large-|e| input the literal form is NaN and rejects every marker."""
    X, y = tpod["gen"], tpod["y"]
    p = X.shape[1]
    xx = (X.astype(np.float64) ** 2).sum(0)
    e = y - y.mean()
    args = (X, np.zeros(p), np.ones(p), xx, e, np.full(p, 100.0), 0.03, 0.5)
    a = O.kmup(*args, seed=9, stable=1, flavour="f")
    b = O.kmup(*args, seed=9, stable=0, flavour="f")
    assert np.array_equal(a["d"], b["d"]) and scaled_err(a["b"], b["b"]) < 1e-6
    big = O.kmup(X, np.zeros(p), np.ones(p), xx, e * 60, np.full(p, 100.0), 0.03, 0.5, seed=9, stable=0, flavour="f")
    assert np.all(big["d"] == 0)          # 0/0 = NaN -> "rbinom(1,NaN)==1" is false for every marker
    ok = O.kmup(X, np.zeros(p), np.ones(p), xx, e * 60, np.full(p, 100.0), 0.03, 0.5, seed=9, stable=1, flavour="f")
    assert ok["d"].sum() > 0


@pytest.mark.parametrize("model", ALL_MODELS)
def test_chain_residual_identity_and_flavours(tpod, model):
    """(1) e == y - mu - X b at the end of a chain; (2) the wide flavour (double residual/accumulators, the GPU's
    parity target) and the float-faithful flavour are the same algorithm: they differ by float round-off only."""
    X, y = tpod["gen"], tpod["y"]
    w = O.bayes(model, y, X, it=20, bi=5, pi=0.9, seed=11)
    f = O.bayes(model, y, X, it=20, bi=5, pi=0.9, seed=11, flavour="f")
    last = w["last"]
    e_expect = y.astype(np.float32).astype(np.float64) - last["mu"] - X.astype(np.float64) @ last["b"].astype(np.float64)
    assert scaled_err(last["e"], e_expect) < 5e-6      # mu is accumulated in float
    assert scaled_err(f["b"], w["b"]) < 2e-5 and scaled_err(f["last"]["e"], last["e"]) < 5e-5
    assert abs(f["ve"] - w["ve"]) / w["ve"] < 2e-5
    if "d" in w:
        assert np.array_equal(w["d"], f["d"])


def test_bayesrr_posterior_mean_matches_ridge():
    """Posterior mean of BayesRR effects vs the closed-form ridge solution at the posterior-mean lambda, within
    Monte-Carlo error."""
    X, y = synth_small(250, 40, seed=2, causal=0.3)
    r = O.bayes("BayesRR", y, X, it=3000, bi=500, seed=5)
    Xd = X.astype(np.float64)
    lam = r["ve"] / r["vb"]
    yc = y - r["mu"]
    ridge = np.linalg.solve(Xd.T @ Xd + lam * np.eye(X.shape[1]), Xd.T @ yc)
    assert np.corrcoef(r["b"], ridge)[0, 1] > 0.995
    assert scaled_err(r["b"], ridge) < 0.1
    assert np.corrcoef(r["hat"], y)[0, 1] > 0.5


def test_return_lists_match_reference_names_and_order(tpod):
    X, y = tpod["gen"][:, :40], tpod["y"]
    names = {
        "BayesA": ["mu", "b", "hat", "vb", "ve", "h2", "MSx"], "BayesL": ["mu", "b", "hat", "vb", "ve", "h2", "MSx"],
        "BayesRR": ["mu", "b", "hat", "vb", "ve", "h2", "MSx"],
        "BayesB": ["mu", "b", "d", "hat", "vb", "ve", "h2", "MSx"], "BayesC": ["mu", "b", "d", "hat", "vb", "ve", "h2", "MSx"],
        "BayesCpi": ["mu", "b", "d", "pi", "hat", "h2", "vb", "ve", "PVAL"], "BayesDpi": ["mu", "b", "d", "pi", "hat", "h2", "vb", "ve", "PVAL"],
    }
    for m, nm in names.items():
        r = O.bayes(m, y, X, it=6, bi=2, seed=1)
        assert [k for k in r.keys() if k != "last"] == nm
        assert np.ndim(r["vb"]) == (1 if m in O.PER_MARKER_VB else 0)
    w = O.wgr(y, X, it=6, bi=2, seed=1)
    assert list(w.keys()) == ["mu", "b", "Vb", "d", "Ve", "hat", "cxx"] and np.ndim(w["Vb"]) == 0
    assert np.ndim(O.wgr(y, X, it=6, bi=2, iv=True, seed=1)["Vb"]) == 1


def test_wgr_oracle_fits_tpod(tpod):
    """man/bWGR.Rd:27-31 `Fit = wgr(y,gen); cor(y,Fit$hat)`: the fit must be sane (no expected value is published)."""
    X, y = tpod["gen"], tpod["y"]
    r = O.wgr(y, X, it=300, bi=100, seed=2)
    assert np.corrcoef(y, r["hat"])[0, 1] > 0.5
    assert abs(r["cxx"] - 379.128) < 1e-3 and np.all(r["d"] == 1)
    rb = O.wgr(y, X, it=300, bi=100, iv=True, pi=0.5, seed=2)
    assert 0.2 < rb["d"].mean() < 0.9 and np.corrcoef(y, rb["hat"])[0, 1] > 0.5


@pytest.mark.parametrize("model", ["BayesA2", "BayesB2", "BayesRR2"])
def test_two_effect_residual_identity_and_flavours(tpod, model):
    """e == y - mu - X1 b1 - X2 b2 after the last iteration (src/Rcpp20260726ai.cpp:1025-1043), and the float-faithful
    and wide restatements agree to float round-off."""
    X, y = tpod["gen"].astype(np.float32), tpod["y"].astype(np.float32)
    X1, X2 = X[:, :250], X[:, 250:]
    kw = dict(it=12, bi=3, seed=4)
    if model == "BayesB2":
        kw["pi"] = 0.8
    w = O.bayes2(model, y, X1, X2, **kw)
    f = O.bayes2(model, y, X1, X2, flavour="f", **kw)
    L = w["last"]
    e = y.astype(np.float64) - L["mu"] - X1.astype(np.float64) @ L["b1"] - X2.astype(np.float64) @ L["b2"]
    assert np.abs(e - L["e"]).max() < 2e-4 * np.abs(y).max()
    assert np.abs(w["b1"] - f["b1"]).max() < 1e-3 * np.abs(w["b1"]).max() + 1e-6
    assert 0.0 < w["h2"] < 1.0 and w["ve"] > 0


@pytest.mark.parametrize("model", ["emRR", "emBA", "emBB", "emBC", "emBCpi", "emDE", "emBL", "emEN", "emML", "lasso"])
def test_em_members_on_tpod_two_flavours(tpod, model):
    """The EM family on the reference's own example data (man/emRR.Rd style calls: emXX(y, gen) with defaults): the wide
    and the float-faithful restatement describe the same fit (they differ only in where float rounding happens), the fit
    explains the phenotype, and the residual identity e = y - hat holds for the members that return y - e."""
    from oracle import oracle as O
    X, y = tpod["gen"], tpod["y"]
    w = O.em(model, y, X, flavour="w")
    f = O.em(model, y, X, flavour="f")
    assert w["iters"] == f["iters"] or model in ("emDE", "emML", "emEN", "lasso")
    assert np.all(np.isfinite(w["b"])) and np.all(np.isfinite(f["b"]))
    if np.std(w["b"]) > 0:
        assert np.corrcoef(w["b"], f["b"])[0, 1] > 0.999
    assert np.corrcoef(w["hat"], f["hat"])[0, 1] > 0.9999
    assert np.corrcoef(w["hat"], y)[0, 1] > 0.5
    assert 0.0 < w["h2"] < 1.0
    if "d" in w:
        assert np.all(w["d"] >= 0) and np.all(w["d"] <= 1)
