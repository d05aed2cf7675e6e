"""The oracle's R-stream back-end (oracle/bwgr_rstream.h, rng_mode = RSTREAM): an UNVERIFIED restatement of R's default generators, there so that
someone with R can pin the oracle to real bWGR (tools/make_r_fixtures.R, tests/test_r_fixtures.py).  What CAN be checked without R:

* known answers of R's stream that every R user has seen printed -- set.seed(1); runif(10), rnorm(3), rexp(3), rbinom(10, 1, .5); set.seed(42);
  set.seed(123) -- quoted from R's own output as it appears throughout R's documentation and tutorials (7 significant digits): they pin the seed
  scrambling, the Mersenne-Twister, the fix-up into (0, 1), the inversion normal (AS 241) and exp_rand;
* the distributions of rgamma (both branches) and rchisq by their moments;
* that the samplers run in this mode, draw in the reference's order (BayesB's extra normal only on rejection, src/Rcpp20260726ai.cpp:678) and agree
  statistically with the Philox runs;
* that adding the mode left the Philox path untouched (tests/test_oracle_goldens.py: bit for bit)."""
import numpy as np

from oracle import oracle as O


def _draws(seed, kind, k, par=0.0):
    O.rstream_seed(seed)
    return np.array([O.rstream_draw(kind, par) for _ in range(k)])


def test_known_answers_of_r():
    assert np.allclose(_draws(1, "unif", 10), [0.2655087, 0.3721239, 0.5728534, 0.9082078, 0.2016819, 0.8983897, 0.9446753, 0.6607978, 0.6291140, 0.0617863], atol=5e-8)
    assert np.allclose(_draws(1, "norm", 3), [-0.6264538, 0.1836433, -0.8356286], atol=5e-8)
    assert np.allclose(_draws(42, "unif", 2), [0.9148060, 0.9370754], atol=5e-8)
    assert np.allclose(_draws(42, "norm", 1), [1.37095845], atol=5e-9)
    assert np.allclose(_draws(123, "unif", 3), [0.2875775, 0.7883051, 0.4089769], atol=5e-8)
    assert np.allclose(_draws(123, "norm", 3), [-0.56047565, -0.23017749, 1.55870831], atol=5e-9)
    assert np.allclose(_draws(1, "exp", 3), [0.7551818, 1.1816428, 0.1457067], atol=5e-8)
    assert list(_draws(1, "binom1", 10, 0.5).astype(int)) == [0, 0, 1, 1, 0, 1, 1, 1, 1, 0]


def test_gamma_and_chisq_moments():
    for a in (0.3, 0.5, 1.0, 2.5, 3.6, 13.0, 100.0):     # GS below 1; GD's three parameter ranges (3.686, 13.022) above
        x = _draws(7, "gamma", 200000, a)
        assert abs(x.mean() - a) < 5 * np.sqrt(a / 200000) and abs(x.var() - a) < 0.03 * a + 0.01, a
    x = _draws(9, "chisq", 200000, 6.0)
    assert abs(x.mean() - 6.0) < 0.05 and abs(x.var() - 12.0) < 0.4
    p = _draws(3, "binom1", 100000, 0.83).mean()
    assert abs(p - 0.83) < 0.006
    assert _draws(3, "binom1", 5, float("nan")).sum() == 0     # rbinom(1, NaN) == 1 is FALSE (the reference's degenerate KMUP branch)


def test_the_two_flavours_keep_separate_streams():
    O.rstream_seed(5, "w"); O.rstream_seed(6, "f")
    a = O.rstream_draw("unif", flavour="w"); b = O.rstream_draw("unif", flavour="f")
    O.rstream_seed(5, "f")
    assert O.rstream_draw("unif", flavour="f") == a and a != b


def test_samplers_run_on_the_r_stream(tpod):
    y, X = tpod["y"], tpod["gen"]
    for model in ("BayesB", "BayesRR", "BayesCpi"):
        O.rstream_seed(2024, "f")
        r1 = O.bayes(model, y, X, it=150, bi=50, pi=0.9, seed=0, rng_mode=O.RSTREAM, flavour="f")
        O.rstream_seed(2024, "f")
        r2 = O.bayes(model, y, X, it=150, bi=50, pi=0.9, seed=999, rng_mode=O.RSTREAM, flavour="f")   # (the seed argument is not used in this mode)
        assert np.array_equal(r1["b"], r2["b"]) and r1["ve"] == r2["ve"]
        ph = O.bayes(model, y, X, it=150, bi=50, pi=0.9, seed=5, flavour="f")
        assert abs(r1["ve"] - ph["ve"]) < 0.15 * ph["ve"] and abs(np.corrcoef(y, r1["hat"])[0, 1] - np.corrcoef(y, ph["hat"])[0, 1]) < 0.06
    O.rstream_seed(7, "w")
    w = O.wgr(y, X, it=60, bi=10, seed=0, rng_mode=O.RSTREAM)
    assert np.isfinite(w["Ve"]) and np.corrcoef(y, w["hat"])[0, 1] > 0.6
