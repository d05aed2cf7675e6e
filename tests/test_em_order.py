"""The EM family's marker order: the reference re-shuffles it before every sweep with
std::shuffle(order.begin(), order.end(), std::mt19937(i)) (src/Rcpp20260726ai.cpp:103, :277, :331, :491).  std::shuffle's
draw sequence is implementation-defined and the reference pins no toolchain; the oracle restates GNU libstdc++ (GCC 11)
in C.  Here that restatement is pinned against the real thing: the product's bwgr_em_order calls the std::shuffle of the
libstdc++ installed in this image (host-only entry point, no GPU needed), and mt19937's published known answer (the
10000th output of the default-seeded engine is 4123659995, C++ standard [rand.predef]) pins the generator."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.parametrize("p", [1, 2, 3, 7, 8, 376, 1000, 65535, 65536, 65537, 70001])
def test_oracle_shuffle_restatement_matches_the_installed_libstdcxx(p):
    import bwgr_amd
    from oracle import oracle as O
    for upto in (0, 1, 4):
        got = O.em_order(p, upto)
        ref = bwgr_amd.em_order(p, upto)
        assert sorted(got.tolist()) == list(range(p))
        np.testing.assert_array_equal(got, ref)
    if p > 2:
        assert not np.array_equal(O.em_order(p, 0), np.arange(p))


def test_two_at_a_time_and_one_at_a_time_paths_are_both_exercised():
    # libstdc++ draws two swap positions per variate while p*p fits the generator's range (p <= 65535) and one otherwise
    from oracle import oracle as O
    a, b = O.em_order(65535, 0), O.em_order(65536, 0)
    assert a.size == 65535 and b.size == 65536 and not np.array_equal(a, b[:65535])


def test_em_oracle_reaches_the_ridge_fixed_point():
    """emML with a fixed lambda is Gauss-Seidel on (X'X + lambda I) b = X'(y - mu): whatever the marker order, its fixed
    point is the ridge solution.  Run the oracle to convergence and compare with numpy at the lambda it ends with."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    n, p = 120, 40
    X = rng.integers(0, 3, size=(n, p)).astype(np.float64)
    y = X[:, :5] @ rng.normal(size=5) + rng.normal(size=n)
    fit = O.em("emML", y, X, maxit=300)
    lam = fit["Ve"] / fit["Vb"]
    Xc = X
    # at the fixed point e has zero mean and b solves the ridge system on the residual's normal equations
    e = y - fit["mu"] - Xc @ fit["b"].astype(np.float64)
    g = Xc.T @ e - lam * fit["b"]
    assert abs(e.mean()) < 1e-4
    assert np.max(np.abs(g)) / np.max(np.abs(Xc.T @ (y - y.mean()))) < 2e-3
    np.testing.assert_allclose(fit["hat"], y - e, atol=2e-4)
