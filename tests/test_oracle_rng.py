"""CPU: pins the RNG contract of the oracle (oracle/bwgr_rng.h)."""
import numpy as np
from oracle import oracle as O

# Known-answer vectors of Philox4x32-10 published with Random123 (kat_vectors): counter, key -> output
KAT = [
    ([0x00000000] * 4, [0x00000000] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_philox4x32_10_known_answers():
    for ctr, key, out in KAT:
        assert O.philox(ctr, key) == out


def test_variates_are_pure_functions_of_their_counters():
    a = O.variate(7, "normal", 3, 2, 0)
    assert a == O.variate(7, "normal", 3, 2, 0)
    assert a != O.variate(7, "normal", 4, 2, 0) and a != O.variate(7, "normal", 3, 3, 0) and a != O.variate(8, "normal", 3, 2, 0)
    assert a != O.variate(7, "normal", 3, 2, 1)


def test_variate_distributions():
    n = 40000
    z = np.array([O.variate(11, "normal", j, 0, 0) for j in range(n)])
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02 and abs((z ** 3).mean()) < 0.05 and abs((z ** 4).mean() - 3) < 0.15
    u = np.array([O.variate(11, "uniform", j, 0, 2) for j in range(n)])
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.003
    for nu in (0.7, 1.5, 6.0, 201.0):
        c = np.array([O.variate(11, "chisq", j, 0, 3, nu=nu) for j in range(n)])
        assert c.min() > 0
        assert abs(c.mean() - nu) < 0.04 * max(nu, 1) and abs(c.var() - 2 * nu) < 0.12 * 2 * nu


def test_degenerate_mode():
    assert O.variate(1, "normal", 0, 0, 0, mode=1) == 0.0
    assert O.variate(1, "uniform", 0, 0, 2, mode=1) == 0.5
    assert O.variate(1, "chisq", 0, 0, 3, nu=6.0, mode=1) == 6.0
