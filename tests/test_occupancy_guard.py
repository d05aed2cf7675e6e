"""The occupancy guard (include/bwgr.h, "Occupancy guard"): a sweep's workgroups wait for one another, so a launch that cannot be
resident beside the sweeps in flight is refused with BWGR_EINVAL before anything is enqueued, instead of spinning to BWGR_ETIMEOUT.
(No reference counterpart: the reference's sweep is one CPU loop, src/Rcpp20260726ai.cpp:681-699.)"""
import ctypes as C
import numpy as np
import pytest

from bwgr_amd import _lib


def _fits(grid, per_cu, cus, busy):
    need = C.c_int(-1)
    rc = _lib.lib().bwgr_debug_occupancy_fits(grid, per_cu, cus, busy, C.byref(need))
    return rc, need.value


def test_guard_arithmetic():
    OK = 0
    # 43 workgroups, one per unit: five such sweeps fill 215 of 256 units, a sixth does not fit
    assert _fits(43, 1, 256, 0) == (OK, 43)
    assert _fits(43, 1, 256, 4 * 43) == (OK, 43)
    rc, need = _fits(43, 1, 256, 5 * 43)
    assert rc != OK and need == 43
    # two workgroups per unit halve the units (rounded up)
    assert _fits(81, 2, 256, 0) == (OK, 41)
    assert _fits(512, 2, 256, 0) == (OK, 256)
    assert _fits(513, 2, 256, 0)[0] != OK
    # exactly full is allowed, one more is not
    assert _fits(40, 1, 256, 216)[0] == OK and _fits(41, 1, 256, 216)[0] != OK
    # a kernel that does not fit a unit at all, and nonsense arguments
    assert _fits(10, 0, 256, 0)[0] != OK
    assert _fits(0, 1, 256, 0)[0] != OK and _fits(10, 1, 0, 0)[0] != OK and _fits(10, 1, 256, -1)[0] != OK


def test_dma_streamer_is_chosen_only_below_4gib_of_columns():
    """k_sweep3's LDS-DMA streamers address a launch's columns by 32-bit lane offsets (sweep3.hip.h, tile_issue): the host selects them only
    while columns x slab rows < 2^32 and falls back to the register-path streamers (64-bit offsets) beyond -- a WGS-size panel such as
    2 000 x 20 M fits the GPU and must not wrap."""
    f = _lib.lib().bwgr_debug_stream3_dma
    assert f(1_000_000, 256) == 1                 # C4: 2.56e8
    assert f(16_777_215, 256) == 1 and f(16_777_216, 256) == 0      # the edge at R = 256
    assert f(20_000_000, 256) == 0                # the WGS-size panel
    assert f(20_000_000, 128) == 1 and f(33_554_432, 128) == 0
    assert f(0, 256) == 1 and f(-1, 256) == 0 and f(10, 0) == 0


@pytest.mark.gpu
def test_oversubscription_is_refused_and_the_chain_survives():
    """cap = bwgr_panel_max_concurrent chains in flight on clones of one panel; one more is refused (BWGR_EINVAL, nothing enqueued), and
    after the others have finished the refused chain runs and is bit for bit the chain run alone."""
    from bwgr_amd import synth
    from bwgr_amd.api import Panel, Chain
    n, p = 10000, 120000
    X = synth.genotypes(n, p, device=0)
    y = synth.scale_phenotype(synth.phenotype(X, n))        # (n float32 on the device)
    P = Panel(X, n=n, device=0)
    cap = P.max_concurrent(True)
    assert 1 <= cap <= 64
    handles = [P] + [P.clone() for _ in range(cap)]          # cap + 1 handles
    iters = 48
    chains = [Chain(h, "BayesB", y, it=iters, bi=0, pi=0.99, seed=100 + i) for i, h in enumerate(handles)]
    try:
        for ch in chains[:cap]:
            ch.run(iters)                                      # asynchronous: cap chains' sweeps are now enqueued or running
        with pytest.raises(_lib.BwgrError) as ei:
            chains[cap].run(iters)
        assert ei.value.code == 1 and "occupancy guard" in str(ei.value), str(ei.value)
        done = C.c_int(-1)
        _lib.lib().bwgr_chain_iterations(chains[cap]._h, C.byref(done))
        assert done.value == 0                                 # refused before anything was enqueued
        for ch in chains[:cap]:
            ch.sync()
        chains[cap].run(iters)                                 # the chip is free again
        late = chains[cap].result()
    finally:
        for ch in chains:
            ch.close()
        for h in handles[1:]:
            h.close()
    alone = Chain(P, "BayesB", y, it=iters, bi=0, pi=0.99, seed=100 + cap)
    try:
        alone.run(iters)
        ref = alone.result()
    finally:
        alone.close()
        P.close()
    assert np.array_equal(late["d"], ref["d"])
    assert np.max(np.abs(late["b"] - ref["b"])) <= 1e-9 * max(1e-30, np.max(np.abs(ref["b"])))
