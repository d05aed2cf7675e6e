import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The selection models' sweeps pick their engine on the device from the chain's inclusion rate (k_sweep3 below 3 % of markers in
# the model, k_sweep2 above).  The parity tests run small dense chains, so they force k_sweep3 everywhere (threshold 1) unless a
# test sets the variable itself; test_engine_choice_follows_the_inclusion_rate covers the default switch.
os.environ.setdefault("BWGR_ENG3_THR", "1")
# some GPU tests import torch after the library is loaded (device tensors as inputs): have the loader import it first
os.environ.setdefault("BWGR_PRELOAD_TORCH", "1")


@pytest.fixture(params=["forced", "shipped"])
def engine_threshold(request, monkeypatch):
    """The core parity matrix runs twice: with k_sweep3 forced for every selection sweep (the suite's default, above) and at the SHIPPED
    device-side engine gate (BWGR_ENG3_THR unset: k_sweep3 below 3 % of the markers in the model, k_sweep2 above).  The variable is read when
    a panel is made, so tests that take this fixture build their panels after it."""
    if request.param == "shipped":
        monkeypatch.delenv("BWGR_ENG3_THR", raising=False)
    else:
        monkeypatch.setenv("BWGR_ENG3_THR", "1")
    return request.param


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_finish(session):
    """Which tests this session selected: per-file counts on the terminal, every id in gpurun_out/selected_tests.txt (merged back from the GPU box)."""
    ids = [it.nodeid for it in session.items]
    per = {}
    for i in ids:
        per[i.split("::")[0]] = per.get(i.split("::")[0], 0) + 1
    print("\nselected %d tests: %s" % (len(ids), ", ".join("%s %d" % (k, v) for k, v in sorted(per.items()))))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "selected_tests.txt"), "w") as f:
            f.write("\n".join(ids) + "\n")
    except OSError:
        pass


@pytest.fixture(scope="session")
def tpod():
    d = np.load(os.path.join(ROOT, "tests", "golden", "tpod.npz"))
    return {"y": d["y"].astype(np.float64), "gen": np.asfortranarray(d["gen"]), "fam": d["fam"], "chr": d["chr"]}


def scaled_err(a, b):
    """max |a-b| / max |b|: the relative error of a vector against the scale of the reference vector.
    (Element-wise relative error is meaningless for effects that pass through zero.)"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    denom = max(float(np.max(np.abs(b))) if b.size else 0.0, 1e-300)
    return float(np.max(np.abs(a - b))) / denom if b.size else 0.0


def synth_small(n, p, seed, h2=0.5, causal=0.05):
    """Small synthetic panel in the spirit of BASELINE.md section 3 (numpy, host)."""
    rs = np.random.RandomState(seed)
    f = rs.uniform(0.05, 0.5, p)
    X = (rs.uniform(size=(n, p)) < f).astype(np.int8) + (rs.uniform(size=(n, p)) < f).astype(np.int8)
    nc = max(1, int(p * causal))
    idx = rs.choice(p, nc, replace=False)
    g = X[:, idx].astype(np.float64) @ rs.normal(size=nc)
    g = (g - g.mean()) / (g.std() + 1e-12)
    y = g * np.sqrt(h2) + rs.normal(size=n) * np.sqrt(1 - h2)
    return np.asfortranarray(X), y
