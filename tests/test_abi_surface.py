"""CPU: the C-ABI library builds, loads, exports every symbol include/bwgr.h declares, and refuses to compute
without a GPU (no CPU fallback)."""
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "bwgr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bwgr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from bwgr_amd import build as B, _lib
    B.build()
    L = _lib.lib()
    decl = _declared()
    assert len(decl) >= 20
    missing = [s for s in decl if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == decl
    L.bwgr_abi_version.restype = int
    assert L.bwgr_abi_version() == 1


def test_no_cpu_fallback():
    import bwgr_amd
    if bwgr_amd.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(bwgr_amd.BwgrError) as ei:
        bwgr_amd.BayesRR(np.zeros(8), np.zeros((8, 4), np.int8), it=2, bi=0)
    assert ei.value.code == 5   # BWGR_ENODEV
    with pytest.raises(bwgr_amd.BwgrError):
        bwgr_amd.KMUP(np.zeros((8, 4), np.int8), np.zeros(4), np.ones(4), np.ones(4), np.zeros(8), np.ones(4), 1.0, 0.0)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bwgr_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                for pat in (r"^\s*(import|from)\s+oracle", r"#include\s+[<\"].*oracle", r"liboracle", r"oracle[/\\.]"):
                    assert not re.search(pat, txt, flags=re.M), (os.path.join(dirpath, f), pat)


def test_host_mirror_signatures_match_reference():
    """Same names, argument order and defaults as R/RcppExports.R:4-6,48-74 and R/wgr.R:2-8."""
    import inspect
    import bwgr_amd as B

    def pos(fn):
        return [(p.name, p.default) for p in inspect.signature(fn).parameters.values()
                if p.kind == inspect.Parameter.POSITIONAL_OR_KEYWORD]
    E = inspect.Parameter.empty
    assert pos(B.KMUP) == [("X", E), ("b", E), ("d", E), ("xx", E), ("e", E), ("L", E), ("Ve", E), ("pi", E)]
    assert pos(B.KMUP2) == [("X", E), ("Use", E), ("b", E), ("d", E), ("xx", E), ("E", E), ("L", E), ("Ve", E), ("pi", E)]   # R/RcppExports.R:8
    for f in (B.BayesA, B.BayesL, B.BayesRR, B.BayesCpi, B.BayesDpi):
        assert pos(f) == [("y", E), ("X", E), ("it", 1500), ("bi", 500), ("df", 5), ("R2", 0.5)]
    for f in (B.BayesB, B.BayesC):
        assert pos(f) == [("y", E), ("X", E), ("it", 1500), ("bi", 500), ("pi", 0.95), ("df", 5), ("R2", 0.5)]
    assert pos(B.wgr) == [("y", E), ("X", E), ("it", 1500), ("bi", 500), ("th", 1), ("bag", 1), ("rp", False), ("iv", False),
                          ("de", False), ("pi", 0), ("df", 5), ("R2", 0.5), ("eigK", None), ("VarK", 0.95), ("verb", False)]
    # two-effect samplers and the EM / Gauss-Seidel family, R/RcppExports.R:12-46, 76-86 (names, order, defaults)
    for f in (B.BayesA2, B.BayesRR2):
        assert pos(f) == [("y", E), ("X1", E), ("X2", E), ("it", 1500), ("bi", 500), ("df", 5), ("R2", 0.5)]
    assert pos(B.BayesB2) == [("y", E), ("X1", E), ("X2", E), ("it", 1500), ("bi", 500), ("pi", 0.95), ("df", 5), ("R2", 0.5)]
    for f in (B.emRR, B.emBA):
        assert pos(f) == [("y", E), ("gen", E), ("df", 10), ("R2", 0.5)]
    for f in (B.emBB, B.emBC, B.emBCpi):
        assert pos(f) == [("y", E), ("gen", E), ("df", 10), ("R2", 0.5), ("Pi", 0.75)]
    assert pos(B.emDE) == [("y", E), ("gen", E), ("R2", 0.5)]
    for f in (B.emBL, B.emEN):
        assert pos(f) == [("y", E), ("gen", E), ("R2", 0.5), ("alpha", 0.02)]
    assert pos(B.emML) == [("y", E), ("gen", E), ("D", None)]
    assert pos(B.lasso) == [("y", E), ("gen", E)]
