"""ctypes front-end of the CPU oracle (oracle/bwgr_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under bwgr_amd/ does.  `flavour` selects the accumulator width the C file was compiled
with: "w" (wide, the GPU's parity target) or "f" (float-faithful, the CPU baseline).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = {"BayesA": 0, "BayesB": 1, "BayesC": 2, "BayesL": 3, "BayesRR": 4, "BayesCpi": 5, "BayesDpi": 6}
PER_MARKER_VB = {"BayesA", "BayesB", "BayesL", "BayesDpi"}
PURPOSE = {"Z1": 0, "Z2": 1, "U": 2, "CHI": 3, "G_MU": 16, "G_VE": 17, "G_VB": 18, "G_VK": 19}
GLOBAL_MARKER = 0xFFFFFFFF

_libs = {}


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("bwgr_oracle.c", "bwgr_rng.h", "bwgr_rstream.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return so


def lib(fast=False):
    key = "fast" if fast else "std"
    if key not in _libs:
        build()
        _libs[key] = C.CDLL(os.path.join(_HERE, "liboracle_fast.so" if fast else "liboracle.so"))
    return _libs[key]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def as_f32_colmajor(X):
    """X (n x p, any dtype) -> float32 column-major copy, the reference's Eigen::MatrixXf."""
    return np.asfortranarray(np.asarray(X), dtype=np.float32)


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().oracle_philox_w((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return [int(v) for v in out]


def variate(seed, kind, marker, it, purpose, k=0, nu=0.0, mode=0):
    f = lib().oracle_variate_w
    f.restype = C.c_double
    kinds = {"normal": 0, "uniform": 1, "chisq": 2}
    return f(C.c_uint64(seed), C.c_int(mode), C.c_int(kinds[kind]), C.c_double(nu), C.c_uint32(marker),
             C.c_uint32(it), C.c_uint32(purpose), C.c_uint32(k))


RSTREAM = 2   # rng_mode: R's own serial stream (oracle/bwgr_rstream.h -- an UNVERIFIED restatement; seed it with rstream_seed = set.seed)


def rstream_seed(seed, flavour="w", fast=False):
    """set.seed(seed) of the R-stream back-end (rng_mode=RSTREAM); each flavour of the C file keeps its own stream."""
    getattr(lib(fast), "oracle_rstream_seed_" + flavour)(C.c_uint32(int(seed) & 0xFFFFFFFF))


def rstream_draw(kind, par=0.0, flavour="w"):
    """One draw from the R-stream back-end: kind in unif, norm, exp, gamma (shape par), chisq (df par), binom1 (probability par)."""
    f = getattr(lib(), "oracle_rstream_draw_" + flavour); f.restype = C.c_double
    return f(C.c_int({"unif": 0, "norm": 1, "exp": 2, "gamma": 3, "chisq": 4, "binom1": 5}[kind]), C.c_double(par))


def stats(X, flavour="w"):
    Xf = as_f32_colmajor(X)
    n, p = Xf.shape
    xx = np.empty(p, np.float32); vx = np.empty(p, np.float32); msx = C.c_float()
    getattr(lib(), "oracle_stats_" + flavour)(_fp(Xf), C.c_int64(n), C.c_int64(p), C.c_int64(n), _fp(xx), _fp(vx), C.byref(msx))
    return xx, vx, float(msx.value)


def fvar(x, flavour="w"):
    x = np.ascontiguousarray(x, np.float32)
    f = getattr(lib(), "oracle_fvar_" + flavour); f.restype = C.c_float
    return float(f(_fp(x), C.c_int64(x.size)))


def kmup(X, b, d, xx, e, L, Ve, pi, seed=1, it=0, rng_mode=0, stable=1, flavour="w", fast=False, marker0=0):
    """Reference KMUP(X,b,d,xx,e,L,Ve,pi) -> dict(b,d,e), src/Rcpp20260726ai.cpp:12-38."""
    Xf = as_f32_colmajor(X)
    n, p = Xf.shape
    b = np.array(b, np.float32); d = np.array(d, np.float32); e = np.array(e, np.float32)
    xx = np.ascontiguousarray(xx, np.float32); L = np.ascontiguousarray(L, np.float32)
    rc = getattr(lib(fast), "oracle_kmup_" + flavour)(
        _fp(Xf), C.c_int64(n), C.c_int64(p), C.c_int64(n), _fp(b), _fp(d), _fp(xx), _fp(e), _fp(L),
        C.c_float(Ve), C.c_float(pi), C.c_uint64(seed), C.c_uint32(it), C.c_int(rng_mode), C.c_int(stable), C.c_uint32(marker0))
    assert rc == 0
    return {"b": b, "d": d, "e": e}


def bayes(model, y, X, it=1500, bi=500, pi=0.95, df=5.0, R2=0.5, seed=1, rng_mode=0, flavour="w", fast=False):
    """Reference BayesA/B/C/L/RR/Cpi/Dpi(y,X,it,bi,[pi,]df,R2), src/Rcpp20260726ai.cpp:589-987.
    Returns the reference's list as a dict plus 'last' (chain state after the final iteration)."""
    Xf = X if (isinstance(X, np.ndarray) and X.dtype == np.float32 and X.flags.f_contiguous) else as_f32_colmajor(X)
    n, p = Xf.shape
    y = np.ascontiguousarray(y, np.float32)
    per = model in PER_MARKER_VB
    B = np.zeros(p, np.float32); D = np.zeros(p, np.float32); hat = np.zeros(n, np.float32)
    VB = np.zeros(p if per else 1, np.float32); PVAL = np.zeros(p, np.float32)
    mu = C.c_float(); ve = C.c_float(); h2 = C.c_float(); msx = C.c_float(); Pi = C.c_float()
    lb = np.zeros(p, np.float32); ld = np.zeros(p, np.float32); le = np.zeros(n, np.float32)
    lvb = np.zeros(p, np.float32); ls = np.zeros(4, np.float32)
    rc = getattr(lib(fast), "oracle_bayes_" + flavour)(
        C.c_int(MODELS[model]), _fp(y), _fp(Xf), C.c_int64(n), C.c_int64(p), C.c_int64(n),
        C.c_float(it), C.c_float(bi), C.c_float(pi), C.c_float(df), C.c_float(R2), C.c_uint64(seed), C.c_int(rng_mode),
        C.byref(mu), _fp(B), _fp(D), _fp(hat), _fp(VB), C.byref(ve), C.byref(h2), C.byref(msx), C.byref(Pi), _fp(PVAL),
        _fp(lb), _fp(ld), _fp(le), _fp(lvb), _fp(ls))
    assert rc == 0
    out = {"mu": mu.value, "b": B}
    if model in ("BayesB", "BayesC", "BayesCpi", "BayesDpi"):
        out["d"] = D
    if model in ("BayesCpi", "BayesDpi"):
        out.update({"pi": Pi.value, "hat": hat, "h2": h2.value, "vb": VB if per else float(VB[0]), "ve": ve.value, "PVAL": PVAL})
    else:
        out.update({"hat": hat, "vb": VB if per else float(VB[0]), "ve": ve.value, "h2": h2.value, "MSx": msx.value})
    out["last"] = {"b": lb, "d": ld, "e": le, "vb": lvb, "mu": float(ls[0]), "ve": float(ls[1]), "vb_common": float(ls[2]), "pi": float(ls[3])}
    return out


def bayes2(model, y, X1, X2, it=1500, bi=500, pi=0.95, df=5.0, R2=0.5, seed=1, rng_mode=0, flavour="w", fast=False):
    """Reference BayesA2 / BayesB2 / BayesRR2(y,X1,X2,it,bi,[pi,]df,R2), src/Rcpp20260726ai.cpp:990-1218.
    Returns the reference's list as a dict plus 'last' (b1, b2, e after the final iteration)."""
    m2 = {"BayesA2": 0, "BayesB2": 1, "BayesRR2": 2}[model]
    X1f, X2f = as_f32_colmajor(X1), as_f32_colmajor(X2)
    n, p1 = X1f.shape; p2 = X2f.shape[1]
    assert X2f.shape[0] == n
    y = np.ascontiguousarray(y, np.float32)
    per = m2 != 2
    B1 = np.zeros(p1, np.float32); D1 = np.zeros(p1, np.float32); VB1 = np.zeros(p1 if per else 1, np.float32)
    B2 = np.zeros(p2, np.float32); D2 = np.zeros(p2, np.float32); VB2 = np.zeros(p2 if per else 1, np.float32)
    hat = np.zeros(n, np.float32); mu = C.c_float(); ve = C.c_float(); h2 = C.c_float()
    lb1 = np.zeros(p1, np.float32); lb2 = np.zeros(p2, np.float32); le = np.zeros(n, np.float32); ls = np.zeros(2, np.float32)
    rc = getattr(lib(fast), "oracle_bayes2_" + flavour)(
        C.c_int(m2), _fp(y), _fp(X1f), C.c_int64(p1), _fp(X2f), C.c_int64(p2), C.c_int64(n),
        C.c_float(it), C.c_float(bi), C.c_float(pi), C.c_float(df), C.c_float(R2), C.c_uint64(seed), C.c_int(rng_mode),
        C.byref(mu), _fp(B1), _fp(D1), _fp(VB1), _fp(B2), _fp(D2), _fp(VB2), C.byref(ve), _fp(hat), C.byref(h2),
        _fp(lb1), _fp(lb2), _fp(le), _fp(ls))
    assert rc == 0
    vb1 = VB1 if per else float(VB1[0]); vb2 = VB2 if per else float(VB2[0])
    if m2 == 1:   # list order of :1146-1149
        out = {"mu": mu.value, "b1": B1, "d1": D1, "vb1": vb1, "b2": B2, "d2": D2, "vb2": vb2, "ve": ve.value, "hat": hat, "h2": h2.value}
    else:         # :1054-1057, :1216-1219
        out = {"hat": hat, "mu": mu.value, "b1": B1, "b2": B2, "vb1": vb1, "vb2": vb2, "ve": ve.value, "h2": h2.value}
    out["last"] = {"b1": lb1, "b2": lb2, "e": le, "mu": float(ls[0]), "ve": float(ls[1])}
    return out


def bag_rows(seed, it, n, k, rp=False):
    use = np.zeros(k, np.int32)
    rc = lib().oracle_bag_rows_w(C.c_uint64(seed), C.c_uint32(it), C.c_int64(n), C.c_int64(k), C.c_int(int(rp)), use.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0
    return use


def kmup2(X, Use, b, d, xx, E, L, Ve, pi, seed=1, it=0, rng_mode=0, stable=1, flavour="w"):
    """Reference KMUP2(X,Use,b,d,xx,E,L,Ve,pi) -> dict(b,d,e), src/Rcpp20260726ai.cpp:41-77 (Use 0-based)."""
    Xf = as_f32_colmajor(X)
    n0, p = Xf.shape
    use = np.ascontiguousarray(Use, np.int32); n = use.size
    b = np.array(b, np.float32); d = np.array(d, np.float32); E = np.ascontiguousarray(E, np.float32)
    xx = np.ascontiguousarray(xx, np.float32); L = np.ascontiguousarray(L, np.float32); e = np.zeros(n, np.float32)
    rc = getattr(lib(), "oracle_kmup2_" + flavour)(
        _fp(Xf), C.c_int64(n0), C.c_int64(p), C.c_int64(n0), use.ctypes.data_as(C.POINTER(C.c_int)), C.c_int64(n), _fp(b), _fp(d),
        _fp(xx), _fp(E), _fp(e), _fp(L), C.c_float(Ve), C.c_float(pi), C.c_uint64(seed), C.c_uint32(it), C.c_int(rng_mode),
        C.c_int(stable), C.c_uint32(0))
    assert rc == 0
    return {"b": b, "d": d, "e": e}


def eigk_truncate(eigK, VarK):
    """R/wgr.R:23-27: pk = which.max((cumsum(V)/length(V)) > VarK); first pk eigenpairs."""
    V = np.asarray(eigK["values"], np.float64)
    U = np.asarray(eigK["vectors"], np.float64)
    pk = int(np.argmax((np.cumsum(V) / V.size) > VarK)) + 1
    return np.asfortranarray(U[:, :pk]), np.ascontiguousarray(V[:pk]), pk


def wgr(y, X, it=1500, bi=500, th=1, bag=1.0, rp=False, iv=False, de=False, pi=0.0, df=5.0, R2=0.5, eigK=None, VarK=0.95, seed=1,
        rng_mode=0, stable=1, flavour="w"):
    """Reference wgr(y,X,it,bi,th,bag=1,rp=F,iv,de,pi,df,R2,eigK,VarK), R/wgr.R:2-169."""
    Xd = np.asfortranarray(np.asarray(X), dtype=np.float64)
    n, p = Xd.shape
    y = np.ascontiguousarray(y, np.float64)
    per = bool(iv or de)
    b = np.zeros(p); d = np.zeros(p); Vb = np.zeros(p if per else 1); hat = np.zeros(n); u = np.zeros(n)
    mu = C.c_double(); Ve = C.c_double(); cxx = C.c_double(); Vk = C.c_double()
    if eigK is not None:
        U, V, pk = eigk_truncate(eigK, VarK)
        Up, Vp = _dp(U), _dp(V)
    else:
        U, V, pk, Up, Vp = None, None, 0, None, None
    rc = getattr(lib(), "oracle_wgr_" + flavour)(
        _dp(y), _dp(Xd), C.c_int64(n), C.c_int64(p), C.c_int64(n), C.c_int(it), C.c_int(bi), C.c_int(th),
        C.c_int(int(iv)), C.c_int(int(de)), C.c_double(pi), C.c_double(df), C.c_double(R2), C.c_uint64(seed),
        C.c_int(rng_mode), C.c_int(stable), Up, Vp, C.c_int64(pk), C.c_double(bag), C.c_int(int(rp)),
        C.byref(mu), _dp(b), _dp(Vb), _dp(d), C.byref(Ve), _dp(hat), C.byref(cxx), _dp(u), C.byref(Vk))
    assert rc == 0
    if eigK is not None:
        return {"mu": mu.value, "b": b, "Vb": Vb if per else float(Vb[0]), "d": d, "Ve": Ve.value, "hat": hat, "u": u,
                "Vk": Vk.value, "cxx": cxx.value}
    return {"mu": mu.value, "b": b, "Vb": Vb if per else float(Vb[0]), "d": d, "Ve": Ve.value, "hat": hat, "cxx": cxx.value}


EM_MODELS = {"emRR": 0, "emBA": 1, "emDE": 2, "emML": 3, "emBB": 4, "emBC": 5, "emBCpi": 6, "emBL": 7, "emEN": 8, "lasso": 9}


def em_order(p, upto):
    """Marker order of sweep `upto` (0-based) of the EM family: the oracle's restatement of libstdc++'s
    std::shuffle(order, std::mt19937(i)) applied for i = 0..upto (src/Rcpp20260726ai.cpp:103)."""
    out = np.zeros(int(p), np.int32)
    rc = lib().oracle_em_order_w(C.c_int64(p), C.c_int(upto), out.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0
    return out


def em(model, y, X, df=10.0, R2=0.5, Pi=0.75, alpha=0.02, D=None, maxit=0, flavour="w", fast=False):
    """Reference emRR / emBA / emBB / emBC / emBCpi / emDE / emBL / emEN / emML (src/Rcpp20260726ai.cpp:80-521, :1502-1550);
    returns the reference's list as a dict plus 'iters' (sweeps run)."""
    Xf = as_f32_colmajor(X)
    n, p = Xf.shape
    y = np.ascontiguousarray(y, np.float32)
    b = np.zeros(p, np.float32); d = np.zeros(p, np.float32); hat = np.zeros(n, np.float32); vbv = np.zeros(p, np.float32)
    s = np.zeros(6, np.float32); mu = C.c_float(); iters = C.c_int()
    Dv = None if D is None else np.ascontiguousarray(D, np.float32)
    par = Pi if model in ("emBB", "emBC", "emBCpi") else alpha
    rc = getattr(lib(fast), "oracle_em_" + flavour)(
        C.c_int(EM_MODELS[model]), _fp(y), _fp(Xf), C.c_int64(n), C.c_int64(p), C.c_int64(n), C.c_float(df), C.c_float(R2),
        C.c_float(par), None if Dv is None else _fp(Dv), C.c_int(maxit), C.byref(mu), _fp(b), _fp(d), _fp(hat), _fp(vbv), _fp(s),
        C.byref(iters))
    assert rc == 0
    s = [float(v) for v in s]
    if model == "emRR":
        out = {"mu": mu.value, "b": b, "hat": hat, "Va": s[0], "Ve": s[1], "h2": s[2]}
    elif model in ("emBA", "emDE"):
        out = {"mu": mu.value, "b": b, "hat": hat, "Vb": vbv, "Ve": s[1], "h2": s[2]}
    elif model == "emBB":
        out = {"mu": mu.value, "b": b, "d": d, "hat": hat, "Vb": vbv, "Ve": s[1], "h2": s[2]}
    elif model == "emBC":
        out = {"mu": mu.value, "b": b, "d": d, "hat": hat, "Vg": s[3], "Va": s[0], "Ve": s[1], "h2": s[2]}
    elif model == "emBCpi":
        out = {"mu": mu.value, "b": b, "d": d, "pi": s[4], "hat": hat, "Vg": s[3], "Va": s[0], "Ve": s[1], "h2": s[2]}
    elif model == "emBL":
        out = {"mu": mu.value, "b": b, "hat": hat, "h2": s[2]}
    elif model == "emEN":
        out = {"mu": mu.value, "b": b, "hat": hat, "Va": s[0], "Ve": s[1], "h2": s[2]}
    elif model == "lasso":
        out = {"mu": mu.value, "b": b, "h2": s[2], "hat": hat, "Lmb": s[0]}
    else:
        out = {"mu": mu.value, "b": b, "hat": hat, "h2": s[2], "Vb": s[0], "Va": s[3], "Ve": s[1]}
    out["iters"] = int(iters.value)
    return out
