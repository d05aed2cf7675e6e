"""Host-side mirror of the reference's R interface for the Gibbs hot path.

Same names, argument order, defaults and return-list element names/order as
  KMUP(X,b,d,xx,e,L,Ve,pi)                          R/RcppExports.R:4-6
  BayesA/BayesL/BayesRR/BayesCpi/BayesDpi(y,X,it=1500,bi=500,df=5,R2=0.5)   R/RcppExports.R:48-74
  BayesB/BayesC(y,X,it=1500,bi=500,pi=0.95,df=5,R2=0.5)
  wgr(y,X,it=1500,bi=500,th=1,bag=1,rp=FALSE,iv=FALSE,de=FALSE,pi=0,df=5,R2=0.5,eigK=NULL,VarK=0.95,verb=FALSE)
                                                    R/wgr.R:2-8
R is not available in the build image, so the host layer above the C ABI is Python (see
INTEGRATION.md for the R .Call shim that binds the same entry points).  Everything numerical happens in
libbwgr_hip.so on the GPU; this module only marshals arrays.  `seed=None` draws the seed from numpy's
global stream, so numpy.random.seed() plays the role of R's set.seed().
"""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import BwgrError, c_f, c_d, check

MODELS = {"BayesA": 0, "BayesB": 1, "BayesC": 2, "BayesL": 3, "BayesRR": 4, "BayesCpi": 5, "BayesDpi": 6}
_PER_MARKER_VB = {"BayesA", "BayesB", "BayesL", "BayesDpi"}
X_I8, X_F32, X_F64 = 0, 1, 2
HOST, DEVICE = 0, 1


def _seed(seed):
    if seed is None:
        return int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))
    return int(seed) & 0xFFFFFFFFFFFFFFFF


def _fp(a):
    return a.ctypes.data_as(c_f)


def _dp(a):
    return a.ctypes.data_as(c_d)


class Panel:
    """Genotype matrix X staged once in HBM (column-major, rows zero-padded to the slab grid), together with
    xx, vx, MSx (src/Rcpp20260726ai.cpp:593-598) and the block-diagonal Gram used by the blocked sweep.

    X may be a numpy array (n x p; int8, float32 or float64 -- an all-integer float matrix within int8 range is
    stored as int8 unless as_int8=False) or a torch CUDA tensor holding the column-major matrix as shape
    (p, ldx) int8/float32 (row j of the tensor = column j of X)."""

    def __init__(self, X, n=None, device=0, block=0, nwg=0, as_int8=None):
        L = _lib.lib()
        self._h = C.c_void_p()
        self._keep = None
        if hasattr(X, "data_ptr"):  # torch tensor on the GPU, (p, ldx)
            import torch
            assert X.is_cuda and X.dim() == 2 and X.is_contiguous(), "device X must be a contiguous (p, ldx) CUDA tensor"
            p, ldx = X.shape
            n = ldx if n is None else int(n)
            xtype = {torch.int8: X_I8, torch.float32: X_F32, torch.float64: X_F64}[X.dtype]
            device = X.device.index or 0
            torch.cuda.synchronize(X.device)
            check(L.bwgr_panel_create(C.byref(self._h), C.c_void_p(X.data_ptr()), xtype, DEVICE, n, p, ldx, device, block, nwg))
        else:
            X = np.asarray(X)
            assert X.ndim == 2, "X must be n x p"
            n, p = X.shape
            if X.dtype != np.int8:
                fits = bool(X.size == 0 or (X.min() >= -128 and X.max() <= 127))   # a wider integer must not wrap
                if as_int8 is None:
                    as_int8 = fits and bool(np.issubdtype(X.dtype, np.integer) or (X.size and np.all(X == np.rint(X))))
                if as_int8:
                    if not fits or not (np.issubdtype(X.dtype, np.integer) or np.all(X == np.rint(X))):
                        raise ValueError("Panel(as_int8=True): X holds values outside -128..127 or non-integers")
                    X = X.astype(np.int8)
                elif X.dtype not in (np.float32, np.float64):
                    X = X.astype(np.float64)
            Xc = np.asfortranarray(X)
            xtype = {np.dtype(np.int8): X_I8, np.dtype(np.float32): X_F32, np.dtype(np.float64): X_F64}[Xc.dtype]
            check(L.bwgr_panel_create(C.byref(self._h), Xc.ctypes.data_as(C.c_void_p), xtype, HOST, n, p, n, device, block, nwg))
        info = (C.c_int64 * 8)()
        check(L.bwgr_panel_info(self._h, info))
        self.n, self.p, self.ld, self.block, self.nwg, self.slab_rows, self.x_bytes, self.gram_bytes = [int(v) for v in info]
        self.device = device

    def set_stream(self, stream_ptr):
        check(_lib.lib().bwgr_panel_set_stream(self._h, C.c_void_p(stream_ptr)))

    def clone(self):
        """A second handle on the same resident genotypes with its own sweep scratch and stream (bwgr_panel_clone):
        chains on a panel and on its clones run side by side on the GPU.  Closed with (or before) the parent."""
        import weakref
        q = Panel.__new__(Panel)
        q._h = C.c_void_p(); q._keep = None
        root = getattr(self, "_parent", None) or self
        check(_lib.lib().bwgr_panel_clone(C.byref(q._h), root._h))
        q._parent = root
        for k in ("n", "p", "ld", "block", "nwg", "slab_rows", "x_bytes", "gram_bytes", "device"):
            setattr(q, k, getattr(root, k))
        if not hasattr(root, "_clones"):
            root._clones = weakref.WeakSet()
        root._clones.add(q)
        return q

    def max_pairs(self):
        """How many pairs of chains (Chain.run_pair) fit the chip at once (bwgr_panel_max_pairs); 0 without k_sweep3."""
        c = C.c_int(0)
        check(_lib.lib().bwgr_panel_max_pairs(self._h, C.byref(c)))
        return c.value

    def max_concurrent(self, selection):
        """How many sweeps of this geometry fit the chip at once (bwgr_panel_max_concurrent)."""
        c = C.c_int()
        check(_lib.lib().bwgr_panel_max_concurrent(self._h, int(bool(selection)), C.byref(c)))
        return int(c.value)

    def centred(self):
        """True when every column's |mean| <= 1e-3 sd (from the panel's own statistics): what the marker-sharded sampler needs to be sound."""
        k = C.c_int()
        check(_lib.lib().bwgr_panel_centred(self._h, C.byref(k)))
        return bool(k.value)

    def set_centred(self, on=True):
        """Sweep the implicitly centred columns x_j - mean(x_j) of this int8 panel from now on (bwgr_panel_set_centred): the genotypes stay
        int8 in HBM and the kernels those of the raw columns; the fused chains (Chain, BayesB / C / Cpi / Dpi) then run the reference's sweep on
        the centred columns, stats() returns their squared norms and centred() is True -- what the marker-sharded sampler needs."""
        check(_lib.lib().bwgr_panel_set_centred(self._h, int(bool(on))))
        return self

    def pipeline(self, selection):
        """How a sweep over this panel is pipelined: dict(generation, lag, feeders, gram_bits) (bwgr_panel_pipeline)."""
        info = (C.c_int * 4)()
        check(_lib.lib().bwgr_panel_pipeline(self._h, int(bool(selection)), info))
        return dict(zip(("generation", "lag", "feeders", "gram_bits"), (int(v) for v in info)))

    def stats(self):
        xx = np.empty(self.p, np.float32); vx = np.empty(self.p, np.float32); msx = C.c_float()
        check(_lib.lib().bwgr_panel_stats(self._h, _fp(xx), _fp(vx), C.byref(msx)))
        return xx, vx, float(msx.value)

    def close(self):
        if self._h:
            for q in list(getattr(self, "_clones", ())):
                q.close()
            check(_lib.lib().bwgr_panel_destroy(self._h))   # refuses while chains or clones are alive: the handle stays valid
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _as_panel(X, **kw):
    return (X, False) if isinstance(X, Panel) else (Panel(X, **kw), True)


class Chain:
    """One MCMC chain of a fused sampler, stepping interface over bwgr_chain_* (state stays on the GPU)."""

    def __init__(self, panel, model, y, it=1500, bi=500, pi=0.95, df=5.0, R2=0.5, seed=None, rng_mode=0, shard=None,
                 e_ext=None):
        """shard = (marker0, p_total, MSx_total) makes this the chain of one marker shard (bwgr_chain_create_sharded);
        e_ext = a torch float64 CUDA tensor of panel.ld entries that will hold the (replicated) residual."""
        self.panel, self.model = panel, model
        self._h = C.c_void_p()
        self.it, self.bi = int(it), int(bi)
        if hasattr(y, "data_ptr"):
            import torch
            assert y.is_cuda and y.dtype == torch.float32 and y.is_contiguous() and y.numel() == panel.n
            torch.cuda.synchronize(y.device)
            yptr, loc = C.c_void_p(y.data_ptr()), DEVICE
        else:
            self._y = np.ascontiguousarray(y, np.float32)
            assert self._y.size == panel.n, "length(y) must equal nrow(X)"
            yptr, loc = self._y.ctypes.data_as(C.c_void_p), HOST
        if shard is None and e_ext is None:
            check(_lib.lib().bwgr_chain_create(C.byref(self._h), panel._h, MODELS[model], yptr, loc, float(it), float(bi),
                                                float(pi), float(df), float(R2), C.c_uint64(_seed(seed)), int(rng_mode)))
        else:
            marker0, p_total, msx_total = shard if shard is not None else (0, panel.p, panel.stats()[2])
            eptr = None
            if e_ext is not None:
                import torch
                assert e_ext.is_cuda and e_ext.dtype == torch.float64 and e_ext.is_contiguous() and e_ext.numel() == panel.ld
                self._e_ext = e_ext
                eptr = C.c_void_p(e_ext.data_ptr())
            check(_lib.lib().bwgr_chain_create_sharded(C.byref(self._h), panel._h, MODELS[model], yptr, loc, float(it), float(bi),
                                                        float(pi), float(df), float(R2), C.c_uint64(_seed(seed)), int(rng_mode),
                                                        int(marker0), int(p_total), float(msx_total), eptr))
        self.nblocks = (panel.p + panel.block - 1) // panel.block

    def run(self, iters):
        check(_lib.lib().bwgr_chain_run(self._h, int(iters)))

    def run_pair(self, other, iters):
        """Advance this chain and `other` (a chain on a clone of the same resident panel) `iters` iterations in lockstep on one set
        of streamer workgroups -- one pass over the genotypes for both (bwgr_chain_run_pair).  Each chain's results are the ones it
        would have alone."""
        check(_lib.lib().bwgr_chain_run_pair(self._h, other._h, int(iters)))

    def sync(self):
        check(_lib.lib().bwgr_chain_sync(self._h))

    def sweep_blocks(self, lo, hi):
        check(_lib.lib().bwgr_chain_sweep_blocks(self._h, int(lo), int(hi)))

    def round_sweep(self, lo, hi, delta):
        """Exchange round, first half (bwgr_chain_round_sweep): delta (torch float64 CUDA tensor of panel.ld entries) receives
        e - e_before after sweeping blocks [lo, hi)."""
        check(_lib.lib().bwgr_chain_round_sweep(self._h, int(lo), int(hi), C.c_void_p(delta.data_ptr())))

    def round_apply(self, delta):
        check(_lib.lib().bwgr_chain_round_apply(self._h, C.c_void_p(delta.data_ptr())))

    def get_sums_dev(self, t):
        """{sum d, sum b^2} of the blocks swept so far into t (torch float64 CUDA tensor, 2 entries): no host round trip."""
        check(_lib.lib().bwgr_chain_get_sums_dev(self._h, C.c_void_p(t.data_ptr())))

    def end_iteration_dev(self, t):
        check(_lib.lib().bwgr_chain_end_iteration_dev(self._h, C.c_void_p(t.data_ptr())))

    def get_sums(self):
        s = np.zeros(2, np.float64)
        check(_lib.lib().bwgr_chain_get_sums(self._h, _dp(s)))
        return s

    def end_iteration(self, sums_total=None):
        if sums_total is None:
            check(_lib.lib().bwgr_chain_end_iteration(self._h, None))
        else:
            s = np.ascontiguousarray(sums_total, np.float64)
            check(_lib.lib().bwgr_chain_end_iteration(self._h, _dp(s)))

    def sweep_ms(self):
        ms = C.c_float(); nl = C.c_int()
        check(_lib.lib().bwgr_chain_sweep_ms(self._h, C.byref(ms), C.byref(nl)))
        return float(ms.value), int(nl.value)

    def redo_count(self):
        """Sweeps that left the fixed-point range of their engine and were redone on the fp64 residual."""
        k = C.c_int()
        check(_lib.lib().bwgr_chain_redo_count(self._h, C.byref(k)))
        return int(k.value)

    def state(self):
        p, n = self.panel.p, self.panel.n
        b = np.empty(p, np.float32); d = np.empty(p, np.float32); e = np.empty(n, np.float32)
        vb = np.empty(p, np.float32); s = np.empty(4, np.float32)
        check(_lib.lib().bwgr_chain_state(self._h, _fp(b), _fp(d), _fp(e), _fp(vb), _fp(s)))
        return {"b": b, "d": d, "e": e, "vb": vb, "mu": float(s[0]), "ve": float(s[1]), "vb_common": float(s[2]), "pi": float(s[3])}

    def result(self):
        """The reference's return list (names and order), src/Rcpp20260726ai.cpp:631-634, 694-698, 916-920."""
        p, n, model = self.panel.p, self.panel.n, self.model
        per = model in _PER_MARKER_VB
        B = np.empty(p, np.float32); D = np.empty(p, np.float32); hat = np.empty(n, np.float32)
        VB = np.empty(p if per else 1, np.float32); PV = np.empty(p, np.float32)
        mu = C.c_float(); ve = C.c_float(); h2 = C.c_float(); msx = C.c_float(); Pi = C.c_float()
        check(_lib.lib().bwgr_chain_result(self._h, C.byref(mu), _fp(B), _fp(D), _fp(hat), _fp(VB), C.byref(ve), C.byref(h2),
                                            C.byref(msx), C.byref(Pi), _fp(PV)))
        vb = VB if per else float(VB[0])
        if model in ("BayesA", "BayesL", "BayesRR"):
            return {"mu": mu.value, "b": B, "hat": hat, "vb": vb, "ve": ve.value, "h2": h2.value, "MSx": msx.value}
        if model in ("BayesB", "BayesC"):
            return {"mu": mu.value, "b": B, "d": D, "hat": hat, "vb": vb, "ve": ve.value, "h2": h2.value, "MSx": msx.value}
        return {"mu": mu.value, "b": B, "d": D, "pi": Pi.value, "hat": hat, "h2": h2.value, "vb": vb, "ve": ve.value, "PVAL": PV}

    def close(self):
        if self._h:
            _lib.lib().bwgr_chain_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _fused(model, y, X, it, bi, pi, df, R2, seed, rng_mode, return_state=False, **panel_kw):
    P, own = _as_panel(X, **panel_kw)
    ch = Chain(P, model, y, it, bi, pi, df, R2, seed, rng_mode)
    try:
        ch.run(int(it))
        out = ch.result()
        if return_state:
            out["last"] = ch.state()
        return out
    finally:
        ch.close()
        if own:
            P.close()


def BayesA(y, X, it=1500, bi=500, df=5, R2=0.5, *, seed=None, rng_mode=0, **kw):
    return _fused("BayesA", y, X, it, bi, 0.0, df, R2, seed, rng_mode, **kw)


def BayesB(y, X, it=1500, bi=500, pi=0.95, df=5, R2=0.5, *, seed=None, rng_mode=0, **kw):
    return _fused("BayesB", y, X, it, bi, pi, df, R2, seed, rng_mode, **kw)


def BayesC(y, X, it=1500, bi=500, pi=0.95, df=5, R2=0.5, *, seed=None, rng_mode=0, **kw):
    return _fused("BayesC", y, X, it, bi, pi, df, R2, seed, rng_mode, **kw)


def BayesL(y, X, it=1500, bi=500, df=5, R2=0.5, *, seed=None, rng_mode=0, **kw):
    return _fused("BayesL", y, X, it, bi, 0.0, df, R2, seed, rng_mode, **kw)


def BayesRR(y, X, it=1500, bi=500, df=5, R2=0.5, *, seed=None, rng_mode=0, **kw):
    return _fused("BayesRR", y, X, it, bi, 0.0, df, R2, seed, rng_mode, **kw)


def BayesCpi(y, X, it=1500, bi=500, df=5, R2=0.5, *, seed=None, rng_mode=0, **kw):
    return _fused("BayesCpi", y, X, it, bi, 0.0, df, R2, seed, rng_mode, **kw)


def BayesDpi(y, X, it=1500, bi=500, df=5, R2=0.5, *, seed=None, rng_mode=0, **kw):
    return _fused("BayesDpi", y, X, it, bi, 0.0, df, R2, seed, rng_mode, **kw)


class Group:
    """One fused-sampler chain over several GPUs of this process (bwgr_group_*, include/bwgr.h): device g takes the g-th
    block-aligned marker shard of the host matrix X, the residual is replicated and re-united by RCCL all-reduces at the
    exchange rounds, one host thread drives everything.  devices=[d] is the plain exact chain on one GPU; more devices
    run the partitioned sampler (sound on centred columns only: centre=True).  This is the path an R .Call takes to more than one GPU; the
    benchmark's one-process-per-GPU driver is bwgr_amd/dist.py."""

    def __init__(self, model, y, X, devices=(0,), it=1500, bi=500, pi=0.95, df=5.0, R2=0.5, seed=None, rng_mode=0, block=0,
                 markers_per_sync=0, centre=False, n=None):
        """centre=True sweeps x_j - mean(x_j): what makes more than one shard statistically sound (on uncentred columns the library refuses
        len(devices) > 1 unless BWGR_GROUP_ALLOW_UNCENTRED=1).  Integer genotypes under the selection models (BayesB / C / Cpi / Dpi) are centred
        IMPLICITLY (the panel stays int8 in HBM: bwgr_group_create_centred; centre="explicit" asks for the float copy); float columns and the affine
        models get an explicitly centred float panel.  Centring is a reparametrisation under the flat intercept
        prior (an exact Gibbs sampler would not notice; bWGR's own chain does a little: DESIGN.md section 8); result() gives mu back in the uncentred
        parametrisation, mu - sum_j mean_j b_j.
        devices may name ONE device several times: the shards then run side by side on that GPU (streams of their own, a sum kernel per exchange
        round, no RCCL).  X may then also be a torch int8 CUDA tensor of shape (p, ldx) on that device (row j = column j of X; n rows of it used)."""
        self._xbar = None
        self._keep = None
        implicit = False
        memloc = HOST
        if hasattr(X, "data_ptr"):     # device-resident int8 genotypes, shards of that one device, implicit centring
            import torch
            assert X.is_cuda and X.dim() == 2 and X.is_contiguous() and X.dtype == torch.int8, "device X: a contiguous (p, ldx) int8 CUDA tensor"
            assert centre and centre != "explicit", "device X: implicitly centred shards (centre=True)"
            assert all(int(d) == (X.device.index or 0) for d in devices), "device X serves shards of its own device only"
            p_, ldx = X.shape
            self.n, self.p = int(ldx if n is None else n), int(p_)
            xb = torch.empty(p_, dtype=torch.float64, device=X.device)
            for c0 in range(0, p_, 65536):
                xb[c0:c0 + 65536] = X[c0:c0 + 65536, :self.n].sum(dim=1, dtype=torch.int64).to(torch.float64) / self.n
            self._xbar = xb.cpu().numpy()
            torch.cuda.synchronize(X.device)
            self._keep = X
            implicit, memloc, xtype, xptr = True, DEVICE, X_I8, C.c_void_p(X.data_ptr())
        else:
            X = np.asarray(X)
            assert X.ndim == 2
            if centre:
                Xd = X.astype(np.float64)
                self._xbar = Xd.mean(0)
                implicit = (centre != "explicit" and model in ("BayesB", "BayesC", "BayesCpi", "BayesDpi") and X.size > 0
                            and bool(np.all(X == np.rint(X))) and X.min() >= -128 and X.max() <= 127)
                if not implicit:
                    X = (Xd - self._xbar).astype(np.float32)
            if X.dtype != np.int8:
                fits = bool(X.size == 0 or (X.min() >= -128 and X.max() <= 127))
                if fits and (np.issubdtype(X.dtype, np.integer) or np.all(X == np.rint(X))):
                    X = X.astype(np.int8)
                elif X.dtype not in (np.float32, np.float64):
                    X = X.astype(np.float64)
            self._X = np.asfortranarray(X)
            self.n, self.p = self._X.shape
            ldx = self.n
            xtype = {np.dtype(np.int8): X_I8, np.dtype(np.float32): X_F32, np.dtype(np.float64): X_F64}[self._X.dtype]
            xptr = self._X.ctypes.data_as(C.c_void_p)
        self.model = model
        if hasattr(y, "data_ptr"):
            y = y.detach().cpu().numpy()
        self._y = np.ascontiguousarray(y, np.float32)
        assert self._y.size == self.n
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._h = C.c_void_p()
        self.implicit_centring = implicit
        args = [C.byref(self._h), len(devices), devs, xptr, xtype, self.n, self.p, int(ldx), int(block), _fp(self._y), MODELS[model], float(it), float(bi),
                float(pi), float(df), float(R2), C.c_uint64(_seed(seed)), int(rng_mode), int(markers_per_sync)]
        if implicit:
            check(_lib.lib().bwgr_group_create_centred(*(args + [memloc])))
        else:
            check(_lib.lib().bwgr_group_create(*args))

    def info(self):
        v = (C.c_int64 * 4)()
        check(_lib.lib().bwgr_group_info(self._h, v))
        out = dict(zip(("devices", "rounds_per_sweep", "markers_per_round", "rccl"), (int(x) for x in v)))
        out["statistically_sound"] = self.sound()
        return out

    def sound(self):
        """True for one device (the exact chain) or centred columns; False for several devices on uncentred columns."""
        k = C.c_int()
        check(_lib.lib().bwgr_group_sound(self._h, C.byref(k)))
        return bool(k.value)

    def run(self, iters):
        check(_lib.lib().bwgr_group_run(self._h, int(iters)))

    def sync(self):
        check(_lib.lib().bwgr_group_sync(self._h))

    def result(self):
        p, n, model = self.p, self.n, self.model
        per = model in _PER_MARKER_VB
        B = np.empty(p, np.float32); D = np.empty(p, np.float32); hat = np.empty(n, np.float32)
        VB = np.empty(p if per else 1, np.float32); PV = np.empty(p, np.float32)
        mu = C.c_float(); ve = C.c_float(); h2 = C.c_float(); msx = C.c_float(); Pi = C.c_float()
        check(_lib.lib().bwgr_group_result(self._h, C.byref(mu), _fp(B), _fp(D), _fp(hat), _fp(VB), C.byref(ve), C.byref(h2),
                                            C.byref(msx), C.byref(Pi), _fp(PV)))
        vb = VB if per else float(VB[0])
        muv = mu.value
        if self._xbar is not None:      # centred columns: the intercept of the uncentred parametrisation (posterior means are linear in it)
            muv = float(muv - float(np.dot(self._xbar, B.astype(np.float64))))
        if model in ("BayesA", "BayesL", "BayesRR"):
            out = {"mu": muv, "b": B, "hat": hat, "vb": vb, "ve": ve.value, "h2": h2.value, "MSx": msx.value}
        elif model in ("BayesB", "BayesC"):
            out = {"mu": muv, "b": B, "d": D, "hat": hat, "vb": vb, "ve": ve.value, "h2": h2.value, "MSx": msx.value}
        else:
            out = {"mu": muv, "b": B, "d": D, "pi": Pi.value, "hat": hat, "h2": h2.value, "vb": vb, "ve": ve.value, "PVAL": PV}
        out["statistically_sound"] = self.sound()     # (beyond the reference's list: False for several devices on uncentred columns)
        return out

    def close(self):
        if self._h:
            _lib.lib().bwgr_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_SELECTION = ("BayesB", "BayesC", "BayesCpi", "BayesDpi")
_DEFAULT_PI = {"BayesB": 0.95, "BayesC": 0.95}


def fit_many(X, jobs, concurrent=None, chunk=4, pair=None, **panel_kw):
    """Several fused-sampler fits on ONE resident X, run side by side: jobs = [dict(model=, y=, it=, bi=, pi=, df=,
    R2=, seed=, rng_mode=), ...] (same defaults as the BayesX functions); returns the BayesX return lists in job order.
    Every job's chain is the one BayesX(y, X, ...) alone would run -- bit for bit -- because a chain's arithmetic does
    not depend on what else is on the chip: the panel is cloned (bwgr_panel_clone: shared genotypes and Gram, private
    sweep scratch and stream) once per slot, a sweep occupies nwg + 1 + feeders compute units, and up to
    Panel.max_concurrent() of them fit.  This is the shape of the reference's multi-fit callers (mcmcCV runs seven
    samplers per fold on the same training matrix, R/cv.R:124-130), which it runs one after the other.

    pair: sparse selection jobs (BayesB / BayesC with pi >= 0.95 on a panel that has k_sweep3: up to about 6 % of the markers in the model a pair does more per compute unit than two chains apart) run TWO to a set of streamer
    workgroups (Chain.run_pair: one pass over the genotypes serves both).  A paired chain is bit for bit the chain it is alone on
    k_sweep3; a single chain whose inclusion rate passes 3 % takes some sweeps on k_sweep2 instead (the device chooses), which
    agrees to ~1e-9, not bit for bit.  None = whenever at least two jobs qualify, False = never, True = every selection job
    (dense chains are slow on k_sweep3)."""
    P, own = _as_panel(X, **panel_kw)
    handles, results = [], [None] * len(jobs)
    live = []   # every chain still open (closed on the way out, whatever happens)
    try:
        norm = []
        for j in jobs:
            j = dict(j)
            model = j.pop("model"); y = j.pop("y")
            norm.append((model, y, int(j.pop("it", 1500)), int(j.pop("bi", 500)), float(j.pop("pi", _DEFAULT_PI.get(model, 0.0))),
                         float(j.pop("df", 5)), float(j.pop("R2", 0.5)), j.pop("seed", None), int(j.pop("rng_mode", 0))))
            if j:
                raise TypeError("fit_many: unknown job keys %s" % sorted(j))
        gen3 = P.pipeline(True)["generation"] == 3   # (int8 panels with k_sweep3)
        if pair is True:
            paired = [i for i, r in enumerate(norm) if r[0] in _SELECTION] if gen3 else []
        elif pair is None:
            paired = [i for i, r in enumerate(norm) if r[0] in ("BayesB", "BayesC") and r[4] >= 0.95] if gen3 else []
        else:
            paired = []
        if len(paired) < 2:
            paired = []
        single = [i for i in range(len(norm)) if i not in set(paired)]

        def make(i, handle):
            model, y, it, bi, pi, df, R2, seed, rng_mode = norm[i]
            ch = Chain(handle, model, y, it, bi, pi, df, R2, seed, rng_mode)
            live.append(ch)
            return [i, ch, it]

        def finish(rec):
            i, ch, _ = rec
            try:
                results[i] = ch.result()
            finally:
                ch.close(); live.remove(ch)

        def need(n):   # handles 0 .. n-1 exist
            while len(handles) < n:
                handles.append(P if not handles else P.clone())

        # ---- pairs: slot s owns handles 2s and 2s+1; a pair holds (streamers + 2) compute units ----
        if paired:
            cap = max(1, P.max_pairs())
            nslot = max(1, min((len(paired) + 1) // 2, cap if concurrent is None else min(int(concurrent), cap)))
            need(2 * nslot)
            queue, active = list(paired), {}

            def unpair(rec):   # BWGR_ERANGE: a pair sweep left the fixed-point range (no redo on that path, include/bwgr.h): the job again, alone
                i, ch, _ = rec
                ch.close(); live.remove(ch)
                single.append(i)

            while queue or active:
                for s_ in range(nslot):
                    rec = active.setdefault(s_, [None, None])
                    for h in (0, 1):
                        if rec[h] is None and queue:
                            rec[h] = make(queue.pop(0), handles[2 * s_ + h])
                for s_, rec in list(active.items()):
                    a_, b_ = rec
                    if a_ is not None and b_ is not None:
                        step = min(int(chunk), a_[2], b_[2])
                        try:
                            a_[1].run_pair(b_[1], step)
                            a_[2] -= step; b_[2] -= step
                        except BwgrError as ex:   # (surfaces when an earlier pair sweep's status is read: either chain may be the one)
                            if ex.code != 6:
                                raise
                            unpair(a_); unpair(b_); rec[0] = rec[1] = None
                    else:
                        for r in (a_, b_):
                            if r is not None:
                                step = min(int(chunk), r[2]); r[1].run(step); r[2] -= step
                for s_, rec in list(active.items()):
                    for h in (0, 1):
                        if rec[h] is not None and rec[h][2] == 0:
                            try:
                                finish(rec[h])
                            except BwgrError as ex:
                                if ex.code != 6:
                                    raise
                                single.append(rec[h][0])   # (finish closed the chain)
                            rec[h] = None
                    if rec[0] is None and rec[1] is None and not queue:
                        del active[s_]
        # ---- the rest, one chain per handle ----
        if single:
            cap = P.max_concurrent(any(norm[i][0] in _SELECTION for i in single))
            nslot = max(1, min(len(single), cap if concurrent is None else min(int(concurrent), cap)))
            need(nslot)
            queue, active = list(single), {}
            while queue or active:
                for slot in range(nslot):
                    if slot not in active and queue:
                        active[slot] = make(queue.pop(0), handles[slot])
                # a few iterations per chain per turn keeps every stream's launch queue fed without one chain hogging the host
                for slot in list(active):
                    rec = active[slot]
                    step = min(int(chunk), rec[2])
                    rec[1].run(step)
                    rec[2] -= step
                for slot in list(active):
                    if active[slot][2] == 0:
                        finish(active[slot])
                        del active[slot]
        return results
    finally:
        for ch in list(live):
            ch.close()
        for h in handles[1:]:
            h.close()
        if own:
            P.close()


_EM = {"emRR": 0, "emBA": 1, "emDE": 2, "emML": 3, "emBB": 4, "emBC": 5, "emBCpi": 6, "emBL": 7, "emEN": 8, "lasso": 9}


def _em(model, y, gen, df=10.0, R2=0.5, par=0.0, D=None, maxit=0, **panel_kw):
    """The EM / Gauss-Seidel family over bwgr_em (include/bwgr.h); returns dict(mu, b, d, hat, vbvec, scal, iters)."""
    P, own = _as_panel(gen, **panel_kw)
    try:
        yv = np.ascontiguousarray(y, np.float32)
        assert yv.size == P.n, "length(y) must equal nrow(gen)"
        b = np.empty(P.p, np.float32); d = np.zeros(P.p, np.float32); hat = np.empty(P.n, np.float32); vbv = np.empty(P.p, np.float32)
        scal = np.zeros(6, np.float32); mu = C.c_float(); iters = C.c_int()
        Dv = None if D is None else np.ascontiguousarray(D, np.float32)
        if Dv is not None:
            assert Dv.size == P.p, "length(D) must equal ncol(gen)"
        check(_lib.lib().bwgr_em(P._h, _EM[model], _fp(yv), float(df), float(R2), float(par), None if Dv is None else _fp(Dv),
                                 int(maxit), C.byref(mu), _fp(b), _fp(d), _fp(hat), _fp(vbv), _fp(scal), C.byref(iters)))
        return {"mu": float(mu.value), "b": b, "d": d, "hat": hat, "vbvec": vbv, "scal": [float(v) for v in scal], "iters": int(iters.value)}
    finally:
        if own:
            P.close()


def emRR(y, gen, df=10, R2=0.5, **kw):
    """emRR(y, gen, df = 10, R2 = 0.5), src/Rcpp20260726ai.cpp:308-354: list(mu, b, hat, Va, Ve, h2)."""
    r = _em("emRR", y, gen, df, R2, **kw); s = r["scal"]
    return {"mu": r["mu"], "b": r["b"], "hat": r["hat"], "Va": s[0], "Ve": s[1], "h2": s[2]}


def emBA(y, gen, df=10, R2=0.5, **kw):
    """emBA(y, gen, df = 10, R2 = 0.5), src/Rcpp20260726ai.cpp:80-128: list(mu, b, hat, Vb, Ve, h2)."""
    r = _em("emBA", y, gen, df, R2, **kw); s = r["scal"]
    return {"mu": r["mu"], "b": r["b"], "hat": r["hat"], "Vb": r["vbvec"], "Ve": s[1], "h2": s[2]}


def emBB(y, gen, df=10, R2=0.5, Pi=0.75, **kw):
    """emBB(y, gen, df = 10, R2 = 0.5, Pi = 0.75), src/Rcpp20260726ai.cpp:131-187: list(mu, b, d, hat, Vb, Ve, h2)."""
    r = _em("emBB", y, gen, df, R2, Pi, **kw); s = r["scal"]
    return {"mu": r["mu"], "b": r["b"], "d": r["d"], "hat": r["hat"], "Vb": r["vbvec"], "Ve": s[1], "h2": s[2]}


def emBC(y, gen, df=10, R2=0.5, Pi=0.75, **kw):
    """emBC(y, gen, df = 10, R2 = 0.5, Pi = 0.75), src/Rcpp20260726ai.cpp:190-247: list(mu, b, d, hat, Vg, Va, Ve, h2)."""
    r = _em("emBC", y, gen, df, R2, Pi, **kw); s = r["scal"]
    return {"mu": r["mu"], "b": r["b"], "d": r["d"], "hat": r["hat"], "Vg": s[3], "Va": s[0], "Ve": s[1], "h2": s[2]}


def emBCpi(y, gen, df=10, R2=0.5, Pi=0.75, **kw):
    """emBCpi(y, gen, df = 10, R2 = 0.5, Pi = 0.75), src/Rcpp20260726ai.cpp:1502-1550 (natural marker order):
    list(mu, b, d, pi, hat, Vg, Va, Ve, h2)."""
    r = _em("emBCpi", y, gen, df, R2, Pi, **kw); s = r["scal"]
    return {"mu": r["mu"], "b": r["b"], "d": r["d"], "pi": s[4], "hat": r["hat"], "Vg": s[3], "Va": s[0], "Ve": s[1], "h2": s[2]}


def emDE(y, gen, R2=0.5, **kw):
    """emDE(y, gen, R2 = 0.5), src/Rcpp20260726ai.cpp:250-305: list(mu, b, hat, Vb, Ve, h2)."""
    r = _em("emDE", y, gen, 0.0, R2, **kw); s = r["scal"]
    return {"mu": r["mu"], "b": r["b"], "hat": r["hat"], "Vb": r["vbvec"], "Ve": s[1], "h2": s[2]}


def emBL(y, gen, R2=0.5, alpha=0.02, **kw):
    """emBL(y, gen, R2 = 0.5, alpha = 0.02), src/Rcpp20260726ai.cpp:357-397: list(mu, b, hat, h2)."""
    r = _em("emBL", y, gen, 0.0, R2, alpha, **kw)
    return {"mu": r["mu"], "b": r["b"], "hat": r["hat"], "h2": r["scal"][2]}


def emEN(y, gen, R2=0.5, alpha=0.02, **kw):
    """emEN(y, gen, R2 = 0.5, alpha = 0.02), src/Rcpp20260726ai.cpp:400-460: list(mu, b, hat, Va, Ve, h2)."""
    r = _em("emEN", y, gen, 0.0, R2, alpha, **kw); s = r["scal"]
    return {"mu": r["mu"], "b": r["b"], "hat": r["hat"], "Va": s[0], "Ve": s[1], "h2": s[2]}


def lasso(y, gen, **kw):
    """lasso(y, gen), src/Rcpp20260726ai.cpp:1463-1500 (natural marker order): list(mu, b, h2, hat, Lmb)."""
    r = _em("lasso", y, gen, 0.0, 0.5, 0.0, **kw); s = r["scal"]
    return {"mu": r["mu"], "b": r["b"], "h2": s[2], "hat": r["hat"], "Lmb": s[0]}


def emML(y, gen, D=None, **kw):
    """emML(y, gen, D = NULL), src/Rcpp20260726ai.cpp:463-521: list(mu, b, hat, h2, Vb, Va, Ve)."""
    r = _em("emML", y, gen, 0.0, 0.5, 0.0, D=D, **kw); s = r["scal"]
    return {"mu": r["mu"], "b": r["b"], "hat": r["hat"], "h2": s[2], "Vb": s[0], "Va": s[3], "Ve": s[1]}


def em_order(p, upto):
    """Marker order of sweep `upto` (0-based) of the EM family: std::shuffle with std::mt19937(0..upto) (host only)."""
    out = np.zeros(int(p), np.int32)
    check(_lib.lib().bwgr_em_order(int(p), int(upto), out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out


def _fused2(base, y, X1, X2, it, bi, pi, df, R2, seed, rng_mode, **panel_kw):
    """BayesA2 / BayesB2 / BayesRR2(y, X1, X2, ...), src/Rcpp20260726ai.cpp:990-1218: two panels, one residual."""
    P1, own1 = _as_panel(X1, **panel_kw)
    own2 = not isinstance(X2, Panel)
    if not own2:
        P2 = X2
    else:
        # the two panels share the residual, hence the slab geometry: try X1's workgroup count for X2, else X2's for X1
        # (an fp32 panel takes fewer rows per slab than an int8 one)
        kw2 = dict(panel_kw); kw2.setdefault("nwg", P1.nwg)
        try:
            P2 = Panel(X2, **kw2)
        except BwgrError:
            if not own1:
                raise
            P2 = Panel(X2, **panel_kw)
            P1.close()
            kw1 = dict(panel_kw); kw1["nwg"] = P2.nwg
            P1 = Panel(X1, **kw1)
    try:
        assert P1.n == P2.n, "X1 and X2 must have the same rows"
        y = np.ascontiguousarray(y, np.float32)
        assert y.size == P1.n
        per = base != "BayesRR"
        b1 = np.zeros(P1.p, np.float32); d1 = np.zeros(P1.p, np.float32); vb1 = np.zeros(P1.p if per else 1, np.float32)
        b2 = np.zeros(P2.p, np.float32); d2 = np.zeros(P2.p, np.float32); vb2 = np.zeros(P2.p if per else 1, np.float32)
        hat = np.zeros(P1.n, np.float32)
        mu = np.zeros(1, np.float32); ve = np.zeros(1, np.float32); h2 = np.zeros(1, np.float32)
        check(_lib.lib().bwgr_bayes2(P1._h, P2._h, MODELS[base], _fp(y), float(it), float(bi), float(pi), float(df), float(R2),
                                      C.c_uint64(_seed(seed)), int(rng_mode), _fp(mu), _fp(b1), _fp(d1), _fp(vb1), _fp(b2), _fp(d2),
                                      _fp(vb2), _fp(ve), _fp(hat), _fp(h2)))
        out = {"hat": hat, "mu": float(mu[0]), "b1": b1, "b2": b2, "vb1": vb1 if per else float(vb1[0]),
               "vb2": vb2 if per else float(vb2[0]), "ve": float(ve[0]), "h2": float(h2[0])}
        if base == "BayesB":   # list order of :1146-1149
            out = {"mu": out["mu"], "b1": b1, "d1": d1, "vb1": vb1, "b2": b2, "d2": d2, "vb2": vb2, "ve": out["ve"], "hat": hat, "h2": out["h2"]}
        return out
    finally:
        if own2:
            P2.close()
        if own1:
            P1.close()


def BayesA2(y, X1, X2, it=1500, bi=500, df=5, R2=0.5, *, seed=None, rng_mode=0, **kw):
    return _fused2("BayesA", y, X1, X2, it, bi, 0.0, df, R2, seed, rng_mode, **kw)


def BayesB2(y, X1, X2, it=1500, bi=500, pi=0.95, df=5, R2=0.5, *, seed=None, rng_mode=0, **kw):
    return _fused2("BayesB", y, X1, X2, it, bi, pi, df, R2, seed, rng_mode, **kw)


def BayesRR2(y, X1, X2, it=1500, bi=500, df=5, R2=0.5, *, seed=None, rng_mode=0, **kw):
    return _fused2("BayesRR", y, X1, X2, it, bi, 0.0, df, R2, seed, rng_mode, **kw)


def sample_rows(seed, it, n, k, rp=False):
    """sort(sample(n, k, rp)) - 1 under the RNG contract (host only)."""
    out = np.zeros(int(k), np.int32)
    check(_lib.lib().bwgr_sample_rows(C.c_uint64(_seed(seed)), C.c_uint32(int(it)), int(n), int(k), int(bool(rp)),
                                       out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out


def _cor_last(dta):
    """cor(dta, use = 'p')[-m, m]: Pearson correlation of every column with the last one over pairwise complete rows."""
    m = dta.shape[1]
    out = np.full(m - 1, np.nan)
    for i in range(m - 1):
        ok = np.isfinite(dta[:, i]) & np.isfinite(dta[:, m - 1])
        if ok.sum() > 1:
            out[i] = np.corrcoef(dta[ok, i], dta[ok, m - 1])[0, 1]
    return out


_CV_FITS = ("BayesA", "BayesB", "BayesC", "BayesL", "BayesRR", "BayesCpi", "BayesDpi")          # f1 .. f7, R/cv.R:124-130
_CV_NAMES = ("BayesA", "BayesB", "BayesC", "BayesL", "BayesCpi", "BayesDpi", "BayesRR")         # column labels, R/cv.R:132-133


def mcmcCV(y, gen, k=5, n=5, it=1500, bi=500, pi=0.95, df=5, R2=0.5, avg=True, llo=None, tbv=None, ReturnGebv=False, *,
           seed=None, **panel_kw):
    """mcmcCV(), R/cv.R:113-216: n random k-fold (or leave-level-out) cross-validation cycles of the seven fused samplers;
    every cycle stages its training rows once and runs the seven chains on that panel, side by side (fit_many).  Returns the predictive
    correlations like the reference (`avg`: one vector sorted decreasingly, else one row per cycle), rounded to 4
    digits, or list(cv, hat, beta) with ReturnGebv.  As in the reference, column i holds the predictions of fit f_i =
    (A, B, C, L, RR, Cpi, Dpi) but is *labelled* (A, B, C, L, Cpi, Dpi, RR) (R/cv.R:124-133).  Fold rows come from the
    RNG contract (iteration word = cycle number) instead of set.seed(cycle); sample()."""
    y = np.asarray(y, np.float64); gen = np.asarray(gen)
    N, p = gen.shape
    base = _seed(seed)
    if llo is None:
        Nk = int(round(N / k))
        cycles = [sample_rows(base, c + 1, N, Nk) for c in range(int(n))]
    else:
        llo = np.asarray(llo).astype(str)
        cycles = [np.nonzero(llo == lev)[0] for lev in dict.fromkeys(llo.tolist())]
    obs = y if tbv is None else np.asarray(tbv, np.float64)
    Ms, Bs = [], []
    for c, w in enumerate(cycles):
        keep = np.ones(N, bool); keep[w] = False
        P = Panel(np.ascontiguousarray(gen[keep]), **panel_kw)
        try:
            B = np.zeros((p, 7))
            ytr = y[keep].astype(np.float32)
            jobs = [dict(model=model, y=ytr, it=it, bi=bi, df=df, R2=R2, seed=(base + 1000003 * (c + 1) + i) & 0x7FFFFFFFFFFFFFFF,
                         **({"pi": pi} if model in ("BayesB", "BayesC") else {})) for i, model in enumerate(_CV_FITS)]
            for i, fit in enumerate(fit_many(P, jobs)):   # the seven chains side by side on the fold's training panel
                B[:, i] = fit["b"]
        finally:
            P.close()
        M = np.empty((len(w), 8))
        M[:, :7] = np.asarray(gen[w], np.float64) @ B
        M[:, 7] = obs[w]
        Ms.append(M); Bs.append(B)
    if avg:
        pa = _cor_last(np.vstack(Ms))
        order = np.argsort(-pa, kind="stable")
        cv = {_CV_NAMES[i]: round(float(pa[i]), 4) for i in order}
    else:
        cv = {"CV_%d" % (c + 1): {_CV_NAMES[i]: round(float(v), 4) for i, v in enumerate(_cor_last(M))} for c, M in enumerate(Ms)}
    if not ReturnGebv:
        return cv
    beta = sum(Bs) / len(Bs)
    hat = np.asarray(gen, np.float64) @ beta + np.nanmean(y)
    return {"cv": cv, "hat": hat, "beta": beta}


def KMUP(X, b, d, xx, e, L, Ve, pi, *, seed=None, it=0, rng_mode=0, **panel_kw):
    """One Gibbs sweep; returns list(b=, d=, e=) like src/Rcpp20260726ai.cpp:37.  Inputs are not modified."""
    P, own = _as_panel(X, **panel_kw)
    try:
        b = np.array(b, np.float32); d = np.array(d, np.float32); e = np.array(e, np.float32)
        xx = np.ascontiguousarray(xx, np.float32); L = np.ascontiguousarray(L, np.float32)
        assert b.size == P.p and d.size == P.p and xx.size == P.p and L.size == P.p and e.size == P.n
        check(_lib.lib().bwgr_kmup(P._h, _fp(b), _fp(d), _fp(xx), _fp(e), _fp(L), float(Ve), float(pi),
                                    C.c_uint64(_seed(seed)), C.c_uint32(int(it)), int(rng_mode)))
        return {"b": b, "d": d, "e": e}
    finally:
        if own:
            P.close()


def KMUP2(X, Use, b, d, xx, E, L, Ve, pi, *, seed=None, it=0, rng_mode=0, **panel_kw):
    """One Gibbs sweep on the row subsample Use (0-based, as wgr passes it: sort(sample(n, n*bag, rp)) - 1, R/wgr.R:68);
    returns list(b=, d=, e=) like src/Rcpp20260726ai.cpp:76 (e: the subsample's residuals).  Inputs are not modified."""
    P, own = _as_panel(X, **panel_kw)
    try:
        use = np.ascontiguousarray(np.asarray(Use), np.int32)
        b = np.array(b, np.float32); d = np.array(d, np.float32); E = np.ascontiguousarray(E, np.float32)
        xx = np.ascontiguousarray(xx, np.float32); L = np.ascontiguousarray(L, np.float32)
        assert b.size == P.p and d.size == P.p and xx.size == P.p and L.size == P.p and E.size == P.n
        e = np.zeros(use.size, np.float32)
        check(_lib.lib().bwgr_kmup2(P._h, use.ctypes.data_as(C.POINTER(C.c_int)), int(use.size), _fp(b), _fp(d), _fp(xx), _fp(E),
                                     _fp(e), _fp(L), float(Ve), float(pi), C.c_uint64(_seed(seed)), C.c_uint32(int(it)), int(rng_mode)))
        return {"b": b, "d": d, "e": e}
    finally:
        if own:
            P.close()


def wgr(y, X, it=1500, bi=500, th=1, bag=1, rp=False, iv=False, de=False, pi=0, df=5, R2=0.5, eigK=None, VarK=0.95,
        verb=False, *, seed=None, rng_mode=0, **panel_kw):
    """wgr(), R/wgr.R:2-169, device-resident.  eigK = {"values": ..., "vectors": ...} (R's eigen(K)) adds the
    polygenic kernel term; bag != 1 sweeps KMUP2 on sort(sample(n, n*bag, rp)) rows each iteration.  The two together
    are refused: the reference itself indexes out of bounds there (R/wgr.R:73-79).
    Returns wgr's list: mu, b, Vb, d, Ve, hat[, u, Vk], cxx (R/wgr.R:155-167)."""
    if bag != 1 and eigK is not None:
        raise NotImplementedError("wgr(bag != 1, eigK=...): undefined in the reference (a subsampled residual is indexed with "
                                  "full-length row ids, R/wgr.R:73-79)")
    y = np.asarray(y, np.float64)
    U0 = V = None
    if eigK is not None:                       # R/wgr.R:23-27
        Vall = np.asarray(eigK["values"], np.float64)
        pk = int(np.argmax((np.cumsum(Vall) / Vall.size) > VarK)) + 1
        U0 = np.asfortranarray(np.asarray(eigK["vectors"], np.float64)[:, :pk])
        V = np.ascontiguousarray(Vall[:pk])
    gen0 = None
    keep = None
    if np.isnan(y).any():   # R/wgr.R:34-39: drop rows with missing y (hat is still returned for every row of gen0)
        if isinstance(X, Panel) or hasattr(X, "data_ptr"):
            raise ValueError("missing y with a pre-staged X: drop the rows before staging X")
        keep = ~np.isnan(y)
        gen0 = np.asarray(X)
        y = y[keep]
    if not isinstance(X, Panel) and not hasattr(X, "data_ptr"):
        X = np.asarray(X)
        if np.issubdtype(X.dtype, np.floating) and np.isnan(X).any():   # R/wgr.R:12-18 mean imputation
            X = X.astype(np.float64, copy=True)
            cm = np.nanmean(X, axis=0); cm[np.isnan(cm)] = 0.0
            idx = np.where(np.isnan(X)); X[idx] = cm[idx[1]]
            if gen0 is not None:
                gen0 = X
        if keep is not None:
            X = X[keep]
    U = None
    if U0 is not None:
        U = np.asfortranarray(U0[keep] if keep is not None else U0)
    P, own = _as_panel(X, **panel_kw)
    try:
        n, p = P.n, P.p
        assert y.size == n, "length(y) must equal nrow(X)"
        per = bool(iv or de)
        b = np.zeros(p); d = np.zeros(p); Vb = np.zeros(p if per else 1); hat = np.zeros(n); u = np.zeros(n)
        mu = C.c_double(); Ve = C.c_double(); cxx = C.c_double(); Vk = C.c_double()
        yc = np.ascontiguousarray(y, np.float64)
        if U is None and bag == 1:
            check(_lib.lib().bwgr_wgr(P._h, _dp(yc), int(it), int(bi), int(th), int(bool(iv)), int(bool(de)), float(pi),
                                       float(df), float(R2), C.c_uint64(_seed(seed)), int(rng_mode), C.byref(mu), _dp(b), _dp(Vb),
                                       _dp(d), C.byref(Ve), _dp(hat), C.byref(cxx)))
        else:
            assert U is None or U.shape[0] == n
            check(_lib.lib().bwgr_wgr_ex(P._h, _dp(yc), int(it), int(bi), int(th), int(bool(iv)), int(bool(de)), float(pi),
                                          float(df), float(R2), C.c_uint64(_seed(seed)), int(rng_mode),
                                          _dp(U) if U is not None else None, _dp(V) if U is not None else None,
                                          C.c_int64(U.shape[1] if U is not None else 0), float(bag), int(bool(rp)), C.byref(mu),
                                          _dp(b), _dp(Vb), _dp(d), C.byref(Ve), _dp(hat), C.byref(cxx), _dp(u), C.byref(Vk)))
        if keep is not None:
            # HAT = B0 + gen0 %*% B (+ U0 %*% H) over ALL rows of gen0 (R/wgr.R:146-152): rows with missing y are predicted.
            # R-level post-processing in the reference too; the rows that were swept keep the device's values.
            full = np.empty(keep.size); full[keep] = hat
            miss = ~keep
            full[miss] = mu.value + np.asarray(gen0, np.float64)[miss] @ b
            if U is not None:
                Hk = np.linalg.lstsq(U, u, rcond=None)[0]      # u = U %*% H on the kept rows; recover H for the others
                ufull = U0 @ Hk
                full[miss] += ufull[miss]
                u = ufull
            hat = full
        out = {"mu": mu.value, "b": b, "Vb": Vb if per else float(Vb[0]), "d": d, "Ve": Ve.value, "hat": hat}
        if U is not None:
            out["u"] = u; out["Vk"] = Vk.value
        out["cxx"] = cxx.value
        return out
    finally:
        if own:
            P.close()


def debug_variates(seed, kind, marker0, count, it=0, purpose=0, nu=0.0, device=0):
    out = np.empty(count, np.float64)
    kinds = {"normal": 0, "uniform": 1, "chisq": 2}
    check(_lib.lib().bwgr_debug_variates(device, C.c_uint64(seed), kinds[kind], float(nu), C.c_uint32(marker0), C.c_uint32(it),
                                          C.c_uint32(purpose), int(count), _dp(out)))
    return out
