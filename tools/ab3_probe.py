"""Diagnostic: sweep time of k_sweep3 under the BWGR_DBG3 experiment switches (some of them break the chain on purpose:
timing only; they are compiled into a library of their own, built here with -DBWGR_EXPERIMENTS) and under BWGR_D3 / BWGR_R3.  Usage: ab3_probe.py "VAR=val,VAR=val" ... (each argument one configuration)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subprocess
from bwgr_amd import build as B
if not os.environ.get("AB_PREBUILT"):   # (tools/variants.py hands over an experiment build through BWGR_LIB)
    so = os.path.join(ROOT, "gpurun_out", "libbwgr_hip_exp.so")     # the experiment switches live in a build of their own (-DBWGR_EXPERIMENTS)
    os.makedirs(os.path.dirname(so), exist_ok=True)
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in B.DEPS):
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DBWGR_EXPERIMENTS", "-o", so] + B.SOURCES)
    B.LIB = so
import torch, bwgr_amd
from bwgr_amd import synth
n, p = 10000, int(os.environ.get("AB_P", "200000"))
model, pi = os.environ.get("AB_MODEL", "BayesB"), float(os.environ.get("AB_PI", "0.99"))
X = synth.genotypes(n, p); y = synth.scale_phenotype(synth.phenotype(X, n))
for cfg in sys.argv[1:] or [""]:
    for kv in cfg.split(","):
        if "=" in kv:
            k, v = kv.split("="); os.environ[k] = v
    P = bwgr_amd.Panel(X, n=n)
    ch = bwgr_amd.Chain(P, model, y, it=6, bi=0, pi=pi, seed=1)
    try:
        ch.run(2); ch.sync()
    except Exception as ex:
        pass
    try:
        ch.sweep_ms()
        ch.run(3)
        try:
            ch.sync()
        except Exception:
            pass
        ms, nl = ch.sweep_ms()
        nb = (p + P.block - 1) // P.block
        st = None
        try:
            st = ch.state()
        except Exception:
            pass
        sel = model in ("BayesB", "BayesC", "BayesCpi", "BayesDpi")
        try:
            nredo = ch.redo_count()
        except Exception:
            nredo = "n/a"
        try:
            import ctypes as C
            from bwgr_amd import _lib
            out = (C.c_ulonglong * 256)(); _lib.lib().bwgr_debug_stamps(P._h, out)
            ev = " ev[poll incomplete %d spins %d | folds %d late %d]" % (out[200], out[201], out[203], out[202])
        except Exception as ex:
            ev = " ev[%r]" % (ex,)
        print("%-40s sweep %8.3f ms  %6.3f us/block  mean_d %s  %s redo %s ve %s" % (cfg, ms, 1e3 * ms / nb, "%.4f" % st["d"].mean() if st else "n/a", P.pipeline(sel), nredo, st["ve"] if st else "n/a") + ev, flush=True)
    finally:
        try:
            ch.close(); P.close()
        except Exception:
            pass
    for kv in cfg.split(","):
        if "=" in kv:
            os.environ.pop(kv.split("=")[0], None)
