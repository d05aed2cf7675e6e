"""Diagnostic only: builds libbwgr_hip_stamps.so (-DBWGR_STAMPS) and prints the share of workgroup 0's cycles spent in
each phase of k_sweep.  Never quote this build's run time (its stamps forbid overlaps the real kernel has)."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bwgr_amd import build as B
so = os.path.join(ROOT, "gpurun_out", "libbwgr_hip_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DBWGR_STAMPS", "-o", so] + B.SOURCES)
B.LIB = so
import numpy as np, torch
import bwgr_amd
from bwgr_amd import synth, _lib
wl = {"c2": (5000, 50000, "BayesA", 0.0), "c4s": (10000, 100000, "BayesB", 0.99)}
names2 = ["streamer: tile commit/issue", "streamer: wait delta", "streamer: slab update", "-", "streamer: dots+publish", "streamer: loop top", "sequencer: top barrier", "sequencer: recurrence (wave 0)", "sequencer: wait at barrier A (helpers + q_{b+1})", "sequencer: outputs, Gram->LDS, gather", "sequencer: cross/spec matvecs", "-"]
names = ["top-barrier", "dot", "combine+exchange", "wait for prefetch waves", "outputs+update", "spec matvec", "recurrence (wave 0)", "-", "wave1: prefetch until tile+stage stored", "wave1: residual vmcnt(0)", "wave1: t(gram loads landed)", "wave1: t(+stage landed)"]
for key in sys.argv[1:] or ["c2", "c4s"]:
    n, p, model, pi = wl[key]
    X = synth.genotypes(n, p); y = synth.scale_phenotype(synth.phenotype(X, n))
    P = bwgr_amd.Panel(X, n=n); del X
    ch = bwgr_amd.Chain(P, model, y, it=4, bi=0, pi=pi, seed=1)
    ch.run(1); ch.sync()
    out = (C.c_ulonglong * 12)(); _lib.lib().bwgr_debug_stamps(P._h, out)
    ch.run(3); ch.sync()
    _lib.lib().bwgr_debug_stamps(P._h, out)
    v = np.array(list(out)[:12], float); nblk = 3 * ((p + P.block - 1) // P.block)
    print(key, model, "n=%d p=%d K=%d m=%d" % (n, p, P.nwg, P.block), "ticks/block (100MHz? s_memtime):")
    import os
    use2 = os.environ.get('BWGR_SWEEP', '2') != '1'
    for nm, x in zip(names2 if use2 else names, v):
        print("   %-50s %9.0f ticks/block" % (nm, x / nblk))
    ms, nl = ch.sweep_ms(); print("   sweep ms", ms)
    ch.close(); P.close()
