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
names2 = {0: "streamer: loop top + tile issue", 9: "streamer: tile commit (vmcnt(0) + LDS stores)", 1: "streamer: wait delta (+ tile issue)", 2: "streamer: delta digits", 3: "streamer: update MFMA (wave 0)", 4: "streamer: update barrier", 5: "streamer: e update + max", 6: "streamer: e digits", 7: "streamer: dots (barrier, MFMA, barrier)", 8: "streamer: q recombine + store",
          16: "sequencer: top barrier", 21: "sequencer: lane constants", 22: "sequencer: recurrence rounds", 17: "sequencer: outputs + delta store", 18: "sequencer: wait at barrier A (helpers, q_{b+1})", 19: "sequencer: post (state, r0 of next block)", 20: "sequencer: post tail"}
names = ["top-barrier", "dot", "combine+exchange", "wait for prefetch waves", "outputs+update", "spec matvec", "recurrence (wave 0)", "-", "wave1: prefetch until tile+stage stored", "wave1: residual vmcnt(0)", "wave1: t(gram loads landed)", "wave1: t(+stage landed)"]
for key in sys.argv[1:] or ["c2", "c4s"]:
    n, p, model, pi = wl[key]
    X = synth.genotypes(n, p); y = synth.scale_phenotype(synth.phenotype(X, n))
    P = bwgr_amd.Panel(X, n=n); del X
    ch = bwgr_amd.Chain(P, model, y, it=4, bi=0, pi=pi, seed=1)
    ch.run(1); ch.sync()
    out = (C.c_ulonglong * 48)(); _lib.lib().bwgr_debug_stamps(P._h, out)
    ch.run(3); ch.sync()
    _lib.lib().bwgr_debug_stamps(P._h, out)
    v = np.array(list(out), float); nblk = 3 * ((p + P.block - 1) // P.block)
    print(key, model, "n=%d p=%d K=%d m=%d" % (n, p, P.nwg, P.block), "cycles/block (s_memtime):")
    use2 = os.environ.get('BWGR_SWEEP', '2') != '1'
    if use2:
        for k in sorted(names2): print("   %-55s %9.0f" % (names2[k], v[k] / nblk))
        print("   streamer 0 total %.0f   sequencer total %.0f" % (v[:16].sum() / nblk, v[16:32].sum() / nblk))
        w = v[32:36]
        print("   wall clock: delta stored -> seen by streamer 0: %.2f us;  q stored by streamer 0 -> gathered from all: %.2f us" % ((w[1] - w[0]) / (nblk - 6) / 100.0, (w[3] - w[2]) / (nblk - 6) / 100.0))
    else:
        for nm, x in zip(names, v): print("   %-50s %9.0f ticks/block" % (nm, x / nblk))
    hw = v[32:48] / (nblk - 3) / 100.0
    print("   sequencer helper wave 1 (us per block, wall clock): wait for prefetch regs %.2f, stores to LDS %.2f, issue next prefetch %.2f, gather q %.2f" % (hw[5] - hw[4], hw[6] - hw[5], hw[7] - hw[6], hw[9] - hw[7]))
    ms, nl = ch.sweep_ms(); print("   sweep ms", ms)
    ch.close(); P.close()
