"""Diagnostic only: builds libbwgr_hip_stamps.so (-DBWGR_STAMPS) and prints where k_sweep3's roles spend their cycles per
block (s_memtime ticks).  Never quote this build's run time.  (Since the sequencer's roles reach the block barrier inside their
own branches, the helper waves' second slot -- "poll + convert", "commit + request", ... -- includes their wait at that barrier;
wave 0's slots are unchanged: constants + r0, rounds incl. the wait for the distance-1 / 2 rows, outputs, barrier.)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bwgr_amd import build as B
so = os.path.join(ROOT, "gpurun_out", "libbwgr_hip_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
LITE = os.environ.get("STAMPS_LITE", "0")   # 0: every stamp; 1: busy / waiting per role; 2: only the sequencer's wave 0, busy / waiting
EXP = ["-DBWGR_EXPERIMENTS"] if os.environ.get("STAMP_EXP") else []   # (with the BWGR_DBG3 switches: roles off, sequencer alone, ...)
subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + EXP + ["-DBWGR_STAMPS=%d" % (1 + int(LITE)), "-o", so] + B.SOURCES)
B.LIB = so
import numpy as np, torch
import bwgr_amd
from bwgr_amd import synth, _lib
wl = {"c4x": (10000, int(os.environ.get("AB_P", "500000")), "BayesB", 0.99), "c4s": (10000, 200000, "BayesB", 0.99), "c4b": (10000, 200000, "BayesB", 0.95), "c4d": (10000, 100000, "BayesCpi", 0.5)}
rows = [("streamer 0, update wave", 0, ["loop top", "fold the list of block b-D", "digits of e and drej", "barrier", "requests (drej, list) after the issue", "update MFMA + recombine", "abort check + tile commit", "next tile issue"]),
        ("streamer 0, first dots wave", 8, ["loop top", "-", "-", "barrier (incl. waiting for the update waves)", "requests (drej, list) after the issue", "dots MFMA + recombine + atomics", "abort check + tile commit", "next tile issue"]),
        ("sequencer wave 0", 16, ["loop top", "constants + r0", "waiting for the distance-1 / 2 rows after the last round", "outputs", "barrier", "rounds", "-", "-"]),
        ("sequencer wave 1 (q poll)", 24, ["loop top", "wait at the block barrier (rest of the phase)", "poll + convert (until the next block's slab dots are complete)", "-", "barrier", "-", "-", "-"]),
        ("sequencer wave 2 (staging)", 32, ["loop top", "wait at the block barrier (rest of the phase)", "commit + request", "-", "barrier", "-", "-", "-"]),
        ("sequencer wave 5 (far field)", 40, ["loop top", "(rest)", "-", "-", "barrier", "wait for last phase's rows", "consume", "issue"]),
        ("sequencer wave 3 (staging)", 48, ["loop top", "wait at the block barrier (rest of the phase)", "commit + request", "-", "barrier", "-", "-", "-"]),
        ("sequencer wave 6 (far field)", 56, ["loop top", "wait at the block barrier (rest of the phase)", "consume + issue", "-", "barrier", "-", "-", "-"]),
        ("sequencer wave 7 (state, sums, lists)", 64, ["loop top", "wait at the block barrier (rest of the phase)", "finish_block", "-", "barrier", "-", "-", "-"])]
for key in sys.argv[1:] or ["c4s"]:
    n, p, model, pi = wl[key]
    X = synth.genotypes(n, p); y = synth.scale_phenotype(synth.phenotype(X, n))
    P = bwgr_amd.Panel(X, n=n); del X
    print(key, model, "n=%d p=%d" % (n, p), P.pipeline(True))
    ch = bwgr_amd.Chain(P, model, y, it=4, bi=0, pi=pi, seed=1)
    ch.run(1); ch.sync()
    out = (C.c_ulonglong * 256)(); _lib.lib().bwgr_debug_stamps(P._h, out)
    ch.run(3); ch.sync()
    _lib.lib().bwgr_debug_stamps(P._h, out)
    v = np.array(list(out), float); nblk = 3 * ((p + P.block - 1) // P.block)
    for title, base, names in rows:
        tot = sum(v[base:base + 8]) / nblk
        print("  %s: %.0f ticks per block" % (title, tot))
        for k, nm in enumerate(names):
            if nm != "-": print("     %-52s %9.0f" % (nm, v[base + k] / nblk))
    ms, nl = ch.sweep_ms(); print("   sweep ms %.3f  (%.2f us per block, stamped build)" % (ms, 1e3 * ms / (nblk / 3)))
    ch.close(); P.close()
