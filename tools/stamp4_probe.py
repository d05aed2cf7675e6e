"""Diagnostic only: builds libbwgr_hip_stamps.so (-DBWGR_STAMPS=4) and prints where k_sweep4's roles spend their cycles per quad
(s_memtime ticks, lane 0 of each wave).  Never quote this build's run time: a stamp drains the wave's scalar / LDS counter."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bwgr_amd import build as B
so = os.path.join(ROOT, "gpurun_out", "libbwgr_hip_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DBWGR_STAMPS=4", "-o", so] + B.SOURCES)
B.LIB = so
import numpy as np, torch
import bwgr_amd
from bwgr_amd import synth, _lib
n, p = 10000, int(os.environ.get("AB_P", "200000"))
X = synth.genotypes(n, p); y = synth.scale_phenotype(synth.phenotype(X, n))
P = bwgr_amd.Panel(X, n=n); del X
ch = bwgr_amd.Chain(P, "BayesB", y, it=4, bi=0, pi=0.99, seed=1)
ch.run(1); ch.sync()
out = (C.c_ulonglong * 256)(); _lib.lib().bwgr_debug_stamps(P._h, out)
ch.run(3); ch.sync()
_lib.lib().bwgr_debug_stamps(P._h, out)
v = np.array(list(out), float); nquad = 3 * ((p + 4 * P.block - 1) // (4 * P.block))
names_u = ["work of the previous phase (update MFMA)", "fold + range check", "digits of e + wait for the tile", "barrier", "tile / drej DMA issue", "list prefetch + request", "-", "-"]
names_d = ["work of the previous phase (dots + atomics)", "wait for the tile (vmcnt)", "digits of drej", "barrier", "tile / drej DMA issue", "-", "-", "-"]
names_q = ["pass, outputs, list words of the previous quad", "HAND-OFF (token stored -> wait left; inside the token wait)", "q complete (poll)", "far rows applied + near entries so far", "next constants requested + candidates announced", "token wait (+ entries as they appear)", "rounds (incl. the wait for the announced rows)", "(included markers)"]
for title, base, names in [("streamer 0 update wave", 0, names_u), ("streamer 0 dots wave", 8, names_d)] + [("sequencer wave %d" % w, 64 + 8 * w, names_q) for w in range(8)]:
    tot = sum(v[base:base + 7]) / nquad
    print("  %s: %.0f ticks per quad" % (title, tot))
    for k, nm in enumerate(names):
        if nm != "-": print("     %-52s %9.1f" % (nm, v[base + k] / nquad))
for w in range(8):
    b = 128 + 8 * w
    print("  wave %d wait loop: %.1f polls per quad, %.2f entries applied in it, %.0f ticks inside apply_near (%.0f per entry); unannounced inclusions %.3f per quad" % (w, v[b + 2] / nquad, v[b + 1] / nquad, v[b] / nquad, v[b] / max(v[b + 1], 1), v[b + 3] / nquad))
print("  per quad, lanes with |z| / hr at announce time >= 1 / 0.9 / 0.8 / 0.6: %.1f / %.1f / %.1f / %.1f" % tuple(v[192:196] / nquad))
print("  per quad, unannounced inclusions by |z| / hr at announce time: >= 1 (third candidate) %.2f, 0.9-1 %.2f, 0.8-0.9 %.2f, 0.6-0.8 %.2f" % tuple(v[196:200] / nquad))
print("  slow rows (DMA + wait on the spot): %.2f per quad over all waves, %.0f ticks each" % (v[201] / nquad, v[200] / max(v[201], 1)))
ms, nl = ch.sweep_ms(); print("   sweep ms %.3f  (%.2f us per block, stamped build)" % (ms, 1e3 * ms / ((p + P.block - 1) // P.block)))
ch.close(); P.close()
