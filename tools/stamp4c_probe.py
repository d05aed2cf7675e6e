"""Diagnostic only: builds libbwgr_hip_stamps.so (-DBWGR_STAMPS=4) and prints where k_sweep4's CHAIN WAVE spends its cycles per
quad (s_memtime ticks).  Never quote this build's run time."""
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
names = ["loop top (4 tasks): decide x3 + adopt x3 of the previous round", "adopt T+3", "decide T", "(inside adopt) waiting for the record + reading it", "(decide) markers nobody had announced (row set fetched on the spot)", "(look) row sets the chain wave requested itself", "(inside decide) waiting for a row set to land", "(included markers)"]
print("chain wave, per quad:")
for k, nm in enumerate(names): print("     %-70s %9.1f" % (nm, v[64 + k] / nquad))
print("     inclusion: b1 / sliver / publish / list words %.0f, apply + look %.0f (ticks per quad)" % (v[72] / nquad, v[73] / nquad))
hn = ["loop turn (copies, next constants requested)", "first 32 rows' registers copied", "outputs of the task two back", "q complete", "first rows applied", "until the record is first published (further entries, waiting to be due)", "following until adopted (+ announces, next task's rows requested)", "(tasks ahead of the chain at first publish, summed)"]
ntask = 8.0 * nquad
for w in range(1, 7):
    b = 80 + 8 * w
    print("  helper wave %d, ticks per task of its own (%.2f tasks per quad):" % (w, 8.0 / 6.0), "  ".join("%s=%.0f" % (k, v[b + k] * 6.0 / ntask) for k in range(8)))
for k, nm in enumerate(hn): print("       %d: %s" % (k, nm))
ms, nl = ch.sweep_ms(); print("   sweep ms %.3f  (%.2f us per block, stamped build)" % (ms, 1e3 * ms / ((p + P.block - 1) // P.block)))
ch.close(); P.close()
