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
wl = {"c2": (5000, 50000, "BayesA", 0.0), "c4s": (10000, 100000, "BayesB", 0.99), "c4d": (10000, 100000, "BayesCpi", 0.5), "c4m": (10000, 100000, "BayesB", 0.95), "c5s": (50000, 100000, "BayesCpi", 0.5)}
names2 = {0: "streamer: loop top + tile issue", 9: "streamer: tile commit (vmcnt(0) + LDS stores)", 10: "streamer: delta poll (early request or loop)", 11: "streamer: wave max + LDS atomic", 1: "streamer: barrier after the poll", 2: "streamer: delta digits", 3: "streamer: update MFMA (wave 0)", 4: "streamer: update barrier", 5: "streamer: e update + max", 6: "streamer: e digits", 7: "streamer: dots (barrier, MFMA, barrier)", 8: "streamer: q recombine + store",
          16: "sequencer: top barrier", 21: "sequencer: lane constants", 22: "sequencer: recurrence rounds", 17: "sequencer: outputs + delta store", 18: "sequencer: wait at barrier A (helpers, q_{b+1})", 19: "sequencer: post (state, r0 of next block)", 20: "sequencer: post tail"}
names = ["top-barrier", "dot", "combine+exchange", "wait for prefetch waves", "outputs+update", "spec matvec", "recurrence (wave 0)", "-", "wave1: prefetch until tile+stage stored", "wave1: residual vmcnt(0)", "wave1: t(gram loads landed)", "wave1: t(+stage landed)"]
names2w = dict(names2)
if os.environ.get("BWGR_WINV", "1") != "0":   # affine sweeps run k_sweep2w: its sequencer stamps wave 4 (sweep2w.hip.h)
    names2w.update({16: "sequencer w8: wait at B0", 17: "-", 18: "sequencer w8: wait at B2 (waves 0-3: distance-1 term + rhs)", 19: "sequencer w8: collect q_{c+1}, request q_{c+2}", 20: "sequencer w8: wait at B3 (waves 4-7: product)", 21: "sequencer w8: steps, granules, digits", 22: "-",
                   24: "sequencer w0: (loop edge)", 25: "sequencer w0: wait at B0", 26: "sequencer w0: distance-1 term + rhs", 27: "sequencer w0: wait at B2 + plane / constant requests", 28: "sequencer w0: wait at B3", 29: "sequencer w0: wait for the distance-2/3 planes (vmcnt)", 30: "-", 31: "sequencer w0: distance-2/3 terms + q (xr_early)"})
for key in sys.argv[1:] or ["c2", "c4s"]:
    n, p, model, pi = wl[key]
    names2 = names2 if pi else names2w
    X = synth.genotypes(n, p); y = synth.scale_phenotype(synth.phenotype(X, n))
    P = bwgr_amd.Panel(X, n=n); del X
    ch = bwgr_amd.Chain(P, model, y, it=4, bi=0, pi=pi, seed=1)
    ch.run(1); ch.sync()
    out = (C.c_ulonglong * 256)(); _lib.lib().bwgr_debug_stamps(P._h, out)
    ch.run(3); ch.sync()
    _lib.lib().bwgr_debug_stamps(P._h, out)
    v = np.array(list(out), float); nblk = 3 * ((p + P.block - 1) // P.block)
    print(key, model, "n=%d p=%d K=%d m=%d" % (n, p, P.nwg, P.block), "cycles/block (s_memtime):")
    use2 = os.environ.get('BWGR_SWEEP', '2') != '1'
    if use2:
        for k in sorted(names2): print("   %-55s %9.0f" % (names2[k], v[k] / nblk))
        print("   streamer 0 total %.0f   sequencer total %.0f" % (sum(v[k] for k in names2 if k < 16) / nblk, sum(v[k] for k in names2 if 16 <= k < 24) / nblk))
        lag = int(os.environ.get('BWGR_LAG', '4')); hw = v[32:48] / (nblk - 3 * lag) / 100.0
        if pi:   # selection models: lag-3 pipeline with the q feeder (wall clock, 100 MHz)
            print("   means (us): delta_i stored -> seen by streamer 0 %.2f -> streamer 0 stores q_{i+lag} %.2f -> feeder puts the sum %.2f -> sequencer has it %.2f"
                  % (hw[1] - hw[0], hw[2] - hw[0], hw[3] - hw[0], hw[10] - hw[0]))
            one = (C.c_ulonglong * 256)(); ch2 = bwgr_amd.Chain(P, model, y, it=2, bi=0, pi=pi, seed=2)
            ch2.run(1); ch2.sync(); _lib.lib().bwgr_debug_stamps(P._h, one); ch2.run(1); ch2.sync(); _lib.lib().bwgr_debug_stamps(P._h, one); ch2.close()
            oo = [int(x) for x in list(one)]; o = oo[32:48]; t0 = o[7]
            print("   one sweep, around block 100 (us after the sequencer stores delta_100): streamer 0 sees delta_100 %.2f, stores q_{100+lag} %.2f, sees delta_101 %.2f; "
                  "feeder puts the sum of q_{100+lag} %.2f; sequencer wave 7 has it %.2f; sequencer stores delta_101 %.2f, delta_102 %.2f, delta_103 %.2f"
                  % tuple((x - t0) / 100.0 for x in (o[4], o[6], o[5], o[13], o[14], o[9], o[11], o[12])))
            mm = lambda k: ((~oo[k]) & 0xFFFFFFFFFFFFFFFF)
            print("   over all streamers (us after delta_100 stored): delta_100 seen first %.2f / last %.2f;  q_{100+lag} stored first %.2f / last %.2f"
                  % ((mm(58) - t0) / 100.0, (oo[59] - t0) / 100.0, (mm(56) - t0) / 100.0, (oo[57] - t0) / 100.0))
            print("   sequencer helper phase for block 100+lag (us after it starts, i.e. after the barrier of block 98+lag): waves 1-5: Gram/constant stores done %.2f, next "
                  "loads issued %.2f;  wave 7: sum of q polled %.2f; wave 6: cross terms done %.2f; wave 7: state of block 98+lag stored %.2f;  barrier of block 99+lag at %.2f)" % tuple((x - oo[48]) / 100.0 for x in (oo[49], oo[50], oo[52], oo[53], oo[54], oo[55])))
            nn = nblk - 3 * lag
            seen = [(v[64 + w] / nn - v[32] / nn) / 100.0 for w in range(P.nwg)]   # mean (delta_i seen by w) - (delta_i stored)
            work = [(v[128 + w] - v[64 + w]) / nn / 100.0 for w in range(P.nwg)]    # mean (q_{i+lag} stored) - (delta_i seen)
            print("   per streamer, mean us: delta_i stored -> seen:", " ".join("%.2f" % x for x in seen))
            print("   per streamer, mean us: seen -> q_{i+lag} stored:", " ".join("%.2f" % x for x in work))
            print("   streamer 0, wave 4, block 100 (us after the update barrier): tile commit done %.2f, next tile's loads issued %.2f (behind the e-update barrier, passed at %.2f)"
                  % ((o[8] & 0xFFFFFFFF) / 100.0, (o[8] >> 32) / 100.0, o[15] / 100.0))
    else:
        for nm, x in zip(names, v): print("   %-50s %9.0f ticks/block" % (nm, x / nblk))
    ms, nl = ch.sweep_ms(); print("   sweep ms", ms)
    ch.close(); P.close()
