"""Diagnostic: one chain swept in one launch per iteration vs in ranges of blocks (k_sweep3), first difference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, bwgr_amd
d = np.load(os.path.join(ROOT, "tests", "golden", "tpod.npz")); X = np.asfortranarray(d["gen"]); y = d["y"].astype(np.float64)
for cfg in sys.argv[1:] or [""]:
    for kv in cfg.split(","):
        if "=" in kv: k, v = kv.split("="); os.environ[k] = v
    step = int(os.environ.get("RSTEP", "5")); blk = int(os.environ.get("RBLK", "32")); its = int(os.environ.get("RIT", "6"))
    P = bwgr_amd.Panel(X, block=blk)
    a = bwgr_amd.Chain(P, "BayesB", y, it=its, bi=1, pi=0.8, seed=3)
    b = bwgr_amd.Chain(P, "BayesB", y, it=its, bi=1, pi=0.8, seed=3)
    msg = "equal"
    if os.environ.get("RFIRST"):
        a.run(its); a.sync()
    for it in range(its):
        if not os.environ.get("RFIRST"): a.run(1)
        for lo in range(0, b.nblocks, step): b.sweep_blocks(lo, min(b.nblocks, lo + step))
        b.end_iteration(None)
        if os.environ.get("RFIRST") and it < its - 1: continue
        sa, sb = a.state(), b.state()
        if not np.array_equal(sa["d"], sb["d"]) or np.abs(sa["b"] - sb["b"]).max() > 1e-7:
            w = np.nonzero((sa["d"] != sb["d"]) | (np.abs(sa["b"] - sb["b"]) > 1e-7))[0]
            msg = "iteration %d: first difference at marker %d (block %d, in-block %d), %d markers differ; e diff %.2e" % (it, w[0], w[0] // blk, w[0] % blk, w.size, np.abs(sa["e"] - sb["e"]).max())
            break
    print("%-36s %s  %s" % (cfg, P.pipeline(True), msg), flush=True)
    a.close(); b.close(); P.close()
    for kv in cfg.split(","):
        if "=" in kv: os.environ.pop(kv.split("=")[0], None)
