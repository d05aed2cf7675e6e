"""Compile-time variants of the kernels side by side (experiments).
  here:        python tools/variants.py build TAG="-DFOO=1 -DBAR" TAG2="..."      -> tools/_bin/libbwgr_TAG.so (git-ignored, travels to the GPU box)
  on the box:  python tools/variants.py run [TAG ...]                              -> the C4 bench (one chain, no CPU leg) and the sequencer alone, per variant
A tag named "base" builds with no extra flag.  Every variant is built with -DBWGR_EXPERIMENTS as well (TAG_exp) for the BWGR_DBG3 switches."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "tools", "_bin")


def build(specs):
    from bwgr_amd import build as B
    os.makedirs(BIN, exist_ok=True)
    procs = []
    for spec in specs:
        tag, _, flags = spec.partition("=")
        for suf, extra in (("", []), ("_exp", ["-DBWGR_EXPERIMENTS"])):
            so = os.path.join(BIN, "libbwgr_%s%s.so" % (tag, suf))
            cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + flags.split() + extra + ["-w", "-o", so] + B.SOURCES
            procs.append((so, subprocess.Popen(cmd)))
            if len(procs) >= 4:
                for so_, p in procs:
                    if p.wait() != 0: raise SystemExit("build failed: " + so_)
                procs = []
    for so_, p in procs:
        if p.wait() != 0: raise SystemExit("build failed: " + so_)


def run(tags):
    if not tags:
        tags = sorted(f[len("libbwgr_"):-3] for f in os.listdir(BIN) if f.startswith("libbwgr_") and f.endswith(".so") and not f.endswith("_exp.so"))
    exps = os.environ.get("VAR_EXP", "BWGR_DBG3=2056").split()
    for tag in tags:
        env = dict(os.environ, BWGR_LIB=os.path.join(BIN, "libbwgr_%s.so" % tag))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu", "--no-extra", "--shards", "0", "--chains", "1"],
                             env=env, capture_output=True, text=True, timeout=600)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            line = "%-16s iter/s %6.2f  kernel %7.3f ms  mean_d %.6f ve %.4f" % (tag, d["value"], d["roofline"]["kernel_ms"], d["chain_check"]["mean_d"], d["chain_check"]["ve"])
        except Exception:
            line = "%-16s FAILED: %s" % (tag, (out.stderr or out.stdout)[-300:])
        print(line, flush=True)
        if exps and exps != ["-"]:
            env = dict(os.environ, BWGR_LIB=os.path.join(BIN, "libbwgr_%s_exp.so" % tag), AB_P="1000000", AB_PREBUILT="1")
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab3_probe.py")] + exps, env=env, capture_output=True, text=True, timeout=900)
            for ln in out.stdout.splitlines():
                if "us/block" in ln: print("    " + ln[:96], flush=True)


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in ("build", "run"):
        raise SystemExit(__doc__)
    (build if sys.argv[1] == "build" else run)(sys.argv[2:])
