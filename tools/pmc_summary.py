"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/<round>_pmc_<workload>.json.

    python tools/pmc_summary.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <out.json> <workload> <n> <p>

Per-kernel means over launches of the counter values (KiB), the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports
half of the bytes of coalesced streaming reads; calibrated in the same run on the Gram-build kernels, whose read volume is
known: k_gram_i8 reads X once, k_gramx_i8 twice per distance), and the traffic of the dominant sweep kernel per launch."""
import csv, glob, json, os, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            rows = list(csv.DictReader(fh))
        per_dispatch = defaultdict(float)
        names = {}
        for r in rows:
            if r.get("Counter_Name") != counter:
                continue
            key = (f, r.get("Dispatch_Id"))
            per_dispatch[key] += float(r["Counter_Value"])       # summed over XCDs / instances
            names[key] = r["Kernel_Name"]
        for key, v in per_dispatch.items():
            acc[names[key]].append(v)
    # a device-side gate leaves some launches idle (the engine the chain's inclusion rate did not choose; the range-recovery launch
    # of a sweep that stayed in range): they fetch a few KiB.  Means are over the launches that did the work.
    out = {}
    for k, v in acc.items():
        top = max(v)
        live = [x for x in v if x >= 0.01 * top] if top > 0 else v
        out[k] = (sum(live) / len(live), len(live), len(v) - len(live))
    return out


def main():
    dfetch, dwrite, out, workload, n, p = sys.argv[1:7]
    n, p = int(n), int(p)
    fetch, write = load(dfetch, "FETCH_SIZE"), load(dwrite, "WRITE_SIZE")
    sweep = [k for k in fetch if "k_sweep" in k and "finish" not in k]
    if not sweep:
        raise SystemExit("no sweep kernel dispatch in %s (kernels: %s)" % (dfetch, sorted(fetch)[:8]))
    sk = max(sweep, key=lambda k: fetch[k][0])   # (with the device-side engine choice both engines are dispatched; the idle one fetches nothing)
    calib = {k[:40]: fetch[k][0] for k in fetch if "k_gram_i8" in k or "k_gramx_i8" in k}
    gi = [fetch[k][0] for k in fetch if "k_gram_i8" in k]
    factor = None
    if gi:
        factor = (float(n + (-n) % 128) * p) / (gi[0] * 1024.0)   # k_gram_i8 reads the padded panel once
    corr = factor if factor and 1.5 < factor < 2.5 else 2.0
    rd = fetch[sk][0] * 1024.0 * corr
    wr = write.get(sk, (0.0, 0, 0))[0] * 1024.0
    res = {
        "workload": workload, "n": n, "p": p, "kernel": sk[:60], "commit": os.environ.get("BWGR_COMMIT", "unknown"),
        "all_sweep_kernels_fetch_kib": {k[:60]: fetch[k][0] for k in sweep},
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), bench.py --no-cpu; tools/pmc_summary.py",
        "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch (mean over %d working launches, summed over XCDs; %d idle launches of the same kernel -- "
                 "device-side gate: range-recovery launch of a sweep that stayed in range, or the engine not chosen -- left out)" % (fetch[sk][1], fetch[sk][2]),
        "correction": "gfx950: FETCH_SIZE reports ~1/2 of the bytes of coalesced streaming reads (MI355X_MICROARCH.md, HBM); factor %.3f "
                      "as the guide prescribes unless a kernel of known read volume calibrates it in the same run; WRITE_SIZE exact" % corr,
        "fetch_size_kib": fetch[sk][0], "write_size_kib": write.get(sk, (0.0, 0, 0))[0],
        "calibration_fetch_size_kib": calib,
        "read_bytes_corrected": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": float(n) * float(p),
        "traffic_over_algorithmic": (rd + wr) / (float(n) * float(p)),
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
