"""The selection rounds' two compares (lane_quick, bwgr_amd/csrc/sweep.hip.h) against the test they stand for (lane_accept), both
restated in numpy with the device code's roundings: for random marker constants, probe residual dots straddling the two radii at
relative distances 1e-15 .. 1e-6 (and at random) and count contradictions -- a "certain accept" that lane_accept's thresholds do
not accept, a "certain reject" they do not reject.  python tools/quick_accept_check.py [trials]"""
import sys
import numpy as np
f32 = np.float32; f64 = np.float64


def exact_x(ra, c):   # lane_b1 + lane_accept's x
    b1 = f32(f64(f64(ra + f64(c['xxb0'])) * c['rden'] + c['sdz1']))
    d1f = f32(b1 - c['b0']); D1 = f64(d1f); D2 = f64(c['d2f'])
    diffd = 2.0 * ra * (D1 - D2) + c['gjj'] * (D2 * D2 - D1 * D1)
    return f32(c['Cc'] * f32(diffd))


def quick(c):         # lane_quick
    b0 = f64(c['b0']); D2 = f64(c['d2f']); rden = c['rden']; gjj = c['gjj']
    k1 = f64(c['xxb0']) * rden + c['sdz1'] - b0
    gr = gjj * rden; A = rden * (2.0 - gr); Bh = (k1 - D2) - gr * k1; C0 = gjj * (D2 * D2 - k1 * k1)
    zc = -Bh / A; Qmin = C0 + Bh * zc
    qa = f64(c['tacc']) / f64(c['Cc']); qr = f64(c['trej']) / f64(c['Cc'])

    def err(h, q):
        ra = abs(zc) + h; d1 = rden * ra + abs(k1); t = d1 + abs(b0); eD = 6.1e-8 * (t + d1)
        return 4.0 * (2.0 * (ra + gjj * d1) * eD + gjj * eD * eD) + 2.5e-7 * abs(q) + 1e-13 * (2.0 * ra * (d1 + abs(D2)) + gjj * (D2 * D2 + d1 * d1))
    ha, hr = np.inf, -1.0
    if A > 0 and np.isfinite(zc):
        if np.isfinite(qa):
            h0 = np.sqrt(max(0.0, (qa - Qmin) / A)); num = qa - Qmin + err(h0, qa); ha = np.sqrt(num / A) * (1 + 1e-12) if num > 0 else -1.0
        if qr == np.inf: hr = np.inf
        elif np.isfinite(qr):
            h0 = np.sqrt(max(0.0, (qr - Qmin) / A)); num = qr - Qmin - err(h0, qr); hr = np.sqrt(num / A) * (1 - 1e-12) if num > 0 else -1.0
    return zc, ha, hr


def run(trials=20000, seed=1, alt_b2=False):
    rng = np.random.default_rng(seed)
    bad = und_random = tot = 0
    with np.errstate(all="ignore"):
        for trial in range(trials):
            n = 10 ** rng.uniform(2, 5); xx = f32(n * rng.uniform(0.1, 2.0)); ve = f32(10 ** rng.uniform(-3, 3)); lam = f32(xx * 10 ** rng.uniform(-3, 2))
            den = f32(xx + lam); sd = f32(np.sqrt(f32(ve / den)))
            b0 = f32(0.0) if rng.random() < 0.5 else f32(rng.normal() * sd * 3)
            b2 = f32(sd * rng.normal()) if alt_b2 else f32(0.0)
            c = dict(b0=b0, xxb0=f32(xx * b0), rden=1.0 / f64(den), sdz1=f64(sd) * rng.normal(), gjj=f64(xx), d2f=f32(b2 - b0), Cc=f32(-0.5 / np.sqrt(ve)))
            u = rng.random(); odds = 10 ** rng.uniform(-2, 2)
            ta = f32(np.log1p(-u * (1 + 1e-6)) - np.log(u * (1 + 1e-6)) - np.log(odds)); tr = f32(np.log1p(-u * (1 - 1e-6)) - np.log(u * (1 - 1e-6)) - np.log(odds))
            c['tacc'] = np.nextafter(ta, f32(-np.inf)); c['trej'] = np.nextafter(tr, f32(np.inf))
            zc, ha, hr = quick(c)
            rs = [(r, True) for r in rng.normal(size=4) * np.sqrt(xx * ve) * 3]
            for h in (ha, hr):
                if np.isfinite(h) and h > 0:
                    for sgn in (-1, 1):
                        for eps in (0, 1e-15, -1e-15, 1e-12, -1e-12, 1e-9, -1e-9, 3e-8, -3e-8, 1e-7, -1e-7, 1e-6, -1e-6):
                            rs.append((zc + sgn * h * (1 + eps), False))
            for ra, rnd in rs:
                ra = f64(ra); z = ra - zc
                qacc = abs(z) > ha; qrej = abs(z) < hr
                x = exact_x(ra, c)
                tot += 1
                bad += int(qacc and qrej) + int(qacc and not (x < c['tacc'])) + int(qrej and not (x > c['trej']))
                und_random += int(rnd and not (qacc or qrej))
    return bad, und_random, tot


if __name__ == "__main__":
    t = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    for alt in (False, True):
        bad, und, tot = run(t, 1, alt)
        print("alt_b2=%s: contradictions %d, undecided among %d random residual dots %d, probes %d" % (alt, bad, 4 * t, und, tot))
