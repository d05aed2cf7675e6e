import numpy as np
rng = np.random.default_rng(1)
f32 = np.float32; f64 = np.float64
def exact_x(ra, c):
    b1 = f32(ra + f64(c['xxb0'])) if False else f32(np.float64(np.float64(ra + f64(c['xxb0'])) * c['rden'] + c['sdz1']))  # fma approx (double rounding tiny)
    d1f = f32(b1 - c['b0']); D1 = f64(d1f); D2 = f64(c['d2f'])
    diffd = 2.0 * ra * (D1 - D2) + c['gjj'] * (D2 * D2 - D1 * D1)
    return f32(c['Cc'] * f32(diffd))
def quick(c):
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
    return zc, ha, hr, A, Qmin
bad = 0; und = 0; tot = 0; nearchecks = 0
for trial in range(20000):
    n = 10 ** rng.uniform(2, 5); xx = f32(n * rng.uniform(0.1, 2.0)); ve = f32(10 ** rng.uniform(-3, 3)); lam = f32(xx * 10 ** rng.uniform(-3, 1))
    b0 = f32(0.0) if rng.random() < 0.5 else f32(rng.normal() * np.sqrt(ve / (xx + lam)) * 3)
    den = f32(xx + lam); sd = f32(np.sqrt(f32(ve / den)))
    c = dict(b0=b0, xxb0=f32(xx * b0), rden=1.0 / f64(den), sdz1=f64(sd) * rng.normal(), gjj=f64(xx), d2f=f32(f32(0.0) - b0), Cc=f32(-0.5 / np.sqrt(ve)))
    # hmm the reference's C is -0.5/sqrt(ve)?? keep
    u = rng.random(); odds = 10 ** rng.uniform(-2, 2)
    ta = f32(np.log1p(-u * (1 + 1e-6)) - np.log(u * (1 + 1e-6)) - np.log(odds)); tr = f32(np.log1p(-u * (1 - 1e-6)) - np.log(u * (1 - 1e-6)) - np.log(odds))
    ta = np.nextafter(ta, f32(-np.inf)); tr = np.nextafter(tr, f32(np.inf))
    c['tacc'] = ta; c['trej'] = tr
    zc, ha, hr, A, Qmin = quick(c)
    # probe r values: random and right at the boundaries
    rs = list(rng.normal(size=4) * np.sqrt(xx * ve) * 3)
    for h in (ha, hr):
        if np.isfinite(h) and h > 0:
            for s in (-1, 1):
                for eps in (0, 1e-15, -1e-15, 1e-12, -1e-12, 1e-9, -1e-9, 3e-8, -3e-8, 1e-7,-1e-7, 1e-6, -1e-6):
                    rs.append(zc + s * h * (1 + eps)); nearchecks += 1
    for ra in rs:
        ra = f64(ra); z = ra - zc
        qa_ = abs(z) > ha; qr_ = abs(z) < hr
        x = exact_x(ra, c)
        ea = x < c['tacc']; er = x > c['trej']
        tot += 1
        if qa_ and qr_: bad += 1; print("both", trial)
        if qa_ and not ea: bad += 1; print("bad accept", trial, x, c['tacc'], z, ha)
        if qr_ and not er: bad += 1; print("bad reject", trial, x, c['trej'], z, hr)
        if not (qa_ or qr_): und += 1
print("bad", bad, "und", und, "tot", tot, "near", nearchecks)
