"""Extract the reference's only data fixture, data/tpod.RData, into tests/golden/tpod.npz.

tpod.RData is an XZ-compressed RDX2 (XDR) serialisation of four objects (man/tpod.Rd:1-21):
y double[196], gen integer[196x376] in {0,1,2}, fam, chr.  This is DATA (inputs only; the
reference holds no expected outputs).  Run once in the build container:
    python tests/golden/make_tpod_fixture.py /root/reference/data/tpod.RData
"""
import lzma, struct, sys
import numpy as np


class R:
    def __init__(self, buf):
        self.b, self.o = buf, 0

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.o)[0]; self.o += 4; return v

    def f64n(self, n):
        v = np.frombuffer(self.b, dtype=">f8", count=n, offset=self.o).astype(np.float64); self.o += 8 * n; return v

    def i32n(self, n):
        v = np.frombuffer(self.b, dtype=">i4", count=n, offset=self.o).astype(np.int32); self.o += 4 * n; return v

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
        if t == 254:  # NILVALUE
            return None
        if t == 255:  # REFSXP
            return self.refs[(flags >> 8) - 1]
        if t == 1:  # SYMSXP
            s = self.item(); self.refs.append(s); return s
        if t == 2:  # LISTSXP (pairlist)
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                flags = self.i32(); t = flags & 0xFF
                has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
                if t == 254:
                    return out
                assert t == 2, t
        if t == 9:  # CHARSXP
            n = self.i32()
            if n == -1:
                return None
            s = self.b[self.o:self.o + n].decode("latin1"); self.o += n; return s
        if t == 13:
            v = self.i32n(self.i32())
        elif t == 14:
            v = self.f64n(self.i32())
        elif t == 16:
            v = [self.item() for _ in range(self.i32())]
        elif t == 19:
            v = [self.item() for _ in range(self.i32())]
        else:
            raise ValueError("unhandled SEXP type %d at %d" % (t, self.o))
        attrs = dict((k, a) for k, a in self.item()) if has_attr else {}
        return (v, attrs) if attrs else v


def main(path):
    raw = lzma.decompress(open(path, "rb").read())
    assert raw[:5] == b"RDX2\n" or raw[:5] == b"RDX3\n", raw[:5]
    r = R(raw); r.o = 5; r.refs = []
    assert raw[r.o:r.o + 2] == b"X\n"; r.o += 2
    r.i32(); r.i32(); r.i32()  # version triple
    if raw[:5] == b"RDX3\n":
        n = r.i32(); r.o += n
    objs = dict(r.item())
    y = objs["y"]; y = y[0] if isinstance(y, tuple) else y
    gen, gattr = objs["gen"]
    dim = gattr["dim"]; dim = dim[0] if isinstance(dim, tuple) else dim
    n, p = int(dim[0]), int(dim[1])
    gen = np.asarray(gen).reshape((p, n)).T  # R is column-major
    fam = objs["fam"]; fam = fam[0] if isinstance(fam, tuple) else fam
    chrv = objs["chr"]; chrv = chrv[0] if isinstance(chrv, tuple) else chrv
    assert y.shape == (196,) and gen.shape == (196, 376) and set(np.unique(gen)) <= {0, 1, 2}
    out = __file__.rsplit("/", 1)[0] + "/tpod.npz"
    np.savez_compressed(out, y=np.asarray(y, np.float64), gen=np.asfortranarray(gen.astype(np.int8)),
                        fam=np.asarray(fam, np.float64), chr=np.asarray(chrv, np.float64))
    print("wrote", out, "var(y)=%.5f" % y.var(ddof=1), "MSx=%.3f" % gen.astype(float).var(axis=0, ddof=1).sum(),
          "mean xx=%.3f" % (gen.astype(float) ** 2).sum(axis=0).mean(), "counts", np.bincount(gen.ravel()))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data/tpod.RData")
