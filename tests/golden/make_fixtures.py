#!/usr/bin/env python3
"""Extract the reference's bundled data into small JSON fixtures (run once, in the
build container, where /root/reference exists; the GPU box only sees the JSON).

Only DATA is extracted (numeric vectors the reference's own tests and vignette
use); no reference source is read or copied.  Sources (read-only):

  /root/reference/R/sysdata.rda   -> P1annual, P1pc, NPlds, NPcv   (internal data)
  /root/reference/data/NPannual.rda, NPpc.rda, theta.rda           (exported data)

File format: bzip2 stream of R serialization v2, XDR/big-endian ("RDX2\\nX\\n").
The reader below implements just the SEXP types that occur in these four files.

Usage:  python tests/golden/make_fixtures.py  [/root/reference]
"""
import bz2
import json
import os
import struct
import sys


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.o = 0
        self.refs = []

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.o)[0]
        self.o += 4
        return v

    def f64(self, n):
        v = struct.unpack_from(">%dd" % n, self.b, self.o)
        self.o += 8 * n
        return list(v)

    def item(self):
        flags = self.i32()
        ty = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        if ty == 254:                      # NILVALUE
            return None
        if ty == 255:                      # REFSXP
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if ty == 1:                        # SYMSXP
            name = self.item()
            sym = ("sym", name)
            self.refs.append(sym)
            return sym
        if ty == 2:                        # LISTSXP (pairlist) -> list of (tag, value)
            out = []
            while True:
                attr = self.item() if has_attr else None   # noqa: F841
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag[1] if tag else None, car))
                flags = self.i32()
                ty = flags & 0xFF
                if ty == 254:
                    return out
                if ty != 2:
                    raise ValueError("improper pairlist tail %d" % ty)
                has_attr = bool(flags & (1 << 9))
                has_tag = bool(flags & (1 << 10))
        if ty == 9:                        # CHARSXP
            n = self.i32()
            if n == -1:
                return None
            s = self.b[self.o:self.o + n].decode("utf-8", "replace")
            self.o += n
            return s
        if ty in (10, 13):                 # LGLSXP / INTSXP
            n = self.i32()
            v = list(struct.unpack_from(">%di" % n, self.b, self.o))
            self.o += 4 * n
            val = v
        elif ty == 14:                     # REALSXP
            n = self.i32()
            val = self.f64(n)
        elif ty == 16:                     # STRSXP
            n = self.i32()
            val = [self.item() for _ in range(n)]
        elif ty == 19:                     # VECSXP
            n = self.i32()
            val = [self.item() for _ in range(n)]
        elif ty == 22:                     # EXTPTRSXP (data.table selfref)
            ext = ("extptr",)
            self.refs.append(ext)
            self.item()
            self.item()
            val = ext
        else:
            raise ValueError("unhandled SEXPTYPE %d at offset %d" % (ty, self.o))
        if has_attr:
            attrs = dict(self.item())
            return {"value": val, "attr": attrs}
        return val


def read_rda(path):
    raw = bz2.decompress(open(path, "rb").read())
    assert raw[:5] == b"RDX2\n", raw[:8]
    assert raw[5:7] == b"X\n"
    r = _Reader(raw)
    r.o = 7
    r.i32(), r.i32(), r.i32()
    return dict(r.item())


def _val(x):
    return x["value"] if isinstance(x, dict) and "value" in x else x


def _names(x):
    return _val(x["attr"]["names"])


def _df(x):
    """data.frame / data.table -> {column: values}"""
    return {n: _val(c) for n, c in zip(_names(x), x["value"])}


def _matrix(x):
    """R matrix (column-major) -> {"nrow","ncol","colmajor"}"""
    dim = _val(x["attr"]["dim"])
    return {"nrow": dim[0], "ncol": dim[1], "colmajor": x["value"]}


def _theta(x):
    return {n: _val(c) for n, c in zip(_names(x), x["value"])}


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out_dir = os.path.dirname(os.path.abspath(__file__))
    sysd = read_rda(os.path.join(ref, "R", "sysdata.rda"))
    npannual = read_rda(os.path.join(ref, "data", "NPannual.rda"))["NPannual"]
    nppc = read_rda(os.path.join(ref, "data", "NPpc.rda"))["NPpc"]
    theta = read_rda(os.path.join(ref, "data", "theta.rda"))["theta"]

    p1a = _df(sysd["P1annual"])
    p1pc = _df(sysd["P1pc"])
    fx = {
        "_about": "data extracted from the reference's bundled .rda files by make_fixtures.py",
        "P1annual": {"year": p1a["year"], "Qa": p1a["Qa"]},
        "P1pc": {"columns": list(p1pc.keys()),
                 "data": [p1pc[k] for k in p1pc.keys()]},      # one list per PC, 406 long
        "NPannual": {k: v for k, v in _df(npannual).items()},
        "NPpc": {"columns": list(_df(nppc).keys()),
                 "data": [_df(nppc)[k] for k in _df(nppc).keys()]},  # one list per PC, 813 long
        "theta": _theta(theta),
    }
    nplds = sysd["NPlds"]
    nplds_d = dict(zip(_names(nplds), nplds["value"]))
    fx["NPlds"] = {"theta": _theta(nplds_d["theta"]), "lik": _val(nplds_d["lik"])}
    # NPcv: the stored result of  cvLDS(NPannual, u, v, start.year = 1600, num.restarts = 20, Z = Z)
    # with Z = make_Z(NPannual$Qa, nRuns = 30, frac = 0.25, contiguous = TRUE)  (vignettes/ldsr.Rmd:133-139):
    # the 30 folds (1-based indices into the 46 instrumental years), the cross-validated flows of
    # every fold, the target, the per-fold metrics and their Tukey-biweight means
    npcv = dict(zip(_names(sysd["NPcv"]), sysd["NPcv"]["value"]))
    ycv = _df(npcv["Ycv"])
    n_inst = len(_df(npcv["target"])["y"])
    n_rep = len(ycv["Y"]) // n_inst
    fx["NPcv"] = {
        "Z": [_val(z) for z in npcv["Z"]],
        "Ycv": [ycv["Y"][r * n_inst:(r + 1) * n_inst] for r in range(n_rep)],
        "year": ycv["year"][:n_inst],
        "target": _df(npcv["target"])["y"],
        "metrics_dist": _df(npcv["metrics.dist"]),
        "metrics": {k: v[0] for k, v in _df(npcv["metrics"]).items()},
    }
    with open(os.path.join(out_dir, "reference_data.json"), "w") as f:
        json.dump(fx, f)
    print("wrote reference_data.json:",
          {k: (len(v) if hasattr(v, "__len__") else v) for k, v in fx.items()})


if __name__ == "__main__":
    main()
