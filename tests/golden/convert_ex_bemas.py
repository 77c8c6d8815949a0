"""Convert the reference's data file interpolation/ex_bemas.txt into a plain .npz.

ex_bemas.txt is a protocol-0 (ASCII) pickle of 38 Beam records x 300 gates written
by the reference's own script (interpolation/interpolation.py:382-391).  It is NOT
unpickled here: this is a small text reader for the handful of protocol-0 opcodes
the file contains.  It imports nothing named in the file and calls nothing from it;
class / function names are kept as inert strings, and only three shapes of record
are understood: a numpy ndarray (shape tuple + dtype code + raw byte string), a
numpy scalar, and the attribute dict of a Beam.  Anything else raises.

Run once in the build container (the reference is not available on the GPU box):
    python tests/golden/convert_ex_bemas.py
writes tests/golden/ex_bemas.npz (arrays only).
"""
import codecs
import os
import sys

import numpy as np

SRC = os.environ.get("COSMO_POL_REFERENCE", "/root/reference") + "/interpolation/ex_bemas.txt"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ex_bemas.npz")

MARK = object()


class Sym(object):
    """An inert name (module.attr) or a pending record; never resolved to code."""

    def __init__(self, name, args=None, state=None):
        self.name, self.args, self.state = name, args, state


def _unquote(line):
    s = line.strip()
    assert s[0] in "'\"" and s[-1] == s[0], s[:20]
    return codecs.escape_decode(s[1:-1].encode("latin1"))[0]


def _finish(obj):
    """Turn a fully described record into plain data."""
    if isinstance(obj, Sym) and obj.name == "numpy.core.multiarray._reconstruct":
        ver, shape, dt, fortran, raw = obj.state
        assert isinstance(dt, Sym) and dt.name == "numpy.dtype" and not fortran
        code = dt.args[0].decode("ascii")
        order = dt.state[1].decode("ascii")
        assert code in ("f4", "f8", "i8", "i4") and order in ("<", "|", "=")
        return np.frombuffer(raw, dtype=np.dtype("<" + code)).reshape(shape).copy()
    if isinstance(obj, Sym) and obj.name == "numpy.core.multiarray.scalar":
        dt, raw = obj.args
        code = dt.args[0].decode("ascii")
        return np.frombuffer(raw, dtype=np.dtype("<" + code))[0]
    return obj


def read_protocol0(path):
    stack, memo = [], {}
    with open(path, "rb") as fh:
        lines = fh.read().decode("latin1").split("\n")
    i = 0

    def pop_mark():
        k = len(stack) - 1
        while stack[k] is not MARK:
            k -= 1
        items = stack[k + 1:]
        del stack[k:]
        return items

    while i < len(lines):
        line = lines[i]
        i += 1
        pos = 0
        while pos < len(line):
            op = line[pos]
            pos += 1
            rest = line[pos:]
            if op == "(":
                stack.append(MARK)
            elif op == "l":
                stack.append(list(pop_mark()))
            elif op == "d":
                items = pop_mark()
                stack.append(dict(zip(items[0::2], items[1::2])))
            elif op == "t":
                stack.append(tuple(pop_mark()))
            elif op == "N":
                stack.append(None)
            elif op == "p":
                memo[int(rest)] = stack[-1]
                break
            elif op == "g":
                stack.append(memo[int(rest)])
                break
            elif op == "S":
                stack.append(_unquote(rest))
                break
            elif op == "I":
                stack.append({"00": False, "01": True}.get(rest, None) if rest in ("00", "01") else int(rest))
                break
            elif op == "F":
                stack.append(float(rest))
                break
            elif op == "c":
                stack.append(Sym(rest + "." + lines[i]))
                i += 1
                break
            elif op == "i":
                name = rest + "." + lines[i]
                i += 1
                assert name == "cosmo_pol.utilities.beam.Beam", name
                pop_mark()
                stack.append(Sym(name))
                break
            elif op == "R":
                args = stack.pop()
                fn = stack.pop()
                assert isinstance(fn, Sym) and fn.name in (
                    "numpy.core.multiarray._reconstruct", "numpy.dtype", "numpy.core.multiarray.scalar"), fn.name
                rec = Sym(fn.name, args=args)
                stack.append(_finish(rec) if fn.name.endswith("scalar") else rec)
            elif op == "b":
                state = stack.pop()
                rec = stack[-1]
                assert isinstance(rec, Sym)
                rec.state = state
                if rec.name.endswith("_reconstruct"):
                    stack[-1] = _finish(rec)
                    for k, v in list(memo.items()):
                        if v is rec:
                            memo[k] = stack[-1]
            elif op == "a":
                v = stack.pop()
                stack[-1].append(v)
            elif op == "s":
                v = stack.pop()
                k = stack.pop()
                stack[-1][k] = v
            elif op == ".":
                return stack[-1]
            else:
                raise ValueError("unsupported opcode %r at line %d" % (op, i))
    raise ValueError("no STOP opcode")


def main():
    beams = read_protocol0(SRC)
    out = {}
    nb = len(beams)
    st = [b.state for b in beams]
    key = lambda d, k: d[k.encode("ascii")]
    for name in ("elev_profile", "heights_profile", "dist_profile", "lats_profile", "lons_profile", "mask"):
        out[name] = np.stack([np.asarray(key(s, name)) for s in st])
    out["GH_pt"] = np.array([[float(x) for x in key(s, "GH_pt")] for s in st], dtype=np.float64)
    out["GH_weight"] = np.array([float(key(s, "GH_weight")) for s in st], dtype=np.float64)
    for v in ("U", "V", "W", "T"):
        out["val_" + v] = np.stack([np.asarray(key(s, "values")[v.encode("ascii")]) for s in st])
    np.savez_compressed(DST, **out)
    print("wrote", DST, "beams:", nb, {k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    sys.exit(main())
