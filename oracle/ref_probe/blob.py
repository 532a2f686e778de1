"""Tiny named-array container shared by make_golden.py and ref_probe.cpp (records: name, dtype code,
ndim, dims (int64), raw little-endian data)."""
import struct

import numpy as np

CODES = {"u1": 0, "i2": 1, "i4": 2, "f8": 3, "i8": 4}
DT = {0: np.uint8, 1: np.int16, 2: np.int32, 3: np.float64, 4: np.int64}


def write(path, arrays):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            code = CODES[a.dtype.str[1:]]
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)))
            f.write(nb)
            f.write(struct.pack("<ii", code, a.ndim))
            f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(a.tobytes())


def read(path):
    out = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<i", f.read(4))
        for _ in range(n):
            (ln,) = struct.unpack("<i", f.read(4))
            name = f.read(ln).decode()
            code, nd = struct.unpack("<ii", f.read(8))
            dims = struct.unpack("<%dq" % nd, f.read(8 * nd)) if nd else ()
            dt = np.dtype(DT[code])
            cnt = int(np.prod(dims)) if nd else 1
            out[name] = np.frombuffer(f.read(cnt * dt.itemsize), dtype=dt).reshape(dims).copy()
    return out
