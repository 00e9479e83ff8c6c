"""Reader/writer for the tiny named-array container ("PHYG") that oracle/ref_driver.c emits.

Layout: b"PHYG", then records  u32 name_len | name | u8 dtype | u32 ndim | u64 dims[ndim] | payload
with dtype 0=f64, 1=i32, 2=i16, 3=u8 (little endian).  Used for golden vectors and model blocks (data files only).
"""
import struct
import numpy as np

_DT = {0: np.dtype("<f8"), 1: np.dtype("<i4"), 2: np.dtype("<i2"), 3: np.dtype("u1")}
_RDT = {np.dtype("float64"): 0, np.dtype("int32"): 1, np.dtype("int16"): 2, np.dtype("uint8"): 3}


def load(path):
    raw = open(path, "rb").read()
    if raw[:4] != b"PHYG":
        raise ValueError(f"{path}: not a PHYG file")
    off, out = 4, {}
    while off < len(raw):
        (nl,) = struct.unpack_from("<I", raw, off); off += 4
        name = raw[off:off + nl].decode(); off += nl
        dt = raw[off]; off += 1
        (nd,) = struct.unpack_from("<I", raw, off); off += 4
        dims = struct.unpack_from(f"<{nd}Q", raw, off); off += 8 * nd
        n = int(np.prod(dims)) if nd else 1
        arr = np.frombuffer(raw, dtype=_DT[dt], count=n, offset=off).reshape(dims)
        off += n * _DT[dt].itemsize
        out[name] = arr.copy()
    return out


def save(path, arrays):
    with open(path, "wb") as f:
        f.write(b"PHYG")
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if a.ndim == 0:
                a = a.reshape(1)
            dt = _RDT[a.dtype]
            nb = name.encode()
            f.write(struct.pack("<I", len(nb))); f.write(nb)
            f.write(bytes([dt])); f.write(struct.pack("<I", a.ndim))
            f.write(struct.pack(f"<{a.ndim}Q", *a.shape))
            f.write(a.astype(_DT[dt]).tobytes())
