"""Reader for the tensor bundles written by oracle/ref_harness.cpp (test tooling only).

Record layout: i32 name_len, name bytes, i32 dtype (0 = float32, 1 = int32), i32 ndim,
i32 dims[ndim], payload (little-endian, C order).
"""
import struct
from collections import OrderedDict

import numpy as np


def read_bundle(path):
    out = OrderedDict()
    with open(path, "rb") as f:
        buf = f.read()
    pos = 0
    while pos < len(buf):
        (nl,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        name = buf[pos:pos + nl].decode()
        pos += nl
        dtype, nd = struct.unpack_from("<ii", buf, pos)
        pos += 8
        dims = struct.unpack_from("<%di" % nd, buf, pos)
        pos += 4 * nd
        count = int(np.prod(dims)) if nd else 1
        dt = np.float32 if dtype == 0 else np.int32
        out[name] = np.frombuffer(buf, dtype=dt, count=count, offset=pos).reshape(dims).copy()
        pos += 4 * count
    return out
