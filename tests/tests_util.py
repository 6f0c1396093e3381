"""Helpers shared by the tests (test tooling; independent of the product's own checkpoint code)."""
import struct

import numpy as np

# Order of the tensor records in a v1 checkpoint (reference model/vit_cnn.hpp:276-394,
# layers/batch_normalization_layer.hpp:384-505, layers/visual_transformer.hpp:244-270).
CKPT_ORDER = [
    ("Layer", ["conv.W", "conv.b"]),
    ("BatchNorm2D", ["bn.gamma", "bn.beta", "bn.running_mean", "bn.running_var"]),
    ("VisualTransformer", ["tok.linear1.W", "tok.linear1.b", "tok.linear2.W", "tok.linear2.b", "proj.linear1.W",
                           "proj.linear2.W", "proj.linear3.W", "proj.linear3.b", "proj.norm.gamma", "proj.norm.beta",
                           "proj.downsample.W", "proj.downsample.b"]),
    ("Dense", ["fc.W", "fc.b"]),
]


def parse_checkpoint(path):
    """Returns {name: ndarray} from a reference .bin checkpoint."""
    buf = open(path, "rb").read()
    pos = 0

    def i32():
        nonlocal pos
        (v,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        return v

    out = {}
    assert i32() == 4
    for typ, names in CKPT_ORDER:
        n = i32()
        assert buf[pos:pos + n].decode() == typ
        pos += n
        if typ == "VisualTransformer":
            assert i32() == len(names)
        for nm in names:
            nd = i32()
            shape = [i32() for _ in range(nd)]
            cnt = i32()
            out[nm] = np.frombuffer(buf, dtype="<f4", count=cnt, offset=pos).reshape(shape).copy()
            pos += 4 * cnt
    assert pos == len(buf)
    return out


def arena_from_checkpoint(path, layout, arena):
    for nm, v in parse_checkpoint(path).items():
        layout.view(arena, nm)[...] = v.reshape(layout.tensors[nm][2])
    return arena
