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


STRIDE, BIG = 7, 4096  # tools/make_golden_shapes.py: tensors above BIG elements are stored strided plus their sum


def shape_case(g):
    """(dims kwargs, n, k, seed, x [n,ci,h,h], labels) of a tests/golden/shape_*.npz fixture; the
    inputs are regenerated exactly as the reference harness draws them."""
    ci, h, tokens, d, c2, heads, k, n, seed = (int(v) for v in g["meta.dims"])
    raw = np.random.RandomState(seed)._bit_generator.random_raw(n * ci * h * h + n).astype(np.uint64)
    x = ((raw[:n * ci * h * h] & 0xFF).astype(np.float32) / np.float32(255.0)).reshape(n, ci, h, h)
    y = (raw[n * ci * h * h:] % k).astype(np.int32)
    dims = dict(num_classes=k, image_channels=ci, input_dim=h, tokens=tokens, token_channels=d, vt_channels=c2, heads=heads)
    return dims, n, k, seed, x, y


def compare_compact(g, key, full, tol, exact=False):
    """Compares a full tensor with its fixture entry (strided + sum when the tensor is big).
    Returns the relative error max|a-b| / max|b| over the stored elements."""
    want = g[key]
    full = np.asarray(full).ravel()
    if key + ".sum" in g:
        got = full[::STRIDE]
        scale = max(float(np.abs(full).sum()), 1e-30)
        assert abs(float(full.astype(np.float64).sum()) - float(g[key + ".sum"][0])) <= 10 * tol * scale, key
    else:
        got = full
        want = want.ravel()
    assert got.shape == want.shape, (key, got.shape, want.shape)
    if exact:
        assert np.array_equal(got, want), key
        return 0.0
    err = float(np.abs(got.astype(np.float64) - want).max() / max(float(np.abs(want).max()), 1e-30))
    assert err < tol, (key, err)
    return err
