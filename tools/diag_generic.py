"""Per-tensor distances between the general-shape HIP step and the CPU oracle (debug aid)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from vit_amd import VTCNN, Dims
from oracle.pyoracle import Oracle, make_dims

def rel(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

dims = dict(num_classes=5, image_channels=2, input_dim=42, tokens=10, token_channels=24, vt_channels=48, heads=3)
n = 9
rng = np.random.default_rng(5)
x = (rng.integers(0, 256, size=(n, 2, 42, 42)).astype(np.float32) / np.float32(255.0))
y = rng.integers(0, 5, size=n).astype(np.int32)
m = VTCNN(Dims(**dims), max_batch=n, seed=42)
arena = m.get_arena().copy()
o = Oracle(make_dims(**dims), n)
po = o.forward(arena.copy(), x, training=True)
go = o.backward(arena, y)
p = m.forward(torch.from_numpy(x).cuda()).cpu().numpy()
print("preds", rel(p, po))
m.backward(torch.from_numpy(y).cuda())
for nm in ("fwd.conv_out", "fwd.x", "fwd.tok_A", "fwd.tok_out", "fwd.tr_a", "fwd.tr_hdn", "fwd.tr_s2", "fwd.tr_out", "fwd.proj_a",
           "fwd.proj_z", "fwd.pooled", "bwd.d_logits", "bwd.d_pooled", "bwd.proj_dz", "bwd.proj_dx", "bwd.proj_dt",
           "bwd.tr_ds2", "bwd.tr_dr", "bwd.tok_dx"):
    try:
        a = m.read(nm); b = o.get(nm)
        if nm == "fwd.conv_out":  # oracle keeps NCHW
            b = b.reshape(n, 24, -1).transpose(0, 2, 1).ravel()
        if nm == "bwd.tok_dx":
            print(nm, "(device holds the BatchNorm2D input delta here)")
            b2 = o.get("bwd.bn_dx").reshape(n, 24, -1).transpose(0, 2, 1).ravel()
            print("  vs bn_dx", rel(a[:b2.size], b2))
            continue
        print(nm, rel(a[:b.size], b))
    except KeyError as e:
        print(nm, "missing", e)
gg = m.get_grads()
for nm, (off, size, shape, kind) in m.layout.tensors.items():
    if kind == 0:
        print("g." + nm, rel(gg[off:off + size], go[off:off + size]) if np.abs(go[off:off+size]).max() > 0 else "zero")
