"""bf16 mode at the wide shape: per-tensor gradient distance to exact arithmetic (fp64 oracle with the run's own pooling
winners and ReLU mask imposed), for a batch of N synthetic images.  VT_BF16_PRODUCTS selects which products are bf16.
    python tools/diag_bf16_sets.py <N> [shape]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pyoracle import Oracle, make_dims
from vit_amd import VTCNN, Dims
n = int(sys.argv[1])
shape = sys.argv[2] if len(sys.argv) > 2 else "wide"
dims = dict(wide=dict(num_classes=8, image_channels=3, input_dim=224, tokens=196, token_channels=512, vt_channels=128, heads=8),
            mid=dict(num_classes=8, image_channels=3, input_dim=56, tokens=49, token_channels=128, vt_channels=64, heads=4))[shape]
rng = np.random.default_rng(11)
x = (rng.integers(0, 256, size=(n, dims["image_channels"], dims["input_dim"], dims["input_dim"])).astype(np.float32) / np.float32(255.0))
y = rng.integers(0, dims["num_classes"], size=n).astype(np.int32)
m = VTCNN(Dims(**dims), max_batch=n, seed=42)
m.set_matmul_precision(True)
arena0 = m.get_arena().copy()
p = m.forward(torch.from_numpy(x).cuda()).cpu().numpy()
m.backward(torch.from_numpy(y).cuda())
gr = m.get_grads()
H, C2 = dims["input_dim"], dims["vt_channels"]
pk = H // 14
idx = m.read("fwd.pool_idx").view(np.int32)[:n * C2 * 196].reshape(n, C2, 14, 14)
ii, jj = np.arange(14).reshape(1, 1, 14, 1), np.arange(14).reshape(1, 1, 1, 14)
slots = ((idx // H - ii * pk) * pk + (idx % H - jj * pk)).astype(np.uint8).ravel()
mask = (m.read("bwd.du") != 0).astype(np.uint8)
o = Oracle(make_dims(**dims), n, f64=True)
a64 = arena0.astype(np.float64)
p64 = o.forward(a64.copy(), x, training=True)
print("mask", os.environ.get("VT_BF16_PRODUCTS", "default"), "N", n, "preds err %.2e" % (np.abs(p - p64).max() / np.abs(p64).max()))
ref_slots, gap = o.pool_decisions()
print("  windows re-decided: %d of %d" % ((ref_slots != slots).sum(), slots.size))
o.set_routing(slots, mask)
o.forward(a64.copy(), x, training=True)
go = o.backward(a64, y)
worst = []
for nm, (off, size, shp, kind) in m.layout.tensors.items():
    if kind == 0 and nm != "mha.o.W":
        den = np.abs(go[off:off + size]).max()
        e = np.abs(gr[off:off + size] - go[off:off + size]).max() / den if den > 0 else 0.0
        worst.append((e, nm))
worst.sort(reverse=True)
print("  " + "  ".join("%s %.1e" % (nm, e) for e, nm in worst[:9]))
