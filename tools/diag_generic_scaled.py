"""General-shape HIP step at the scaled shape vs the fp64 oracle with the HIP path's discrete decisions imposed."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from tests_util import shape_case
from vit_amd import VTCNN, Dims
from oracle.pyoracle import Oracle, make_dims

def rel(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

g = load_golden("shape_%s_n2.npz" % (sys.argv[1] if len(sys.argv) > 1 else "scaled"))
dims, n, k, seed, x, y = shape_case(g)
H = dims["input_dim"]; pk = H // 14; C2 = dims["vt_channels"]
m = VTCNN(Dims(**dims), max_batch=n, seed=42)
arena = m.get_arena().copy()
p = m.forward(torch.from_numpy(x).cuda()).cpu().numpy()
m.backward(torch.from_numpy(y).cuda())
gg = m.get_grads()
idx = m.read("fwd.pool_idx").view(np.int32)[:n * C2 * 196].reshape(n, C2, 14, 14)
hh, ww = idx // H, idx % H
ii = np.arange(14).reshape(1, 1, 14, 1); jj = np.arange(14).reshape(1, 1, 1, 14)
slots = ((hh - ii * pk) * pk + (ww - jj * pk)).astype(np.uint8).ravel()
mask = (m.read("bwd.du") != 0).astype(np.uint8)
o = Oracle(make_dims(**dims), n, f64=True)
a64 = arena.astype(np.float64)
p64 = o.forward(a64.copy(), x, training=True)
rs, gap = o.pool_decisions()
diff = np.flatnonzero(rs != slots)
print("preds", rel(p, p64), "pool flips", diff.size, "of", slots.size, "max gap", gap[diff].max() if diff.size else 0,
      "pooled max", np.abs(o.get("fwd.pooled")).max())
o.set_routing(slots, mask)
o.forward(a64.copy(), x, training=True)
go = o.backward(a64, y)
for nm, (off, size, shape, kind) in m.layout.tensors.items():
    if kind == 0 and np.abs(go[off:off+size]).max() > 0:
        print("g." + nm, rel(gg[off:off + size], go[off:off + size]))
# the oracle's own decisions (fp32 build, nothing imposed) against the HIP path's
o32 = Oracle(make_dims(**dims), n)
o32.forward(arena.copy(), x, training=True)
g32 = o32.backward(arena, y)
dg = o32.get("bwd.d_projout"); pbn = o32.get("fwd.proj_bn")
own = (dg != 0) & (pbn > 0)
dev = m.read("bwd.du")[:own.size] != 0
bad = np.flatnonzero(own != dev)
print("mask mismatches", bad.size, "routed", int((dg != 0).sum()), "active own", int(own.sum()), "active dev", int(dev.sum()))
if bad.size:
    print(" pbn at mismatches", pbn[bad][:10], " dg", dg[bad][:10], "z dev", m.read("fwd.proj_z")[bad][:10], "z own", o32.get("fwd.proj_z")[bad][:10])
for nm in ("conv.W", "fc.W", "proj.linear1.W"):
    off, size, _, _ = m.layout.tensors[nm]
    print("unforced fp32 oracle vs dev g." + nm, rel(gg[off:off + size], g32[off:off + size]))
