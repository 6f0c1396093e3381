"""Diagnostic: per-tensor gradient distance of the HIP path and of the fp32 oracle to the fp64
oracle at a given batch size (run on a GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import rel_err
from vit_amd import VTCNN, Dims
from oracle.pyoracle import Oracle, make_dims

n, k = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(n)
x = (rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0))
y = rng.integers(0, k, size=n).astype(np.int32)
m = VTCNN(Dims(num_classes=k), max_batch=n, seed=42)
a32 = m.get_arena().copy()
o64 = Oracle(make_dims(num_classes=k), n, f64=True)
o32 = Oracle(make_dims(num_classes=k), n)
p64 = o64.forward(a32.astype(np.float64), x); g64 = o64.backward(a32.astype(np.float64), y)
p32 = o32.forward(a32.copy(), x); g32 = o32.backward(a32, y)
pg = m.forward(torch.from_numpy(x).cuda()).cpu().numpy(); m.backward(torch.from_numpy(y).cuda()); gg = m.get_grads()
print("preds gpu-f64 %.2e  ref32-f64 %.2e" % (rel_err(pg, p64), rel_err(p32, p64)))
for nm in ["fwd.tok_out", "fwd.tr_out", "fwd.proj_z", "bwd.proj_dx", "bwd.proj_dt", "bwd.tr_ds2", "bwd.tr_dr", "bwd.tok_dx"]:
    print("%-14s gpu-f64 %.2e  ref32-f64 %.2e" % (nm, rel_err(m.read(nm), o64.get(nm)), rel_err(o32.get(nm), o64.get(nm))))
for name, (off, size, shape, kind) in m.layout.tensors.items():
    if kind == 0 and name != "mha.o.W":
        print("%-20s gpu-f64 %.2e  ref32-f64 %.2e" % (name, rel_err(gg[off:off+size], g64[off:off+size]), rel_err(g32[off:off+size], g64[off:off+size])))
