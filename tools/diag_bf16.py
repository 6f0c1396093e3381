"""bf16 mode at a general shape: per-tensor gradient distance to the reference fixture (full tensors only)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from tests_util import shape_case, STRIDE
from vit_amd import VTCNN, Dims
name = sys.argv[1] if len(sys.argv) > 1 else "wide"
g = load_golden("shape_%s_n2.npz" % name)
dims, n, k, seed, x, y = shape_case(g)
for bf in (0, 1):
    m = VTCNN(Dims(**dims), max_batch=n, seed=42)
    m.set_matmul_precision(bool(bf))
    p = m.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    m.backward(torch.from_numpy(y).cuda())
    gr = m.get_grads()
    print("bf16" if bf else "fp32", "preds err", np.abs(p - g["fwd.preds"]).max() / g["fwd.preds"].max())
    for nm, (off, size, shape, kind) in m.layout.tensors.items():
        if kind != 0 or "g." + nm not in g: continue
        a = gr[off:off + size]; b = g["g." + nm].ravel()
        if b.size != size: a = a[::STRIDE]
        if np.abs(b).max() == 0: continue
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        print("  %-20s rel %.3e cos %.5f" % (nm, np.abs(a - b).max() / np.abs(b).max(), cos))
