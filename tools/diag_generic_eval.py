"""Eval-mode predictions after one update at a general shape: HIP path, reference fixture and the fp64 oracle."""
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

name = sys.argv[1] if len(sys.argv) > 1 else "wide"
g = load_golden("shape_%s_n2.npz" % name)
dims, n, k, seed, x, y = shape_case(g)
m = VTCNN(Dims(**dims), max_batch=n, seed=42)
arena = m.get_arena().copy()
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
p1 = m.forward(xd).cpu().numpy().copy(); m.backward(yd); m.update_weights(32.0)
p2 = m.forward(xd).cpu().numpy().copy(); m.set_training(False); pe = m.forward(xd).cpu().numpy().copy()
for f64 in (True, False):
    o = Oracle(make_dims(**dims), n, f64=f64)
    a = arena.astype(np.float64 if f64 else np.float32)
    q1 = o.forward(a, x, training=True).copy(); gr = o.backward(a, y); o.sgd(a, gr, 0.001, 32.0)
    q2 = o.forward(a, x, training=True).copy(); qe = o.forward(a, x, training=False).copy()
    tag = "fp64" if f64 else "fp32"
    print(tag, "oracle: dev p1 %.2e p2 %.2e eval %.2e | fixture p1 %.2e p2 %.2e eval %.2e" % (
        rel(p1, q1), rel(p2, q2), rel(pe, qe), rel(g["fwd.preds"], q1), rel(g["fwd2.preds"], q2), rel(g["eval.preds"], qe)))
print("eval preds dev", pe[0], "\nfixture", g["eval.preds"][0])
