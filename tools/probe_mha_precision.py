"""Attention-backward sweep: time per launch with two-piece / three-piece operands and how large the
score bound is for the bench's own data (seed-42 weights, noise images).

    PYTHONPATH=. VT_MHA_PRECISION=two|three python tools/probe_mha_precision.py [batch]
"""
import sys

import numpy as np
import torch

from vit_amd import VTCNN, Dims

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(123)
xs = torch.from_numpy(rng.integers(0, 256, size=(4, B, 1, 28, 28), dtype=np.uint8).astype(np.float32) / np.float32(255.0)).cuda()
ys = torch.from_numpy(rng.integers(0, 8, size=(4, B)).astype(np.int32)).cuda()
m = VTCNN(Dims(num_classes=8), max_batch=B, seed=42)
for b in range(4):
    x, y = xs[b], ys[b]
    for _ in range(3):
        m.forward(x); m.backward(y)
    m.profile("mha_bwd_rows")
    for _ in range(20):
        m.forward(x); m.backward(y)
    ms, cnt = m.profile_read()["mha_bwd_rows"]
    m.profile(None)
    qkv = m.read("bwd.mha_qkv").reshape(3, B * 16, 32)
    q, k = qkv[0], qkv[1]
    out = []
    for h in range(2):
        qn = np.sqrt((q[:, 16 * h:16 * h + 16] ** 2).sum(1)).reshape(16, B)
        kn = np.sqrt((k[:, 16 * h:16 * h + 16] ** 2).sum(1)).reshape(16, B)
        bound = 0.25 * 1.4427 * kn.max(1) * qn.max(1)
        out.append("head %d bound %.2f..%.2f" % (h, bound.min(), bound.max()))
    print("batch %d: %s; mha_bwd_rows %.1f us per launch (event interval)" % (b, "; ".join(out), ms / cnt * 1e3), flush=True)
