"""Is the training step bound by the host's launch rate or by the device?  For each mode prints the
host time to enqueue a step (no synchronisation inside the loop) and the wall time per step.

    PYTHONPATH=. python tools/probe_host_bound.py [batch] [steps]
"""
import sys
import time

import numpy as np
import torch

from vit_amd import VTCNN, Dims

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.random((B, 1, 28, 28), dtype=np.float32)).cuda()
y = torch.from_numpy(rng.integers(0, 8, size=B).astype(np.int32)).cuda()
for mode in ("eager", "eager+deferred", "one-stream", "graph"):
    m = VTCNN(Dims(num_classes=8), max_batch=B, seed=42)
    if mode == "eager+deferred":
        m.set_deferred_attention(True)
    if mode == "one-stream":
        m.set_concurrency(False)
    if mode == "graph":
        m.capture_step(x, y, float(B))

    def step():
        if mode == "graph":
            m.step_graph()
        else:
            m.forward(x); m.backward(y); m.update_weights(float(B))

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-16s batch %d: host enqueue %.1f us/step, wall %.1f us/step" % (mode, B, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6), flush=True)
    del m
