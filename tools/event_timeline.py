"""Event timeline of one training step without a profiler attached: start / end of every launcher
from the library's own HIP events (vt_model_profile_timeline), for both launch orders of the backward.

    PYTHONPATH=. python tools/event_timeline.py [batch]
"""
import sys

import numpy as np
import torch

from vit_amd import VTCNN, Dims

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.random((B, 1, 28, 28), dtype=np.float32)).cuda()
y = torch.from_numpy(rng.integers(0, 8, size=B).astype(np.int32)).cuda()
for conc in (0, 1):
    m = VTCNN(Dims(num_classes=8), max_batch=B, seed=42)
    m.set_concurrency(bool(conc))
    for _ in range(5):
        m.forward(x); m.backward(y); m.update_weights(float(B))
    torch.cuda.synchronize()
    m.profile("*")
    m.forward(x); m.backward(y); m.update_weights(float(B))
    tl = m.profile_timeline()
    print("two-stream backward: %d" % conc)
    for name, a, b in sorted(tl, key=lambda r: r[1]):
        print("  %8.1f +%6.1f  %s" % (a, b - a, name))
    print("  step end %.1f us" % max(r[2] for r in tl))
