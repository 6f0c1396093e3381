import sys, numpy as np, torch
sys.path.insert(0, '.')
from vit_amd import VTCNN, Dims
n, k = 32, 8
rng = np.random.default_rng(n)
xs = [torch.from_numpy(rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda() for _ in range(3)]
ys = [torch.from_numpy(rng.integers(0, k, size=n).astype(np.int32)).cuda() for _ in range(3)]
def eager(steps):
    a = VTCNN(Dims(num_classes=k), max_batch=n, seed=42)
    for i in range(steps):
        a.forward(xs[i]); a.backward(ys[i]); a.update_weights(32.0)
    return a
def graph(steps):
    b = VTCNN(Dims(num_classes=k), max_batch=n, seed=42)
    xbuf, ybuf = torch.empty_like(xs[0]), torch.empty_like(ys[0])
    b.capture_step(xbuf, ybuf, 32.0)
    if TEMP: VTCNN(Dims(num_classes=k), max_batch=n, seed=42).get_arena()
    for i in range(steps):
        xbuf.copy_(xs[i]); ybuf.copy_(ys[i]); b.step_graph()
    torch.cuda.synchronize()
    return b
for steps, TEMP in ((3, False), (3, True), (3, True), (3, False)):
    a, b = eager(steps), graph(steps)
    wa, wb = a.get_arena(), b.get_arena()
    for name, (off, size, shape, kind) in a.layout.tensors.items():
        d = np.abs(wa[off:off+size] - wb[off:off+size]).max()
        if d > 0: print(steps, name, d)
    ga, gb = a.get_grads(), b.get_grads()
    print(steps, TEMP, "grads max diff", np.abs(ga-gb).max())
