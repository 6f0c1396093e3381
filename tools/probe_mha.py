import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd._lib import lib
N, L, D = 1024, 16, 32
rows = N * L
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(3 * rows * D, device="cuda", generator=g) * 0.5
dc = torch.randn(rows * D, device="cuda", generator=g) * 0.1
tok = torch.randn(rows * D, device="cuda", generator=g)
scratch = torch.zeros(2 * 4 * (2 * D + 3) * rows + 2 * (rows // 64 + 1) * 1536 + L * D, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
def run():
    return lib.vt_mha_bwd_rows(p(qkv), p(dc), p(tok), p(scratch), N, L, None)
for var in [0, 1, 2, 3, 4]:
    os.environ["VT_MHA_VAR"] = str(var)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("var %d: %.1f us" % (var, e0.elapsed_time(e1) * 1000 / 20), flush=True)
