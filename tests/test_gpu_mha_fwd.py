"""MultiHeadAttention::forward through the C ABI (vt_mha_fwd) against fixtures from the real reference
(tools/make_golden_selfattn.py, `ref_harness mha`): the layer's own output (zero), every head's output
as the reference's Tensor / DenseLayer operations give it, and the projection of the concatenated heads."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tokens", "ragged"])
def test_mha_forward_matches_reference(case):
    import torch
    from vit_amd._lib import check, lib
    g = np.load(os.path.join(GOLD, "mha_fwd_%s.npz" % case))
    B, T, D = g["x"].shape
    H = sum(1 for k in g.files if k.startswith("head"))
    hd = D // H
    wqkv = np.concatenate([g["%s%d.W" % (n, h)].ravel() for h in range(H) for n in "qkv"]).astype(np.float32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()  # noqa: E731
    x, w, wo = dev(g["x"]), dev(wqkv), dev(g["o.W"])
    heads, out = torch.empty(B, T, D, device="cuda"), torch.full((B, T, D), 7.0, device="cuda")
    scratch = torch.empty(3 * B * T * hd + B * T * T, device="cuda")
    pv = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.vt_mha_fwd(pv(x), pv(w), pv(wo), pv(heads), pv(out), pv(scratch), scratch.numel(), B, T, D, H, 1, st))
    assert np.array_equal(out.cpu().numpy(), g["out"]) and not g["out"].any()   # the reference's zero output
    hn = heads.cpu().numpy()
    for h in range(H):
        assert rel_err(hn[:, :, h * hd:(h + 1) * hd], g["head%d" % h]) < 1e-4, h
    check(lib.vt_mha_fwd(pv(x), pv(w), pv(wo), pv(heads), pv(out), pv(scratch), scratch.numel(), B, T, D, H, 0, st))
    assert rel_err(out.cpu().numpy(), g["out_concat"]) < 1e-4
    # too little scratch, heads that do not divide the width
    assert lib.vt_mha_fwd(pv(x), pv(w), pv(wo), pv(heads), pv(out), pv(scratch), 16, B, T, D, H, 1, st) != 0
    assert lib.vt_mha_fwd(pv(x), pv(w), pv(wo), pv(heads), pv(out), pv(scratch), scratch.numel(), B, T, D, D + 1, 1, st) != 0
