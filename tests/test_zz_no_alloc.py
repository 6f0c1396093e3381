"""vt_fc_wgrad_static: the allocation-free variant of vt_fc_wgrad (partial slab in a fixed device
array inside the library) must write the same bytes as vt_fc_wgrad with a caller-provided scratch
buffer, on two calls in a row, the second with a smaller input (the array keeps the first call's
contents), and must reject sizes beyond its fixed capacity."""
import ctypes as C

import numpy as np
import pytest


def _ptr(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.gpu
def test_fc_wgrad_static_matches_fc_wgrad_on_two_calls_in_a_row():
    import torch
    from vit_amd._lib import check, lib
    Q = 3136
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(11)
    # second call smaller (fewer 128-image chunks, fewer classes: a narrower slab row inside the same array),
    # third one a single chunk with the bias gradient not adjacent to the weight gradient
    for n, k in ((1024, 10), (300, 8), (77, 8)):
        pooled = torch.from_numpy(rng.standard_normal((n, Q), dtype=np.float32)).cuda()
        dl = torch.from_numpy(rng.standard_normal((n, k), dtype=np.float32)).cuda()
        scratch = torch.full((((n + 127) // 128) * (Q * k + k),), float("nan"), device="cuda")
        gw_a, gb_a = torch.zeros(Q * k, device="cuda"), torch.zeros(k, device="cuda")
        gw_b, gb_b = torch.full_like(gw_a, 7.0), torch.full_like(gb_a, 7.0)
        check(lib.vt_fc_wgrad(_ptr(pooled), _ptr(dl), _ptr(gw_a), _ptr(gb_a), _ptr(scratch), n, Q, k, st), "vt_fc_wgrad")
        check(lib.vt_fc_wgrad_static(_ptr(pooled), _ptr(dl), _ptr(gw_b), _ptr(gb_b), n, Q, k, st), "vt_fc_wgrad_static")
        torch.cuda.synchronize()
        assert torch.equal(gw_a, gw_b), (n, k)
        assert torch.equal(gb_a, gb_b), (n, k)
        # and both are the contraction the reference computes (layers/dense_layer.hpp:92-96)
        want_w = pooled.double().T @ dl.double()
        want_b = dl.double().sum(0)
        assert float((gw_b.view(Q, k).double() - want_w).abs().max() / want_w.abs().max()) < 1e-5
        assert float((gb_b.double() - want_b).abs().max() / want_b.abs().max()) < 1e-5


@pytest.mark.gpu
def test_fc_wgrad_static_rejects_sizes_beyond_its_array():
    import torch
    from vit_amd._lib import lib
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    Q, k, n = 3136, 16, 2049  # 17 chunks of 128 images: one more than the array holds
    pooled = torch.zeros((n, Q), device="cuda")
    dl = torch.zeros((n, k), device="cuda")
    gw, gb = torch.zeros(Q * k, device="cuda"), torch.zeros(k, device="cuda")
    rc = lib.vt_fc_wgrad_static(_ptr(pooled), _ptr(dl), _ptr(gw), _ptr(gb), n, Q, k, st)
    assert rc == -6, rc
    assert b"capacity" in lib.vt_error_string(rc)
    assert lib.vt_fc_wgrad_static(_ptr(pooled), _ptr(dl), _ptr(gw), _ptr(gb), 2048, Q, k, st) == 0
    assert lib.vt_fc_wgrad_static(None, _ptr(dl), _ptr(gw), _ptr(gb), 8, Q, k, st) == -1
    torch.cuda.synchronize()
