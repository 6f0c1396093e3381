"""Measurement aids of the C ABI: the copy yardstick and the per-launcher event timeline."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_probe_copy_copies(mode):
    import torch
    from vit_amd._lib import check, lib
    blocks, floats = 37, 4 * 333
    src = torch.rand(blocks * floats, device="cuda")
    dst = torch.zeros_like(src)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.vt_probe_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), blocks, floats, mode, st), "vt_probe_copy")
    assert torch.equal(src, dst)
    assert lib.vt_probe_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), blocks, 6, mode, st) != 0  # not a multiple of 4
    assert lib.vt_probe_copy(None, C.c_void_p(dst.data_ptr()), blocks, floats, mode, st) != 0
    assert lib.vt_probe_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), blocks, floats, 4, st) != 0  # no such mode


@pytest.mark.gpu
def test_profile_timeline_orders_the_step():
    """Every launcher of one step appears once per launch with start <= end; the forward comes first,
    the update last, and the two backward chains (two streams) both lie between head_bwd and the reduction."""
    import torch
    from vit_amd import VTCNN, Dims
    n = 64
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.random((n, 1, 28, 28), dtype=np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, 8, size=n).astype(np.int32)).cuda()
    m = VTCNN(Dims(num_classes=8), max_batch=n, seed=42)
    m.forward(x); m.backward(y); m.update_weights(32.0)
    m.profile("*")
    m.forward(x); m.backward(y); m.update_weights(32.0)
    tl = m.profile_timeline()
    m.profile(None)
    names = [t[0] for t in tl]
    assert names.count("bn_stats_finalize") == 4 and names.count("bn_bwd_finalize") == 4
    for one in ("conv3x3_fwd", "tokenizer_fwd", "head_fwd", "head_bwd", "projector_bwd", "mha_bwd_rows", "tokenizer_bwd",
                "conv3x3_wgrad", "fc_wgrad", "reduce_rows", "sgd_update"):
        assert names.count(one) == 1, one
    at = {t[0]: t for t in tl if names.count(t[0]) == 1}
    assert all(t[1] <= t[2] for t in tl)
    assert at["conv3x3_fwd"][1] == min(t[1] for t in tl)
    assert at["sgd_update"][2] == max(t[2] for t in tl)
    for side in ("tokenizer_bwd", "conv3x3_wgrad", "fc_wgrad", "mha_bwd_rows"):
        assert at["head_bwd"][2] <= at[side][1] and at[side][2] <= at["reduce_rows"][2], side
