"""SelfAttention (reference layers/self_attention.hpp) through the C ABI against fixtures made by the
real reference layer (tools/make_golden_selfattn.py): output, both weight gradients and the input
delta, plus the softmax launchers against a float64 evaluation."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4  # relative to the tensor's largest magnitude: the fp32 tolerance of the hot path


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["small", "tokens", "ragged"])
def test_self_attention_matches_reference(case):
    import torch
    from vit_amd import SelfAttention
    g = np.load(os.path.join(GOLD, "selfattn_%s.npz" % case))
    x, delta = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["delta"]).cuda()
    B, T, D = g["x"].shape
    sa = SelfAttention(D, g["key.W"].shape[1])
    sa.set_parameters(g["key.W"], g["query.W"])
    out = sa.forward(x)
    assert rel_err(out.cpu().numpy(), g["out"]) < TOL
    sa.backward(delta)
    assert rel_err(sa.g_wk.cpu().numpy(), g["g.key.W"]) < TOL
    assert rel_err(sa.g_wq.cpu().numpy(), g["g.query.W"]) < TOL
    assert rel_err(sa.get_input_deltas().cpu().numpy(), g["dx"]) < TOL
    # DenseLayer::backward assigns its gradient: a second backward leaves the same values
    sa.backward(delta)
    assert rel_err(sa.g_wk.cpu().numpy(), g["g.key.W"]) < TOL
    # the layer has no optimizer of its own in the reference: update_weights leaves the weights alone
    w0 = sa.wk.clone()
    sa.update_weights(32.0)
    assert torch.equal(w0, sa.wk)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,width", [(1, 1), (7, 16), (129, 16), (33, 77), (9, 200), (5, 1000), (6, 1024), (4, 1500)])
def test_softmax_rows_and_backward(rows, width):
    import torch
    from vit_amd._lib import check, lib
    rng = np.random.default_rng(rows * 1000 + width)
    s = (rng.standard_normal((rows, width)) * 4).astype(np.float32)
    da = rng.standard_normal((rows, width)).astype(np.float32)
    st, dt = torch.from_numpy(s).cuda(), torch.from_numpy(da).cuda()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.vt_softmax_rows(C.c_void_p(st.data_ptr()), rows, width, stream), "vt_softmax_rows")
    e = np.exp(s.astype(np.float64) - s.astype(np.float64).max(1, keepdims=True))
    a = e / e.sum(1, keepdims=True)
    assert rel_err(st.cpu().numpy(), a) < 1e-6
    check(lib.vt_softmax_bwd_rows(C.c_void_p(st.data_ptr()), C.c_void_p(dt.data_ptr()), rows, width, stream), "vt_softmax_bwd_rows")
    want = a * (da - (a * da).sum(1, keepdims=True))
    assert rel_err(dt.cpu().numpy(), want) < 1e-5


@pytest.mark.gpu
def test_self_attention_rejects_bad_arguments():
    from vit_amd._lib import lib
    assert lib.vt_self_attention_fwd(None, None, None, None, None, None, None, 1, 1, 1, 1, None) != 0
    assert lib.vt_softmax_rows(None, 1, 1, None) != 0


@pytest.mark.gpu
def test_cpp_layer_mirror_matches_reference(tmp_path):
    """cpp/layers/self_attention.hpp (the reference's Layer surface over the C ABI), driven by
    cpp/tests/self_attention_check.cpp on the generator's inputs: the weights it draws from
    Tensor::global_gen are the reference's bit for bit, results within the fp32 tolerance."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLD), "..", "cpp", "build", "self_attention_check")
    if not os.path.exists(exe):
        pytest.fail("cpp/build/self_attention_check is not built (run __graft_entry__.build())")
    g = np.load(os.path.join(GOLD, "selfattn_small.npz"))
    out = str(tmp_path / "sa.bin")
    subprocess.check_call([exe, "3", "40", "24", "16", "7", out], timeout=120)
    raw = np.fromfile(out, dtype=np.float32)
    pos = 0
    for name, exact in (("key.W", True), ("query.W", True), ("out", False), ("g.key.W", False), ("g.query.W", False), ("dx", False)):
        want = g[name]
        got = raw[pos:pos + want.size].reshape(want.shape)
        pos += want.size
        if exact:
            assert np.array_equal(got, want), name
        else:
            assert rel_err(got, want) < TOL, name
    assert pos == raw.size
