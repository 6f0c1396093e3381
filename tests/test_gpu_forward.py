"""GPU parity of the forward chain (through the C ABI) against the reference's own tensors
(tests/golden/, produced by the real reference) and against the CPU oracle at larger batches.
Tolerance: 1e-4 relative (max|a-b|/max|b|) per tensor; argmax labels bit-exact."""
import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(k, max_batch, g=None):
    import torch
    from vit_amd import VTCNN, Dims
    from oracle.pyoracle import Layout, make_dims
    m = VTCNN(Dims(num_classes=k), max_batch=max_batch, seed=42)
    if g is not None:
        lo = Layout(make_dims(num_classes=k))
        m.set_arena(lo.pack(g, "p0."))
    m.keep_internals(True)
    return m, torch


def test_init_matches_reference_draw_order():
    g = load_golden("step_n32_k10.npz")
    m, torch = _model(10, 4)
    arena = m.get_arena()
    for name, (off, size, shape, kind) in m.layout.tensors.items():
        assert np.array_equal(arena[off:off + size], g["p0." + name].ravel()), name


@pytest.mark.parametrize("fx,n,k", [("step_n2_k8_full.npz", 2, 8), ("step_n32_k10.npz", 32, 10), ("step_n5_k8.npz", 5, 8)])
def test_forward_vs_reference(fx, n, k):
    g = load_golden(fx)
    if "p0.conv.W" not in g:
        g.update({kk: v for kk, v in load_golden("step_n2_k8_full.npz").items() if kk.startswith("p0.")})
    m, torch = _model(k, 32, g)
    x = torch.from_numpy(g["in.X"]).cuda()
    p = m.forward(x).cpu().numpy()
    if "fwd.conv_out" in g:
        P = 784
        checks = {
            "fwd.conv_out": g["fwd.conv_out"].reshape(n, 8, P).transpose(0, 2, 1),
            "fwd.tok_A": g["fwd.tok_A"], "fwd.tok_out": g["fwd.tok_out"], "fwd.tr_a": g["fwd.tr_a"],
            "fwd.tr_hdn": g["fwd.tr_hdn"], "fwd.tr_out": g["fwd.tr_out"], "fwd.proj_a": g["fwd.proj_a"],
            "fwd.proj_z": g["fwd.proj_z"], "fwd.pooled": g["fwd.pooled"], "fwd.logits": g["fwd.logits"],
        }
        for name, ref in checks.items():
            e = rel_err(m.read(name), ref)
            assert e < TOL, (name, e)
    assert rel_err(p, g["fwd.preds"]) < TOL
    _, am = m.metrics()
    assert np.array_equal(am.cpu().numpy(), g["fwd.argmax"])
    y = torch.from_numpy(g["in.labels"].astype(np.int32)).cuda()
    ls, _ = m.metrics(y)
    assert abs(float(ls.item()) / n - float(g["fwd.loss"][0])) < TOL * abs(float(g["fwd.loss"][0]))
