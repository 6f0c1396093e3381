"""Grouped ("virtual rank") training step through the C ABI: n / group independent reference mini-batches in one
launch set, gradients summed, one update (vt_model_train_step_grouped, vt_model_forward_grouped in training mode).

Pins: (1) the REAL reference's per-shard gradients of two 4-image mini-batches from identical weights
(tests/golden/dp_g2_n4_k8.npz, made by oracle/ref_harness.cpp `dp`): their sum and the update with divisor 2 * 32;
(2) the ungrouped step of this library run on each mini-batch alone from the same weights: probabilities, forward
activations bit for bit in the first mini-batch and to 5e-6 in the others (the projector's second moments are
accumulated around the running mean, which the separate steps update in between), running statistics to 1e-6, every
propagated delta to 2e-5 (the partial sums of the BatchNorm backward reductions are cut differently), the summed gradients to rounding of the sums (the grouped step adds all
images' partial gradients in one pass, the separate steps per mini-batch first)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
LOOSE = {"proj.downsample.b": 5e-4, "proj.norm.beta": 5e-4, "conv.b": 5e-4, "bn.beta": 5e-4, "proj.linear3.b": 5e-4}


def _fixture_model(max_batch):
    from vit_amd import VTCNN, Dims
    from oracle.pyoracle import Layout, make_dims
    init = load_golden("step_n2_k8_full.npz")
    lo = Layout(make_dims(num_classes=8))
    arena0 = lo.pack(init, "p0.")
    m = VTCNN(Dims(num_classes=8), max_batch=max_batch, seed=None)
    m.set_arena(arena0)
    return m, lo, arena0


def test_two_groups_of_four_match_the_reference_shard_gradients():
    import torch
    g = load_golden("dp_g2_n4_k8.npz")
    m, lo, arena0 = _fixture_model(8)
    x = torch.from_numpy(np.concatenate([g["in0.X"], g["in1.X"]])).cuda()
    y = torch.from_numpy(np.concatenate([g["in0.labels"], g["in1.labels"]]).astype(np.int32)).cuda()
    p = m.forward_grouped(x, 4).cpu().numpy()
    assert rel_err(p[:4], g["fwd0.preds"]) < TOL and rel_err(p[4:], g["fwd1.preds"]) < TOL
    m.backward(y)
    grads = m.get_grads()
    bad = []
    for name, (off, size, shape, kind) in m.layout.tensors.items():
        if kind != 0:
            continue
        ref = g["g0." + name].astype(np.float64) + g["g1." + name].astype(np.float64)
        e = rel_err(grads[off:off + size], ref)
        if not e < LOOSE.get(name, TOL):
            bad.append((name, e))
    assert not bad, bad
    assert np.all(m.layout.view(grads, "mha.o.W") == 0.0)
    # the whole step in one call from the same start: w <- w - lr * (g0 + g1) / (2 * 32)
    m2, _, _ = _fixture_model(8)
    p2 = m2.train_step_grouped(x, y, 4, 32.0).cpu().numpy()
    assert np.array_equal(p, p2)
    a2 = m2.get_arena()
    for name, (off, size, shape, kind) in m.layout.tensors.items():
        if kind != 0:
            continue
        ref_sum = (g["g0." + name].astype(np.float32) + g["g1." + name].astype(np.float32)).ravel()
        expect = arena0[off:off + size] - (ref_sum * np.float32(1.0 / 64.0)) * np.float32(0.001)
        assert np.abs(a2[off:off + size] - expect).max() <= 1e-6 * max(np.abs(expect).max(), 1e-3), name
    assert not m2.get_grads().any()
    for nm in ("tok.linear1.W", "tok.linear2.W"):
        assert np.array_equal(lo.view(a2, nm), lo.view(arena0, nm))
    # ... and equal to forward_grouped + backward + update_weights(2 * 32) bit for bit
    m.update_weights(64.0)
    assert np.array_equal(m.get_arena(), a2)


PER_GROUP_BUFFERS = ("bwd.d_logits", "bwd.proj_dx", "bwd.proj_dt", "bwd.proj_dv", "bwd.proj_dk", "bwd.tr_ds2", "bwd.tr_da",
                     "bwd.tr_dr", "bwd.tr_dc", "bwd.tok_dT0", "bwd.tok_dx", "bwd.mha_qkv", "fwd.conv_out", "fwd.tok_out",
                     "fwd.tr_out", "fwd.proj_z", "fwd.pooled")


@pytest.mark.parametrize("group,ngroups", [(32, 4), (4, 3), (64, 2), (32, 32)])
def test_grouped_step_equals_separate_mini_batches(group, ngroups):
    import torch
    from vit_amd import VTCNN, Dims
    n = group * ngroups
    rng = np.random.default_rng(1000 * group + ngroups)
    xs = rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)
    ys = rng.integers(0, 8, size=n).astype(np.int32)
    x, y = torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda()
    a = VTCNN(Dims(num_classes=8), max_batch=n, seed=42)    # the grouped step
    b = VTCNN(Dims(num_classes=8), max_batch=group, seed=42)  # one mini-batch at a time, weights never updated in between
    # a few plain steps first so that gamma / beta / running statistics are not at their initial values
    for t in range(2):
        sel = slice(t * group % n, t * group % n + group)
        for m in (a, b):
            m.train_step(x[sel].contiguous(), y[sel].contiguous(), 32.0)
    assert np.array_equal(a.get_arena(), b.get_arena())
    pa = a.forward_grouped(x, group).cpu().numpy().copy()
    a.backward(y)
    ga = a.get_grads().astype(np.float64)
    bufs_a = {k: a.read(k) for k in PER_GROUP_BUFFERS}
    gsum = np.zeros_like(ga)
    full = ngroups <= 4  # every intermediate for the small cases, probabilities and gradients for the large one
    for gi in range(ngroups):
        sel = slice(gi * group, (gi + 1) * group)
        pb = b.forward(x[sel].contiguous()).cpu().numpy()
        # bit for bit in the first mini-batch; later ones accumulate the projector's second moments around a different
        # pivot (the running mean, which the separate steps update in between): rounding of the statistics
        assert np.array_equal(pa[sel], pb) if gi == 0 else rel_err(pa[sel], pb) < 5e-6, gi
        b.backward(y[sel].contiguous())
        gsum += b.get_grads().astype(np.float64)
        if full:
            for k in PER_GROUP_BUFFERS:
                vb = b.read(k)
                va = bufs_a[k]
                if k == "bwd.mha_qkv":  # three [rows, D] matrices one after another
                    va = va.reshape(3, n * 16 * 32)[:, gi * vb.size // 3:(gi + 1) * vb.size // 3].ravel()
                else:
                    va = va[gi * vb.size:(gi + 1) * vb.size]
                if gi == 0 and (k.startswith("fwd.") or k == "bwd.d_logits"):
                    assert np.array_equal(va, vb), (k, gi)
                else:  # the partial sums of the BatchNorm backward reductions are cut differently (chunks of the head
                    # backward depend on the launch's image count): last-bit differences in the coefficients
                    assert rel_err(va, vb) < 2e-5, (k, gi, rel_err(va, vb))
    # running statistics: the groups' updates one after another, in order
    nt = a.layout.n_trainable
    assert rel_err(a.get_arena()[nt:], b.get_arena()[nt:]) < 1e-6
    bad = []
    for name, (off, size, shape, kind) in a.layout.tensors.items():
        if kind != 0:
            continue
        e = rel_err(ga[off:off + size], gsum[off:off + size])
        if not e < 2e-5:
            bad.append((name, e))
    assert not bad, bad


def test_grouped_train_step_refusals_and_single_group():
    import torch
    from vit_amd import VTCNN, Dims
    from vit_amd._lib import VtError
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.integers(0, 256, size=(64, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda()
    y = torch.from_numpy(rng.integers(0, 8, size=64).astype(np.int32)).cuda()
    a, b = (VTCNN(Dims(num_classes=8), max_batch=64, seed=42) for _ in range(2))
    with pytest.raises((VtError, ValueError)):
        a.train_step_grouped(x[:60].contiguous(), y[:60].contiguous(), 32)  # not a multiple of the group
    with pytest.raises(VtError):
        a.train_step_grouped(x[:60].contiguous(), y[:60].contiguous(), 6)  # group not a multiple of 4
    # one group = the plain step
    pa = a.train_step_grouped(x, y, 64, 32.0).clone()
    pb = b.train_step(x, y, 32.0)
    assert torch.equal(pa, pb) and np.array_equal(a.get_arena(), b.get_arena())
    # eval mode keeps the evaluation sweep (BatchNorm2D on the running statistics)
    a.set_training(False)
    b.set_training(False)
    pe = a.forward_grouped(x, 32).clone()
    assert rel_err(pe[:32].cpu().numpy(), b.forward(x[:32].contiguous()).cpu().numpy()) < 1e-5
    with pytest.raises(VtError):
        a.backward(y)  # no backward for an evaluation sweep
