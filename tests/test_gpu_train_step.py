"""GPU parity of one full training step (forward, backward, SGD) through the C ABI against the
reference's own tensors (tests/golden/) and, at larger batches, against the CPU oracle.

Tolerances (max|a-b|/max|b| per tensor): 1e-4 for activations, deltas and gradients; the few
gradients that are plain sums over N*784 rows get 5e-4 because the reference's own sequential
fp32 sum carries ~2e-4 of rounding noise there (measured against an fp64 restatement).
Updated weights: 1e-6.  Frozen tokenizer tensors and the zero attention output-projection
gradient: exact."""
import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
LOOSE = {"proj.downsample.b": 5e-4, "proj.norm.beta": 5e-4, "conv.b": 5e-4, "bn.beta": 5e-4, "proj.linear3.b": 5e-4}


def _setup(fx, k):
    import torch
    from vit_amd import VTCNN, Dims
    from oracle.pyoracle import Layout, make_dims
    g = load_golden(fx)
    if "p0.conv.W" not in g:
        g.update({kk: v for kk, v in load_golden("step_n2_k8_full.npz").items() if kk.startswith("p0.")})
    m = VTCNN(Dims(num_classes=k), max_batch=32, seed=None)
    lo = Layout(make_dims(num_classes=k))
    m.set_arena(lo.pack(g, "p0."))
    x = torch.from_numpy(g["in.X"]).cuda()
    y = torch.from_numpy(g["in.labels"].astype(np.int32)).cuda()
    return g, m, x, y, torch


def _check_grads(m, g):
    grads = m.get_grads()
    bad = []
    for name, (off, size, shape, kind) in m.layout.tensors.items():
        if kind != 0:
            continue
        e = rel_err(grads[off:off + size], g["g." + name])
        if not e < LOOSE.get(name, TOL):
            bad.append((name, e))
    assert not bad, bad
    assert np.all(m.layout.view(grads, "mha.o.W") == 0.0)


def test_backward_intermediates_n2():
    g, m, x, y, torch = _setup("step_n2_k8_full.npz", 8)
    m.forward(x)
    m.backward(y)
    n, P = 2, 784
    checks = {
        "bwd.d_logits": g["bwd.d_logits"],
        "bwd.proj_dx": g["bwd.proj_dx"], "bwd.proj_dt": g["bwd.proj_dt"], "bwd.tr_ds2": g["bwd.tr_ds2"],
        "bwd.tr_dr": g["bwd.tr_dr"], "bwd.tok_dx": g["bwd.tok_dx"],
    }
    bad = []
    for name, ref in checks.items():
        e = rel_err(m.read(name), ref)
        if not e < TOL:
            bad.append((name, e))
    assert not bad, bad
    _check_grads(m, g)


@pytest.mark.parametrize("fx,n,k", [("step_n2_k8_full.npz", 2, 8), ("step_n32_k10.npz", 32, 10), ("step_n5_k8.npz", 5, 8)])
def test_train_step_vs_reference(fx, n, k):
    g, m, x, y, torch = _setup(fx, k)
    p = m.forward(x).cpu().numpy()
    assert rel_err(p, g["fwd.preds"]) < TOL
    m.backward(y)
    _check_grads(m, g)
    m.update_weights(32.0)
    arena = m.get_arena()
    for name, (off, size, shape, kind) in m.layout.tensors.items():
        if "p1." + name not in g:
            continue
        ref = g["p1." + name].ravel().astype(np.float64)
        err = np.abs(arena[off:off + size].astype(np.float64) - ref).max()
        if kind == 0:
            # w1 = w0 - g/32*lr: fp32 storage rounding of w plus the gradient's own tolerance
            bound = 1e-6 * np.abs(ref).max() + (0.001 / 32.0) * LOOSE.get(name, TOL) * np.abs(g["g." + name]).max()
        elif kind == 2:
            bound = 1e-5 * np.abs(ref).max() + 1e-6
        else:
            bound = 0.0
        assert err <= bound, (name, err, bound)
    for nm in ("tok.linear1.W", "tok.linear1.b", "tok.linear2.W", "tok.linear2.b"):
        assert np.array_equal(m.layout.view(arena, nm), g["p0." + nm].reshape(m.layout.tensors[nm][2])), nm
    assert np.all(m.get_grads() == 0.0)  # zero_grad
    # second training forward with the updated weights, then eval-mode forward
    p2 = m.forward(x).cpu().numpy()
    assert rel_err(p2, g["fwd2.preds"]) < TOL
    m.set_training(False)
    pe = m.forward(x).cpu().numpy()
    assert rel_err(pe, g["eval.preds"]) < TOL
    _, am = m.metrics()
    assert np.array_equal(am.cpu().numpy(), g["eval.argmax"])


def test_twelve_step_trajectory_vs_reference():
    """Twelve consecutive training steps from the seed-42 initialisation on the batches the real
    reference was given (fixture from `ref_harness traj`): per-step loss within 1e-4, every arg-max
    label identical, and the weight CHANGE over the run (the weights themselves barely move at
    lr = 0.001) within 1e-2 of the reference's change per tensor.  That bound is the fixture's own
    uncertainty: the float64 build of the CPU restatement (exact arithmetic, same batches) ends 1.7e-3
    (conv.W) to 4.7e-3 (proj.linear3.W) of the change away from the reference's fp32 result, and this
    path lands at the same distances (1.6e-3, 5.6e-3): twelve steps amplify the reference's rounding."""
    import torch
    from vit_amd import VTCNN, Dims
    g = load_golden("traj_n32_k8_t12.npz")
    m = VTCNN(Dims(num_classes=8), max_batch=32, seed=42)
    p0 = m.get_arena().copy()
    for t in range(12):
        x = torch.from_numpy(g["traj.x_u8"][t].astype(np.float32) / np.float32(255.0)).cuda()
        y = torch.from_numpy(g["traj.y"][t]).cuda()
        m.forward(x)
        loss, am = m.metrics(y)
        assert abs(float(loss) / 32.0 - float(g["traj.loss"][t])) < TOL * abs(float(g["traj.loss"][t])), t
        assert np.array_equal(am.cpu().numpy(), g["traj.argmax"][t]), t
        m.backward(y)
        m.update_weights(32.0)
    pT = m.get_arena()
    bad = []
    for name, (off, size, shape, kind) in m.layout.tensors.items():
        ref = g["pT." + name].ravel().astype(np.float64)
        if kind == 1:
            assert np.array_equal(pT[off:off + size], p0[off:off + size]), name  # frozen tokenizer
            continue
        moved = np.abs(ref - p0[off:off + size]).max()
        err = np.abs(pT[off:off + size] - ref).max()
        if moved > 0 and not err <= 1e-2 * moved + 2e-6 * np.abs(ref).max():
            bad.append((name, err, moved))
    assert not bad, bad


def _routing_from_gpu(m, n):
    """Max-pool slots and the effective ReLU mask the HIP path used in its last backward."""
    slots = m.read("fwd.pool_idx").view(np.uint8)[:n * 3136].copy()
    mask = (m.read("bwd.du") != 0.0).astype(np.uint8)  # only positions that received a delta matter
    return slots, mask


@pytest.mark.parametrize("n,k", [(64, 8), (200, 10), (1024, 8)])
def test_train_step_vs_oracle_larger_batch(n, k):
    _step_vs_f64_oracle(n, k, 1, 28)


@pytest.mark.parametrize("n,k,ci,h", [(33, 5, 3, 28), (40, 8, 1, 14), (16, 16, 4, 14), (7, 2, 2, 28), (1, 8, 1, 28), (3, 10, 1, 14)])
def test_train_step_other_inputs_on_the_fused_kernels(n, k, ci, h):
    """The fused kernels also serve the reference's hyper-parameters with colour inputs (up to four
    image channels) and 14x14 images (pool kernel 1: every pixel its own window), up to sixteen classes:
    same two-part check against the float64 restatement as for the larger batches."""
    _step_vs_f64_oracle(n, k, ci, h)


@pytest.mark.parametrize("n", [32, 1024])
@pytest.mark.parametrize("factor", [8.0, 32.0])
def test_attention_backward_with_large_query_key_weights(n, factor):
    """The attention backward shifts its softmax by a Cauchy-Schwarz bound on the scores and forms its
    products from bf16 hi / lo halves; both are only ever exercised at initialisation scale by the other
    tests.  Here the query and key weights are 8x and 32x larger (scores 64x / 1024x: saturated softmax
    rows, bound far above the row maximum, heavy cancellation in the softmax backward): every gradient,
    the attention block's included, must still agree with exact arithmetic, and nothing may be NaN.
    At 32x the scores reach the thousands and fp32 itself runs out: the reference's own arithmetic (the
    fp32 oracle) is 1.4e-4 (mha.q0.W) and 2.1e-4 (mha.k0.W) away from exact arithmetic at N = 32, so the
    query / key gradients get 3e-4 there; everything else, and everything at 8x, stays at 1e-4."""
    loose = {"mha.%s%d.W" % (t, h): 3e-4 for t in "qk" for h in (0, 1)} if factor > 8 else {}
    _step_vs_f64_oracle(n, 8, 1, 28, qk_factor=factor, loose=loose)


@pytest.mark.parametrize("n", [32, 1024])
def test_train_step_vs_oracle_after_200_training_steps(n):
    """The same two-part check from weights that have been trained (200 SGD steps at a learning rate 5x
    the reference's, on a fixed set of batches, so that every tensor has left its initialisation scale)
    instead of the seed-42 draw.  Training conditions the problem worse (some gradient tensors shrink by
    orders of magnitude next to the activations they are differences of), and the reference's own fp32
    arithmetic then misses exact arithmetic by more than 1e-4 on them; each gradient tensor is therefore
    held to 1e-4 or to three times the distance of the fp32 oracle (the reference's arithmetic, same
    decisions imposed) from the fp64 oracle, whichever is larger; that distance is taken as the largest over four
    batches on the same weights, because a single batch is one noisy draw of it."""
    _step_vs_f64_oracle(n, 8, 1, 28, pretrain_steps=200)


def _step_vs_f64_oracle(n, k, ci, h, qk_factor=1.0, loose=None, pretrain_steps=0):
    """Batches beyond the committed reference fixtures, up to the benchmark's 1024.

    Checker: the oracle (pinned bit-exactly to the reference at N=2/5/32) in its double-precision
    build, because by N=200 the reference's own sequential fp32 sums over N*784 rows are ~2e-3
    away from exact math (tools/diag_errors.py prints both distances).
    The step has two discontinuous operations, the max-pool arg-max and the ReLU mask; two correct
    evaluations may pick differently at near-ties, after which deltas differ by O(1) at those
    pixels.  So parity is checked in two parts:
      1. discrete decisions: every pooling window where the HIP path and the oracle disagree must
         be a near-tie (gap between the two largest window values < 1e-5 of the largest pooled
         value), and such windows must be rare (< 0.1 %); predicted labels are identical;
      2. floats: with the HIP path's own decisions imposed on the fp64 oracle, every gradient
         tensor is within 1e-4 (max|a-b|/max|b|); the forward needs no such care (continuous).
    Exercises multi-tile attention groups (N > 64), ragged row blocks, the chunked fc gradient."""
    import torch
    from vit_amd import VTCNN, Dims
    from oracle.pyoracle import Oracle, make_dims
    rng = np.random.default_rng(n)
    x = (rng.integers(0, 256, size=(n, ci, h, h)).astype(np.float32) / np.float32(255.0))
    y = rng.integers(0, k, size=n).astype(np.int32)
    m = VTCNN(Dims(num_classes=k, image_channels=ci, input_dim=h), max_batch=n, seed=42)
    arena32 = m.get_arena().copy()
    if qk_factor != 1.0:
        for name in m.layout.tensors:
            if name.startswith("mha.q") or name.startswith("mha.k"):
                m.layout.view(arena32, name)[...] *= np.float32(qk_factor)
        m.set_arena(arena32)
    if pretrain_steps:
        m.lr = 0.005
        tx = [torch.from_numpy(rng.integers(0, 256, size=(32, ci, h, h)).astype(np.float32) / np.float32(255.0)).cuda() for _ in range(8)]
        ty = [torch.from_numpy(rng.integers(0, k, size=32).astype(np.int32)).cuda() for _ in range(8)]
        for t in range(pretrain_steps):
            m.forward(tx[t % 8]); m.backward(ty[t % 8]); m.update_weights(32.0)
        m.lr = 0.001
        arena32 = m.get_arena().copy()
        assert np.all(np.isfinite(arena32))
        moved = np.abs(m.layout.view(arena32, "mha.q0.W") - m.layout.view(VTCNN(Dims(num_classes=k, image_channels=ci, input_dim=h), max_batch=1, seed=42).get_arena(), "mha.q0.W")).max()
        assert moved > 1e-3, moved
    pg = m.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    m.backward(torch.from_numpy(y).cuda())
    gg = m.get_grads()
    assert np.all(np.isfinite(gg))
    slots, mask = _routing_from_gpu(m, n)
    inter = {name: m.read(name) for name in ("bwd.proj_dx", "bwd.proj_dt", "bwd.tr_ds2", "bwd.tr_dr", "bwd.tok_dx")}
    # part 1: forward floats and discrete decisions vs exact math (fp64 oracle, its own decisions)
    o = Oracle(make_dims(num_classes=k, image_channels=ci, input_dim=h), n, f64=True)
    arena64 = arena32.astype(np.float64)
    p64 = o.forward(arena64.copy(), x, training=True)
    assert rel_err(pg, p64) < TOL
    top2 = np.sort(p64, axis=1)[:, -2:]
    flips = np.nonzero(pg.argmax(1) != p64.argmax(1))[0]
    assert np.all(top2[flips, 1] - top2[flips, 0] < 1e-5), flips  # labels equal except exact-tie samples
    ref_slots, gap = o.pool_decisions()
    diff = np.nonzero(ref_slots != slots)[0]
    assert diff.size <= 1e-3 * slots.size, diff.size
    assert np.all(gap[diff] < 1e-5 * np.abs(o.get("fwd.pooled")).max()), gap[diff].max()
    # part 2: floats vs exact math under identical decisions
    o.set_routing(slots, mask)
    po = o.forward(arena64.copy(), x, training=True)
    go = o.backward(arena64, y)
    assert rel_err(pg, po) < TOL
    if pretrain_steps:  # how far the reference's own arithmetic is from exact arithmetic on these weights
        o32 = Oracle(make_dims(num_classes=k, image_channels=ci, input_dim=h), n)
        o32.set_routing(slots, mask)
        o32.forward(arena32.copy(), x, training=True)
        g32 = o32.backward(arena32, y)
        loose = dict(loose or {})
        for name, (off, size, shape, kind) in m.layout.tensors.items():
            if kind == 0:
                loose[name] = max(TOL, 3.0 * rel_err(g32[off:off + size], go[off:off + size]))
        # One batch gives one draw of that distance, and on an ill-conditioned tensor (a gradient that is a small
        # difference of large sums) draws scatter by an order of magnitude: three more batches on the same weights,
        # each with the decisions the HIP path takes on it, and the largest distance seen is the yardstick.
        o64b = Oracle(make_dims(num_classes=k, image_channels=ci, input_dim=h), n, f64=True)
        for _ in range(3):
            xe = (rng.integers(0, 256, size=(n, ci, h, h)).astype(np.float32) / np.float32(255.0))
            ye = rng.integers(0, k, size=n).astype(np.int32)
            m.set_arena(arena32)
            m.forward(torch.from_numpy(xe).cuda())
            m.backward(torch.from_numpy(ye).cuda())
            se, me = _routing_from_gpu(m, n)
            for oo, ar in ((o32, arena32), (o64b, arena64)):
                oo.set_routing(se, me)
                oo.forward(ar.copy(), xe, training=True)
            g32e, g64e = o32.backward(arena32, ye), o64b.backward(arena64, ye)
            for name, (off, size, shape, kind) in m.layout.tensors.items():
                if kind == 0:
                    loose[name] = max(loose[name], 3.0 * rel_err(g32e[off:off + size], g64e[off:off + size]))
    bad = []
    for name in inter:
        e = rel_err(inter[name], o.get(name))
        if not e < TOL:
            bad.append((name, e))
    for name, (off, size, shape, kind) in m.layout.tensors.items():
        if kind == 0:
            e = rel_err(gg[off:off + size], go[off:off + size])
            if not e < (loose or {}).get(name, TOL):
                bad.append((name, e))
    assert not bad, bad


def test_checkpoint_v2_resume_is_bit_exact(tmp_path):
    """Train 2 steps, save v2, train a 3rd; a fresh model that loads the v2 file and takes the same
    3rd step ends with a bit-identical arena (weights, transformer block, all running statistics).
    The same through a v1 file differs: v1 drops the transformer block (model/vit_cnn.hpp:276-394)."""
    import torch
    from vit_amd import VTCNN, Dims
    n, k = 32, 8
    rng = np.random.default_rng(77)
    xs = [torch.from_numpy(rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda()
          for _ in range(3)]
    ys = [torch.from_numpy(rng.integers(0, k, size=n).astype(np.int32)).cuda() for _ in range(3)]

    def step(m, i):
        m.forward(xs[i])
        m.backward(ys[i])
        m.update_weights(32.0)

    a = VTCNN(Dims(num_classes=k), max_batch=n, seed=42)
    step(a, 0)
    step(a, 1)
    a.save_v2(tmp_path / "s2.bin", epoch=0, step=2, loader_resets=1)
    a.save(tmp_path / "s2_v1.bin")
    step(a, 2)
    want = a.get_arena().copy()

    b = VTCNN(Dims(num_classes=k), max_batch=n, seed=42)
    st = b.load_v2(tmp_path / "s2.bin")
    assert st == {"version": 2, "epoch": 0, "step": 2, "loader_seed": 42, "loader_resets": 1}
    step(b, 2)
    assert np.array_equal(b.get_arena(), want)

    c = VTCNN(Dims(num_classes=k), max_batch=n, seed=42)
    assert c.load_v2(tmp_path / "s2_v1.bin")["version"] == 1
    step(c, 2)
    nt = c.layout.n_trainable
    assert not np.array_equal(c.get_arena()[:nt], want[:nt])


@pytest.mark.parametrize("n", [32, 200])
def test_graph_step_is_bit_identical_to_eager(n):
    """The HIP-graph replay of a training step (vt_model_graph_capture / launch) issues the same
    kernels in the same order as the eager calls: three replayed steps on refilled buffers must
    leave a bit-identical arena."""
    import torch
    from vit_amd import VTCNN, Dims
    k = 8
    rng = np.random.default_rng(n)
    xs = [torch.from_numpy(rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda()
          for _ in range(3)]
    ys = [torch.from_numpy(rng.integers(0, k, size=n).astype(np.int32)).cuda() for _ in range(3)]
    a = VTCNN(Dims(num_classes=k), max_batch=n, seed=42)
    for i in range(3):
        a.forward(xs[i])
        a.backward(ys[i])
        a.update_weights(32.0)
    want = a.get_arena().copy()
    b = VTCNN(Dims(num_classes=k), max_batch=n, seed=42)
    xbuf, ybuf = torch.empty_like(xs[0]), torch.empty_like(ys[0])
    b.capture_step(xbuf, ybuf, 32.0)
    assert np.array_equal(b.get_arena(), VTCNN(Dims(num_classes=k), max_batch=n, seed=42).get_arena())  # capture runs nothing
    for i in range(3):
        xbuf.copy_(xs[i])
        ybuf.copy_(ys[i])
        b.step_graph()
    torch.cuda.synchronize()
    assert np.array_equal(b.get_arena(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [32, 200])
def test_two_call_backward_and_overlapped_update_are_bit_identical(n):
    """vt_model_backward_main + vt_model_backward_attention write exactly the bytes of
    vt_model_backward, and the data-parallel step that lets the attention block's update trail into
    the next step (vit_amd/parallel.py, here with a world of one: same order of device work, no
    collective) ends on exactly the weights of plain steps."""
    import torch
    from vit_amd import VTCNN, Dims
    from vit_amd.parallel import DataParallel
    rng = np.random.default_rng(7 * n)
    xs = [torch.from_numpy(rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda()
          for _ in range(3)]
    ys = [torch.from_numpy(rng.integers(0, 8, size=n).astype(np.int32)).cuda() for _ in range(3)]
    a, b = (VTCNN(Dims(num_classes=8), max_batch=n, seed=42) for _ in range(2))
    a.forward(xs[0]); a.backward(ys[0])
    b.forward(xs[0]); b.backward_main(ys[0])
    first, count = b.attention_range()
    assert count == 6 * 32 * 16 and first % 4 == 0
    gb = b.get_grads()
    assert not gb[first:first + count].any()           # untouched by the first call
    b.backward_attention()
    assert np.array_equal(a.get_grads(), b.get_grads())
    with pytest.raises(RuntimeError):
        b.backward_attention()                          # nothing pending any more
    a.update_weights(float(n)); b.update_weights(float(n))
    dp = DataParallel(b, world=1, rank=0, overlap=True)
    if n == 200:
        # the same through RCCL: a one-rank group still launches its all-reduce kernels on the
        # collective stream, so the stream ordering between exchange and update is the real one
        import os
        import socket
        import torch.distributed as dist
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        dp = DataParallel(b, overlap=True)
        dp.always_exchange = True
    try:
        _overlapped_steps(a, b, dp, xs, ys, n, first, count)
    finally:
        if n == 200:
            dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [32, 1024])
def test_two_stream_backward_is_bit_identical_to_one_stream(n):
    """vt_model_set_concurrency: the tokenizer / BatchNorm2D / conv chain on the model's own stream
    beside the transformer chain writes exactly the gradients and weights of the one-stream order,
    step after step (a stale or racing buffer would show up from the second step on), also when the
    step is replayed as a captured graph."""
    import torch
    from vit_amd import VTCNN, Dims
    rng = np.random.default_rng(11 * n)
    xs = [torch.from_numpy(rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda()
          for _ in range(4)]
    ys = [torch.from_numpy(rng.integers(0, 8, size=n).astype(np.int32)).cuda() for _ in range(4)]
    a, b, c = (VTCNN(Dims(num_classes=8), max_batch=n, seed=42) for _ in range(3))
    a.set_concurrency(False)
    b.set_concurrency(True)
    c.set_concurrency(True)
    xbuf, ybuf = torch.empty_like(xs[0]), torch.empty_like(ys[0])
    c.capture_step(xbuf, ybuf, float(n))
    for t in range(4):
        a.forward(xs[t]); a.backward(ys[t])
        b.forward(xs[t]); b.backward(ys[t])
        assert np.array_equal(a.get_grads(), b.get_grads()), t
        a.update_weights(float(n)); b.update_weights(float(n))
        xbuf.copy_(xs[t]); ybuf.copy_(ys[t])
        c.step_graph()
        wa = a.get_arena()
        assert np.array_equal(wa, b.get_arena()), t
        assert np.array_equal(wa, c.get_arena()), t


@pytest.mark.parametrize("n", [32, 256])
def test_deferred_attention_backward_is_bit_identical(n):
    """vt_model_set_deferred_attention: the attention block's backward and its weight update trail on the
    model's own stream into the next forward.  Weights after every step, the gradients read between
    backward and update, an eval forward in between and varying batch sizes all give exactly the bytes of
    the in-order step (a token buffer overwritten too early, a missed join or an update landing late
    would show from the second step on)."""
    import torch
    from vit_amd import VTCNN, Dims
    rng = np.random.default_rng(7 * n)
    sizes = [n, n, n // 2, n, n]
    xs = [torch.from_numpy(rng.integers(0, 256, size=(k, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda() for k in sizes]
    ys = [torch.from_numpy(rng.integers(0, 8, size=k).astype(np.int32)).cuda() for k in sizes]
    a, b = (VTCNN(Dims(num_classes=8), max_batch=n, seed=42) for _ in range(2))
    b.set_deferred_attention(True)
    first, count = a.attention_range()
    for t in range(len(sizes)):
        pa = a.forward(xs[t]).clone(); a.backward(ys[t])
        pb = b.forward(xs[t]).clone(); b.backward(ys[t])
        assert torch.equal(pa, pb), t
        if t == 1:  # gradients read while the deferred block is in flight: complete and identical
            ga, gb = a.get_grads(), b.get_grads()
            assert np.abs(ga[first:first + count]).max() > 0
            assert np.array_equal(ga, gb), t
        a.update_weights(float(n)); b.update_weights(float(n))
        if t == 2:  # two forwards in a row (evaluation) while the update is still trailing
            for m in (a, b):
                m.set_training(False)
            ea = a.forward(xs[0]).clone(); a.forward(xs[1])
            eb = b.forward(xs[0]).clone(); b.forward(xs[1])
            assert torch.equal(ea, eb)
            for m in (a, b):
                m.set_training(True)
        if t != 3:  # step 3 -> 4 runs without any host read in between
            assert np.array_equal(a.get_arena(), b.get_arena()), t
    assert np.array_equal(a.get_arena(), b.get_arena())
    assert np.array_equal(a.get_grads(), b.get_grads())
    b.set_deferred_attention(False)
    a.forward(xs[0]); a.backward(ys[0]); a.update_weights(float(n))
    b.forward(xs[0]); b.backward(ys[0]); b.update_weights(float(n))
    assert np.array_equal(a.get_arena(), b.get_arena())


def test_graph_replay_waits_for_a_deferred_attention_backward():
    """vt_model_graph_launch after an eager step whose attention backward was deferred: the replay overwrites the
    buffers, gradients and attention weights that deferred work still reads and writes, so it must wait for it.
    Eager deferred step, then two graph replays, against the same three steps in order on a plain model."""
    import torch
    from vit_amd import VTCNN, Dims
    n = 256
    rng = np.random.default_rng(99)
    xs = [torch.from_numpy(rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda() for _ in range(3)]
    ys = [torch.from_numpy(rng.integers(0, 8, size=n).astype(np.int32)).cuda() for _ in range(3)]
    a, b = (VTCNN(Dims(num_classes=8), max_batch=n, seed=42) for _ in range(2))
    xbuf, ybuf = torch.empty_like(xs[0]), torch.empty_like(ys[0])
    b.capture_step(xbuf, ybuf, float(n))
    b.set_deferred_attention(True)
    for rep in range(2):
        a.forward(xs[0]); a.backward(ys[0]); a.update_weights(float(n))
        b.forward(xs[0]); b.backward(ys[0]); b.update_weights(float(n))  # attention block trails on its own stream
        for t in (1, 2):
            a.forward(xs[t]); a.backward(ys[t]); a.update_weights(float(n))
            xbuf.copy_(xs[t]); ybuf.copy_(ys[t])
            b.step_graph()
        assert np.array_equal(a.get_arena(), b.get_arena()), rep


def test_batch_size_may_change_from_step_to_step():
    """One model (max_batch 96) stepping through batches of 96, 5, 33, 1, 64, 96 images (the short last
    batch of an epoch, then full ones again) computes, at every step, exactly what a model created for that
    batch size computes from the same weights: nothing in the workspace survives from a larger batch."""
    import torch
    from vit_amd import VTCNN, Dims
    rng = np.random.default_rng(5)
    a = VTCNN(Dims(num_classes=8), max_batch=96, seed=42)
    for t, n in enumerate([96, 5, 33, 1, 64, 96]):
        x = torch.from_numpy(rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda()
        y = torch.from_numpy(rng.integers(0, 8, size=n).astype(np.int32)).cuda()
        b = VTCNN(Dims(num_classes=8), max_batch=n, seed=None)
        b.set_arena(a.get_arena())
        pa = a.forward(x).cpu().numpy().copy()
        pb = b.forward(x).cpu().numpy()
        assert np.array_equal(pa, pb), (t, n)
        a.backward(y); b.backward(y)
        assert np.array_equal(a.get_grads(), b.get_grads()), (t, n)
        a.update_weights(32.0); b.update_weights(32.0)
        assert np.array_equal(a.get_arena(), b.get_arena()), (t, n)
        # evaluation in between, as Trainer::evaluate does after an epoch
        a.set_training(False); b.set_training(False)
        assert np.array_equal(a.forward(x).cpu().numpy(), b.forward(x).cpu().numpy()), (t, n)
        a.set_training(True)


def _overlapped_steps(a, b, dp, xs, ys, n, first, count):
    for t in range(3):
        a.forward(xs[t]); a.backward(ys[t]); a.update_weights(float(n))
        dp.train_step(xs[t], ys[t], n)
        wa, wb = a.get_arena(), b.get_arena()
        lag = np.flatnonzero(wa != wb)
        assert lag.size > 0 and lag.min() >= first and lag.max() < first + count
    dp.finish()
    assert np.array_equal(a.get_arena(), b.get_arena())
    assert not b.get_grads().any()


@pytest.mark.gpu
def test_adam_step_matches_textbook_recurrences():
    """vt_model_adam_step (extension; the reference has SGD only): three steps against numpy's
    m <- b1 m + (1-b1) g, v <- b2 v + (1-b2) g^2, w <- w - lr (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)
    on the gradients the model itself produced, g scaled by 1/32 first."""
    import torch
    from vit_amd import VTCNN, Dims
    n = 16
    rng = np.random.default_rng(3)
    m = VTCNN(Dims(num_classes=8), max_batch=n, seed=42)
    nt = m.layout.n_trainable
    w = m.get_arena()[:nt].astype(np.float64)
    m1, m2 = np.zeros(nt), np.zeros(nt)
    for t in range(1, 4):
        x = torch.from_numpy(rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda()
        y = torch.from_numpy(rng.integers(0, 8, size=n).astype(np.int32)).cuda()
        m.forward(x)
        m.backward(y)
        g = m.get_grads().astype(np.float64) / 32.0
        m.update_weights_adam(32.0, lr=1e-3)
        m1 = 0.9 * m1 + 0.1 * g
        m2 = 0.999 * m2 + 0.001 * g * g
        w = w - 1e-3 * (m1 / (1 - 0.9 ** t)) / (np.sqrt(m2 / (1 - 0.999 ** t)) + 1e-8)
        got = m.get_arena()[:nt]
        assert rel_err(got, w) < 1e-5, t
        assert not m.get_grads().any()
        w = got.astype(np.float64)  # follow the device weights so that only one step's rounding is compared


@pytest.mark.parametrize("n", [32, 1024])
def test_train_step_entry_point_is_bit_identical_to_three_calls(n):
    """vt_model_train_step (forward + backward + SGD, the update applied inside the launch that finishes the weight
    gradient reductions) against vt_model_forward / vt_model_backward / vt_model_sgd_step: same probabilities, same
    arena and an all-zero gradient arena after each of three steps."""
    import torch
    from vit_amd import VTCNN, Dims
    rng = np.random.default_rng(7 + n)
    xs = [torch.from_numpy(rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda() for _ in range(3)]
    ys = [torch.from_numpy(rng.integers(0, 8, size=n).astype(np.int32)).cuda() for _ in range(3)]
    a, b = (VTCNN(Dims(num_classes=8), max_batch=n, seed=42) for _ in range(2))
    for x, y in zip(xs, ys):
        pa = a.forward(x).clone()
        a.backward(y)
        a.update_weights(32.0)
        pb = b.train_step(x, y, 32.0).clone()
        assert torch.equal(pa, pb)
        assert np.array_equal(a.get_arena(), b.get_arena())
        assert not b.get_grads().any() and not a.get_grads().any()


@pytest.mark.parametrize("n", [1, 5, 32, 200, 1024])
def test_reductions_closed_in_kernel_match_the_finalize_launches(n):
    """The BatchNorm reductions closed by the last workgroup of their producer kernels (opt-in) against the same
    step with the separate finalize launches (default): statistics, gradients and updated weights agree to rounding of the
    double-precision sums (1e-6), over three steps; and two runs of the in-kernel form are bit-identical (the
    result does not depend on which workgroup finishes last)."""
    import torch
    from vit_amd import VTCNN, Dims
    rng = np.random.default_rng(11 + n)
    xs = [torch.from_numpy(rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)).cuda() for _ in range(3)]
    ys = [torch.from_numpy(rng.integers(0, 8, size=n).astype(np.int32)).cuda() for _ in range(3)]
    a, b, c = (VTCNN(Dims(num_classes=8), max_batch=n, seed=42) for _ in range(3))
    a.set_fused_reductions(True)
    b.set_fused_reductions(True)
    for x, y in zip(xs, ys):
        outs = []
        for m in (a, b, c):
            p = m.forward(x).cpu().numpy().copy()
            m.backward(y)
            g = m.get_grads().copy()
            st = [m.read(k)[:w] for k, w in (("stat.bn", 16), ("stat.norm1", 64), ("stat.norm2", 64), ("stat.proj.norm", 32))]
            m.update_weights(32.0)
            outs.append((p, g, st, m.get_arena().copy()))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][3], outs[1][3])
        assert rel_err(outs[0][0], outs[2][0]) < 1e-5
        for sa, sc in zip(outs[0][2], outs[2][2]):
            assert rel_err(sa, sc) < 1e-6
        nt = a.layout.n_trainable
        for name, (off, size, shape, kind) in a.layout.tensors.items():
            if kind == 0 and np.abs(outs[2][1][off:off + size]).max() > 0:
                assert rel_err(outs[0][1][off:off + size], outs[2][1][off:off + size]) < 2e-4, name
        assert rel_err(outs[0][3][:nt], outs[2][3][:nt]) < 5e-6  # three steps of last-bit differences in the statistics

