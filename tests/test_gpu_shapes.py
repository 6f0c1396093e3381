"""The scaled (224x224x3, 64 conv channels, 16 tokens, 8 heads) and wide (196 tokens of 512 channels)
VTCNN shapes through the C ABI: one training step against the REAL reference's outputs
(tests/golden/shape_*_n2.npz from tools/make_golden_shapes.py; big tensors stored as every 7th
element plus their sum), and an irregular small shape against the CPU oracle.  These shapes run
the general step (one matrix-core GEMM per DenseLayer / batched matmul), not the headline kernels."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from tests_util import STRIDE, compare_compact, shape_case

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("name", ["scaled", "wide"])
def test_step_matches_reference(name):
    import torch
    from vit_amd import VTCNN, Dims
    g = load_golden("shape_%s_n2.npz" % name)
    dims, n, k, seed, x, y = shape_case(g)
    m = VTCNN(Dims(**dims), max_batch=n, seed=42)
    lo = m.layout
    arena0 = m.get_arena().copy()
    for nm, (off, size, shape, kind) in lo.tensors.items():
        if "p0." + nm in g:
            compare_compact(g, "p0." + nm, arena0[off:off + size], 0, exact=True)  # seed-42 draw order
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    p = m.forward(xd).cpu().numpy()
    compare_compact(g, "fwd.preds", p, TOL)
    assert np.array_equal(p.argmax(1), g["fwd.argmax"].ravel())
    m.backward(yd)
    gr = m.get_grads()
    assert not gr[lo.tensors["mha.o.W"][0]:][:lo.tensors["mha.o.W"][1]].any()
    direct = True
    for nm, (off, size, shape, kind) in lo.tensors.items():
        if kind == 0:
            try:
                compare_compact(g, "g." + nm, gr[off:off + size], TOL)
            except AssertionError:
                direct = False
    if not direct:
        # The deltas that reach the projector are sparse (one per pooling window), so a single window
        # whose two largest values tie to fp32 rounding moves a gradient by percents when the
        # reference's sequential sums and this path's tiled sums pick different winners.  Then the
        # check is the one used for large batches: every disagreement with exact arithmetic (the
        # oracle's fp64 build, itself pinned to this fixture by tests/test_oracle_shapes.py) must be a
        # near-tie, and with this path's decisions imposed every gradient must agree to 1e-4.
        _check_against_exact_math(m, dims, n, x, y, arena0, p, gr)
    # update, second training forward, eval forward: the reference's numbers for the same sequence
    m.update_weights(32.0)
    p2 = m.forward(xd).cpu().numpy().copy()
    m.set_training(False)
    pe = m.forward(xd).cpu().numpy().copy()
    assert np.array_equal(pe.argmax(1), g["eval.argmax"].ravel())
    try:
        compare_compact(g, "fwd2.preds", p2, TOL)
        compare_compact(g, "eval.preds", pe, TOL)
    except AssertionError:
        # At 100 352 pixel rows the reference's own sequential fp32 sums are the less exact side (wide
        # shape, eval mode: the fixture is 3e-4 from fp64 arithmetic, this path 5e-5;
        # tools/diag_generic_eval.py).  Then the yardstick is the same sequence in the fp64 oracle.
        from oracle.pyoracle import Oracle, make_dims
        o = Oracle(make_dims(**dims), n, f64=True)
        a64 = arena0.astype(np.float64)
        o.forward(a64, x, training=True)
        o.sgd(a64, o.backward(a64, y), 0.001, 32.0)
        q2 = o.forward(a64, x, training=True).copy()
        qe = o.forward(a64, x, training=False).copy()
        assert rel_err(p2, q2) < TOL and rel_err(pe, qe) < TOL
        assert rel_err(g["fwd2.preds"], q2) < 1e-3 and rel_err(g["eval.preds"], qe) < 1e-3


def _check_against_exact_math(m, dims, n, x, y, arena0, p, gr, tol=TOL, gap_tol=1e-5, frac_tol=1e-3, loose=None):
    from oracle.pyoracle import Oracle, make_dims
    H, C2 = dims["input_dim"], dims["vt_channels"]
    pk = H // 14
    idx = m.read("fwd.pool_idx").view(np.int32)[:n * C2 * 196].reshape(n, C2, 14, 14)
    ii, jj = np.arange(14).reshape(1, 1, 14, 1), np.arange(14).reshape(1, 1, 1, 14)
    slots = ((idx // H - ii * pk) * pk + (idx % H - jj * pk)).astype(np.uint8).ravel()
    mask = (m.read("bwd.du") != 0).astype(np.uint8)
    o = Oracle(make_dims(**dims), n, f64=True)
    a64 = arena0.astype(np.float64)
    p64 = o.forward(a64.copy(), x, training=True)
    assert rel_err(p, p64) < tol
    ref_slots, gap = o.pool_decisions()
    diff = np.nonzero(ref_slots != slots)[0]
    assert diff.size <= frac_tol * slots.size, diff.size
    assert np.all(gap[diff] < gap_tol * np.abs(o.get("fwd.pooled")).max()), gap[diff].max() / np.abs(o.get("fwd.pooled")).max()
    o.set_routing(slots, mask)
    o.forward(a64.copy(), x, training=True)
    go = o.backward(a64, y)
    bad = []
    for nm, (off, size, shape, kind) in m.layout.tensors.items():
        if kind == 0 and nm != "mha.o.W":
            e = rel_err(gr[off:off + size], go[off:off + size])
            if not e < (loose or {}).get(nm, tol):
                bad.append((nm, round(e, 4)))
    assert not bad, "; ".join("%s %.4f" % b for b in bad)


@pytest.mark.parametrize("dims,n", [
    (dict(num_classes=5, image_channels=2, input_dim=42, tokens=10, token_channels=24, vt_channels=48, heads=3), 9),
    (dict(num_classes=3, image_channels=1, input_dim=14, tokens=7, token_channels=20, vt_channels=10, heads=5), 3),
    (dict(num_classes=11, image_channels=4, input_dim=56, tokens=33, token_channels=36, vt_channels=6, heads=1), 70),
    # more than 256 projector channels: a thread of the pooling-delta sums owns several channels
    (dict(num_classes=4, image_channels=1, input_dim=28, tokens=6, token_channels=16, vt_channels=264, heads=2), 3),
])
def test_irregular_shape_against_oracle(dims, n):
    """Dimensions that are multiples of nothing convenient (odd channel counts, token counts, head
    widths of 4 to 36, 1x1 to 4x4 pooling windows, a batch above one 64-row tile): forward, every
    gradient, the two-call backward."""
    import torch
    from vit_amd import VTCNN, Dims
    from oracle.pyoracle import Oracle, make_dims
    rng = np.random.default_rng(5)
    ci, hw, k = dims["image_channels"], dims["input_dim"], dims["num_classes"]
    x = (rng.integers(0, 256, size=(n, ci, hw, hw)).astype(np.float32) / np.float32(255.0))
    y = rng.integers(0, k, size=n).astype(np.int32)
    m = VTCNN(Dims(**dims), max_batch=n, seed=42)
    arena = m.get_arena().copy()
    o = Oracle(make_dims(**dims), n)
    po = o.forward(arena.copy(), x, training=True)
    p = m.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    assert rel_err(p, po) < TOL
    for nm in ("fwd.tok_out", "fwd.tr_out", "fwd.proj_z", "fwd.pooled"):
        assert rel_err(m.read(nm)[:o.get(nm).size], o.get(nm)) < TOL, nm
    m.backward(torch.from_numpy(y).cuda())
    gg = m.get_grads()
    # gradients against exact arithmetic (bias gradients are cancelling sums: the fp32 oracle's own
    # rounding is 2e-4 there), with this path's pooling / ReLU decisions imposed
    _check_against_exact_math(m, dims, n, x, y, arena, p, gg)
    # the two-call backward writes the same bytes
    m.forward(torch.from_numpy(x).cuda())
    m.zero_grad()
    m.backward_main(torch.from_numpy(y).cuda())
    m.backward_attention()
    # (running statistics moved between the two forwards; the weights did not, so gradients agree)
    assert np.array_equal(m.get_grads(), gg)
    # ... and so does the one-stream order (by default the transformer / attention half runs on a side stream)
    m.set_concurrency(False)
    m.forward(torch.from_numpy(x).cuda())
    m.zero_grad()
    m.backward(torch.from_numpy(y).cuda())
    assert np.array_equal(m.get_grads(), gg)


def test_wide_shape_bf16_mode_within_1e_2():
    """The optional bf16-MFMA mode at the wide shape against the reference's fp32 step: predictions, loss and the
    weights after one update within 1e-2 (the bar stated for that mode), and EVERY gradient tensor within 1e-2 of exact
    arithmetic under the run's own pooling / ReLU decisions, at the fixture's two images and at sixteen synthetic ones."""
    import torch
    from vit_amd import VTCNN, Dims
    g = load_golden("shape_wide_n2.npz")
    dims, n, k, seed, x, y = shape_case(g)
    m = VTCNN(Dims(**dims), max_batch=16, seed=42)
    m.set_matmul_precision(True)
    lo = m.layout
    arena0 = m.get_arena().copy()
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    p = m.forward(xd).cpu().numpy()
    compare_compact(g, "fwd.preds", p, 1e-2)
    loss = float(-np.log(p[np.arange(n), y]).mean())
    assert abs(loss - float(g["fwd.loss"][0])) < 1e-2 * float(g["fwd.loss"][0])
    m.backward(yd)
    gr = m.get_grads()
    m.update_weights(32.0)
    arena1 = m.get_arena()
    # bf16 noise in the projector output re-decides near-tied pooling windows, and the deltas behind the pool
    # are sparse, so against the reference's own winners the upstream gradients can differ by percents
    # without any arithmetic being wrong; the weights after the update are compared with the fixture, the
    # gradients with exact arithmetic under this run's own decisions (below).
    for nm, (off, size, shape, kind) in lo.tensors.items():
        if kind != 0 or "g." + nm not in g or not np.abs(g["g." + nm]).max() > 0:
            continue
        b = g["g." + nm].ravel()
        w0 = arena0[off:off + size]
        if np.abs(w0).max() > 0 and b.size == size:  # weights after the update, where there were weights before
            want = w0 - (b * np.float32(1 / 32.0)) * np.float32(0.001)
            assert rel_err(arena1[off:off + size], want) < 1e-2, nm
    # The bar proper: with this run's own pooling winners and ReLU mask imposed on exact arithmetic (fp64 oracle), every
    # gradient tensor within 1e-2 (measured: worst 4e-3).  Windows where the bf16 run picked another winner than exact
    # arithmetic must be near-ties at the bf16 noise level (gap below 5e-2 of the largest pooled value).
    _check_against_exact_math(m, dims, n, x, y, arena0, p, gr, tol=1e-2, gap_tol=5e-2, frac_tol=5e-2)
    # the same at sixteen images (synthetic), from the initial weights: worst tensor 5e-3
    n16 = 16
    rng = np.random.default_rng(11)
    x16 = (rng.integers(0, 256, size=(n16, dims["image_channels"], dims["input_dim"], dims["input_dim"])).astype(np.float32) / np.float32(255.0))
    y16 = rng.integers(0, dims["num_classes"], size=n16).astype(np.int32)
    m.set_arena(arena0)
    p16 = m.forward(torch.from_numpy(x16).cuda()).cpu().numpy()
    m.backward(torch.from_numpy(y16).cuda())
    _check_against_exact_math(m, dims, n16, x16, y16, arena0, p16, m.get_grads(), tol=1e-2, gap_tol=5e-2, frac_tol=5e-2)
    with pytest.raises(RuntimeError):
        VTCNN(Dims(num_classes=8), max_batch=2, seed=42).set_matmul_precision(True)  # headline kernels: fp32 only


def test_scaled_shape_pooling_without_the_stored_matrix_is_bit_identical():
    """VT_GEN_STORE_Z=0: the scaled shape forms the projector's pre-norm output twice (statistics, then normalise + ReLU +
    pool) instead of storing it (the default).  Same probabilities, pooled values, arg-max positions,
    gradients and, read on request, the same matrix, bit for bit."""
    import os
    import torch
    from vit_amd import VTCNN, Dims
    g = load_golden("shape_scaled_n2.npz")
    dims, n, k, seed, x, y = shape_case(g)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    out = {}
    for mode in ("0", "1"):
        os.environ["VT_GEN_STORE_Z"] = mode
        try:
            m = VTCNN(Dims(**dims), max_batch=n, seed=42)
            p = m.forward(xd).cpu().numpy().copy()
            pooled = m.read("fwd.pooled").copy()
            m.backward(yd)
            out[mode] = (p, pooled, m.get_grads().copy(), m.read("fwd.proj_z").copy(), m.get_arena().copy())
        finally:
            os.environ.pop("VT_GEN_STORE_Z", None)
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("exact", [False, True])
def test_headline_shape_on_the_general_launchers_matches_the_tuned_step(exact, monkeypatch):
    """VT_GENERAL_PATH=1 at create runs the headline shape on the general-shape launchers; with vt_set_exact_products
    every product of that step is an exact fp32 matrix instruction.  Same weights, same batch: predictions and every
    gradient against the tuned step's (whose deep contractions are hi/lo split products), pooling decisions checked
    through exact arithmetic as for the other shapes when a near-tie flips one."""
    import torch
    from vit_amd import VTCNN, Dims
    from vit_amd._lib import lib
    dims = dict(num_classes=8, image_channels=1, input_dim=28, tokens=16, token_channels=32, vt_channels=16, heads=2)
    n = 64
    rng = np.random.default_rng(11)
    x = (rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0))
    y = rng.integers(0, 8, size=n).astype(np.int32)
    tuned = VTCNN(Dims(**dims), max_batch=n, seed=42)
    monkeypatch.setenv("VT_GENERAL_PATH", "1")
    before = lib.vt_set_exact_products(1 if exact else 0)
    try:
        gen = VTCNN(Dims(**dims), max_batch=n, seed=42)
        monkeypatch.delenv("VT_GENERAL_PATH")
        arena = gen.get_arena().copy()
        assert np.array_equal(arena, tuned.get_arena())
        xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        pt = tuned.forward(xd).cpu().numpy().copy()
        pg = gen.forward(xd).cpu().numpy().copy()
        assert rel_err(pg, pt) < TOL
        tuned.backward(yd)
        gen.backward(yd)
        gt, gg = tuned.get_grads(), gen.get_grads()
        lo = gen.layout
        bad = [nm for nm, (off, size, shape, kind) in lo.tensors.items()
               if kind == 0 and rel_err(gg[off:off + size], gt[off:off + size]) > 5e-4]
        if bad:  # a pooling window decided the other way: both against exact arithmetic under their own decisions
            _check_against_exact_math(gen, dims, n, x, y, arena, pg, gg)
    finally:
        lib.vt_set_exact_products(before)
