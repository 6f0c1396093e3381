"""Pins the CPU oracle (oracle/vtcnn_oracle.c) to the real reference.

The fixtures under tests/golden/ were produced by oracle/ref_harness.cpp, i.e. by the reference's
own headers compiled in the build container (tools/make_golden.py).  Tolerances: 1e-4 relative
(max|a-b|/max|b|) per tensor, the bar the fp32 GPU path is held to as well; argmax bit-exact.
Long sequential fp32 sums in the reference (e.g. the downsample bias gradient over N*784 rows)
carry ~2e-4 of their own rounding noise at N=32, so those few tensors get 5e-4.
"""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle.pyoracle import Oracle, make_dims

TOL = 1e-4
LOOSE = {"proj.downsample.b": 5e-4, "proj.norm.beta": 5e-4, "conv.b": 5e-4, "bn.beta": 5e-4,
         "proj.linear3.b": 5e-4}

FWD_MAP = {
    "fwd.conv_out": "fwd.conv_out", "fwd.x": "fwd.x", "fwd.tok_A": "fwd.tok_A", "fwd.tok_out": "fwd.tok_out",
    "fwd.tr_a": "fwd.tr_a", "fwd.tr_hdn": "fwd.tr_hdn", "fwd.tr_out": "fwd.tr_out", "fwd.proj_a": "fwd.proj_a",
    "fwd.proj_z": "fwd.proj_z", "fwd.proj_out": "fwd.proj_out", "fwd.pooled": "fwd.pooled",
    "fwd.logits": "fwd.logits", "fwd.mha_out": "fwd.mha_out",
}
BWD = ["bwd.d_logits", "bwd.d_pooled", "bwd.d_projout", "bwd.proj_dz", "bwd.proj_dx", "bwd.proj_dt", "bwd.tr_ds2",
       "bwd.tr_dr", "bwd.tr_dx", "bwd.tok_dx", "bwd.bn_dx"]


def run_step(g, n, k):
    dims = make_dims(num_classes=k)
    o = Oracle(dims, n)
    arena = o.layout.pack(g, "p0.")
    preds = o.forward(arena, g["in.X"], training=True)
    loss = o.loss(g["in.labels"])
    grads = o.backward(arena, g["in.labels"])
    return o, arena, preds, loss, grads


@pytest.mark.parametrize("fx,n,k", [("step_n2_k8_full.npz", 2, 8), ("step_n32_k10.npz", 32, 10)])
def test_step_matches_reference(fx, n, k):
    g = load_golden(fx)
    o, arena, preds, loss, grads = run_step(g, n, k)
    assert rel_err(preds, g["fwd.preds"]) < TOL
    assert abs(loss - float(g["fwd.loss"][0])) < TOL * abs(float(g["fwd.loss"][0]))
    assert np.array_equal(o.argmax(), g["fwd.argmax"])
    for name, (off, size, shape, kind) in o.layout.tensors.items():
        if kind != 0:
            continue
        e = rel_err(grads[off:off + size], g["g." + name])
        assert e < LOOSE.get(name, TOL), (name, e)
    assert np.all(o.layout.view(grads, "mha.o.W") == 0.0)
    for nm in ("tok.linear1.W", "tok.linear1.b", "tok.linear2.W", "tok.linear2.b"):
        assert rel_err(o.get("g." + nm), g["g." + nm]) < TOL, nm
    if "fwd.conv_out" in g:
        for mine, ref in FWD_MAP.items():
            assert rel_err(o.get(mine), g[ref]) <= TOL, mine
        assert np.all(o.get("fwd.mha_out") == 0.0)
        for nm in BWD:
            assert rel_err(o.get(nm), g[nm]) < TOL, nm
    # SGD + running statistics
    o.sgd(arena, grads, 0.001, 32.0)
    for name, (off, size, shape, kind) in o.layout.tensors.items():
        e = rel_err(arena[off:off + size], g["p1." + name])
        assert e < (1e-6 if kind != 1 else 1e-12), (name, e)
    # frozen tokenizer: bit-identical to its initial value
    for nm in ("tok.linear1.W", "tok.linear2.W"):
        assert np.array_equal(o.layout.view(arena, nm), g["p0." + nm])
    # second training forward, then eval-mode forward (BN2D on running stats only)
    p2 = o.forward(arena, g["in.X"], training=True)
    assert rel_err(p2, g["fwd2.preds"]) < TOL
    pe = o.forward(arena, g["in.X"], training=False)
    assert rel_err(pe, g["eval.preds"]) < TOL
    assert np.array_equal(o.argmax(), g["eval.argmax"])
    for nm in ("bn.running_mean", "bn.running_var", "norm2.running_var", "proj.norm.running_mean"):
        assert rel_err(o.layout.view(arena, nm), g["p2." + nm]) < 1e-5, nm


def test_odd_batch():
    g = load_golden("step_n5_k8.npz")
    g0 = load_golden("step_n2_k8_full.npz")
    g.update({k: v for k, v in g0.items() if k.startswith("p0.")})
    o, arena, preds, loss, grads = run_step(g, 5, 8)
    assert rel_err(preds, g["fwd.preds"]) < TOL
    for name, (off, size, shape, kind) in o.layout.tensors.items():
        if kind == 0:
            assert rel_err(grads[off:off + size], g["g." + name]) < LOOSE.get(name, TOL), name


def test_blood_checkpoint_labels():
    """Shipped 8-class checkpoint over a 256-image slice of the shipped BloodMNIST test split,
    eval mode, batches of 32 in the reference loader's order: predicted labels bit-exact."""
    import os
    from conftest import GOLD
    g = load_golden("blood_t10k_256.npz")
    raw = np.fromfile(os.path.join(GOLD, "blood_t10k_256-images-idx3-ubyte"), dtype=np.uint8)[16:]
    imgs = (raw.astype(np.float32) / np.float32(255.0)).reshape(256, 1, 28, 28)
    from tests_util import arena_from_checkpoint
    dims = make_dims(num_classes=8)
    o = Oracle(dims, 32)
    init = load_golden("step_n2_k8_full.npz")
    arena = o.layout.pack(init, "p0.")
    arena_from_checkpoint(os.path.join(GOLD, "blood_model.bin"), o.layout, arena)
    order = g["order"]
    pred = []
    for s in range(0, 256, 32):
        o.forward(arena, imgs[order[s:s + 32]], training=False)
        pred.extend(o.argmax().tolist())
    assert np.array_equal(np.array(pred), g["pred"])


def test_twelve_step_trajectory_matches_reference():
    """Twelve consecutive training steps from the seed-42 model on the reference's own batches
    (tools/make_golden.py, `ref_harness traj`): the restatement follows the real reference step after
    step -- per-step loss, every predicted label, and the weights it ends on."""
    g = load_golden("traj_n32_k8_t12.npz")
    dims = make_dims(num_classes=8)
    o = Oracle(dims, 32)
    arena = o.layout.pack(load_golden("step_n2_k8_full.npz"), "p0.")
    for t in range(12):
        x = g["traj.x_u8"][t].astype(np.float32) / np.float32(255.0)
        o.forward(arena, x, training=True)
        assert abs(o.loss(g["traj.y"][t]) - float(g["traj.loss"][t])) < 1e-5 * abs(float(g["traj.loss"][t])), t
        assert np.array_equal(o.argmax(), g["traj.argmax"][t]), t
        grads = o.backward(arena, g["traj.y"][t])
        o.sgd(arena, grads, 0.001, 32.0)
    for name, (off, size, shape, kind) in o.layout.tensors.items():
        e = rel_err(arena[off:off + size], g["pT." + name])
        assert e < 2e-6, (name, e)
