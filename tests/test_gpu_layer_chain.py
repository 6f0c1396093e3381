"""The C++ Layer mirrors (cpp/layers/*.hpp: Conv2DLayer, BatchNorm2DLayer, VisualTransformer with its
FilterTokenizer / TransformerLayer / MultiHeadAttention / ProjectorLayer, MaxPool2DLayer, FlattenLayer,
DenseLayer, SoftmaxLayer) wired into VTCNN the way the reference wires them and driven by
cpp/tests/layer_chain_check.cpp: one training step on the reference fixture's batch.  Weights drawn from
Tensor::global_gen in construction order must be the reference's bit for bit; probabilities, every gradient
tensor and the weights after update_weights(32) within the fp32 tolerance of the reference's own dump."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLD, load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
LOOSE = {"proj.downsample.b": 5e-4, "proj.norm.beta": 5e-4, "conv.b": 5e-4, "bn.beta": 5e-4, "proj.linear3.b": 5e-4}


def _records(path):
    raw = open(path, "rb").read()
    out, pos = {}, 0
    while pos < len(raw):
        (ln,) = struct.unpack_from("<i", raw, pos)
        name = raw[pos + 4:pos + 4 + ln].decode()
        (cnt,) = struct.unpack_from("<i", raw, pos + 4 + ln)
        pos += 8 + ln
        out[name] = np.frombuffer(raw, dtype=np.float32, count=cnt, offset=pos).copy()
        pos += 4 * cnt
    return out


@pytest.mark.parametrize("fixture,n,k", [("step_n2_k8_full.npz", 2, 8), ("step_n5_k8.npz", 5, 8)])
def test_layer_chain_matches_reference(tmp_path, fixture, n, k):
    exe = os.path.join(os.path.dirname(GOLD), "..", "cpp", "build", "layer_chain_check")
    if not os.path.exists(exe):
        pytest.fail("cpp/build/layer_chain_check is not built (run __graft_entry__.build())")
    g = load_golden(fixture)
    if "p0.conv.W" not in g:
        g.update({kk: v for kk, v in load_golden("step_n2_k8_full.npz").items() if kk.startswith("p0.")})
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(np.ascontiguousarray(g["in.X"], dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(g["in.labels"], dtype=np.float32).tobytes())
    subprocess.check_call([exe, str(n), str(k), fin, fout], timeout=300)
    r = _records(fout)
    bad = []
    for name, got in r.items():
        if name.startswith("p0."):
            assert np.array_equal(got, g[name].ravel()), name  # construction / draw order of Tensor::global_gen
    assert rel_err(r["fwd.preds"], g["fwd.preds"]) < TOL
    for name, got in r.items():
        if name.startswith("g."):
            e = rel_err(got, g[name])
            if not e < LOOSE.get(name[2:], TOL):
                bad.append((name, e))
    assert not bad, bad
    assert not r["g.mha.o.W"].any()  # the zero concat
    for name, got in r.items():
        if name.startswith("p1.") and name in g:
            want = g[name].ravel().astype(np.float64)
            gname = "g." + name[3:]
            bound = 1e-6 * np.abs(want).max() + (0.001 / 32.0) * LOOSE.get(name[3:], TOL) * (np.abs(g[gname]).max() if gname in g else 0.0)
            assert np.abs(got - want).max() <= bound, name
    for nm in ("tok.linear1.W", "tok.linear1.b", "tok.linear2.W", "tok.linear2.b"):
        assert np.array_equal(r["p1." + nm], r["p0." + nm]), nm  # the tokenizer's layers have no optimizer


def test_cnn_and_mlp_models_step_matches_reference(tmp_path):
    """The reference's two other models, CNN (model/cnn.hpp:13-119) and MLP with its DropoutLayer (model/mlp.hpp:13-178,
    layers/dropout_layer.hpp:4-52), through the Layer mirrors (cpp/model/cnn.hpp, cpp/model/mlp.hpp): one training step
    each, driven by cpp/tests/models_check.cpp, against the REAL reference's dump of the same step
    (tests/golden/models_step_n8.npz, tools/make_golden_models.py).  Weights before the step bit for bit (construction
    and draw order); probabilities (training mode, hence the MLP's dropout mask drawn from Tensor::global_gen in the
    reference's order; a second training forward with the next mask; eval mode), gradients and updated weights to 1e-4."""
    exe = os.path.join(os.path.dirname(GOLD), "..", "cpp", "build", "models_check")
    if not os.path.exists(exe):
        pytest.fail("cpp/build/models_check is not built (run __graft_entry__.build())")
    g = load_golden("models_step_n8.npz")
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        for pfx in ("cnn", "mlp"):
            f.write(np.ascontiguousarray(g[pfx + ".X"], dtype=np.float32).tobytes())
            f.write(np.ascontiguousarray(g[pfx + ".labels"], dtype=np.float32).tobytes())
    subprocess.check_call([exe, "8", fin, fout], timeout=300)
    r = _records(fout)
    assert set(r) == set(k for k in g if not k.endswith(".X") and not k.endswith(".labels")), set(g) ^ set(r)
    bad = []
    for name, got in r.items():
        want = g[name].ravel()
        parts = name.split(".")
        if parts[1] == "p0":
            assert np.array_equal(got, want), name
        elif parts[1] in ("preds", "preds2", "preds_eval"):
            e = rel_err(got, want)
            if not e < TOL:
                bad.append((name, e))
            assert np.array_equal(got.reshape(8, -1).argmax(1), want.reshape(8, -1).argmax(1)), name
        elif parts[1] == "g":
            e = rel_err(got, want)
            if not e < (5e-4 if parts[-1] in ("b", "beta") else TOL):
                bad.append((name, e))
        elif parts[1] == "p1":
            if parts[-1].startswith("running"):
                e = rel_err(got, want)
                if not e < 1e-5:
                    bad.append((name, e))
                continue
            gname = ".".join([parts[0], "g"] + parts[2:])
            lr = 0.002 if parts[0] == "cnn" else 0.01
            bound = 1e-6 * np.abs(want).max() + (lr / 32.0) * 5e-4 * np.abs(g[gname]).max()
            if not np.abs(got.astype(np.float64) - want).max() <= bound:
                bad.append((name, float(np.abs(got - want).max())))
    assert not bad, bad
    # the dropout mask really was applied in training mode: the eval probabilities differ
    assert rel_err(r["mlp.preds"], r["mlp.preds_eval"]) > 1e-3
