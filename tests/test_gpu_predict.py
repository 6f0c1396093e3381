"""Predict path on the GPU against the REAL reference's outputs (tests/golden/blood_t10k_256.npz):
the shipped 8-class BloodMNIST checkpoint over a 256-image slice of the shipped test split, eval
mode, batches of 32 in the reference DataLoader's order (constructor shuffle + one reset()).

Bar: predicted labels and the loader order bit-exact, probabilities within 1e-4 relative, mean
loss / accuracy as the reference prints them."""
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLD, ROOT, load_golden, rel_err

pytestmark = pytest.mark.gpu


def _paths():
    return (os.path.join(GOLD, "blood_t10k_256-images-idx3-ubyte"), os.path.join(GOLD, "blood_t10k_256-labels-idx1-ubyte"),
            os.path.join(GOLD, "blood_model.bin"))


def test_python_trainer_evaluate_test_matches_reference():
    from vit_amd import VTCNN, Dims
    from vit_amd.engine import DataLoader, Trainer, load_dataset
    g = load_golden("blood_t10k_256.npz")
    img, lbl, ckpt = _paths()
    ds = load_dataset(img, lbl, 10000)
    assert ds.images.shape == (256, 784)
    m = VTCNN(Dims(num_classes=8), max_batch=32, seed=42)
    m.load(ckpt)
    m.set_training(False)
    test_loader = DataLoader(ds.images, ds.labels, 32)
    lines = []
    tr = Trainer(m, test_loader, test_loader, 8, 32, test_loader=test_loader, out=lines.append)
    loss, acc, pred, truth, cm = tr.evaluate_test()
    assert np.array_equal(test_loader.indices, g["order"])
    assert np.array_equal(truth, g["truth"])
    assert np.array_equal(pred, g["pred"])
    assert abs(loss - float(g["loss"][0])) < 1e-4 * max(1.0, abs(float(g["loss"][0])))
    assert abs(acc - float(g["acc"][0])) < 1e-5
    assert int(cm.sum()) == 256 and int(np.trace(cm)) == int((g["pred"] == g["truth"]).sum())
    # probabilities batch by batch (evaluate_test leaves the model in training mode, trainer.hpp:169)
    m.set_training(False)
    order = g["order"]
    x_all = torch.from_numpy(ds.images.reshape(256, 1, 28, 28)).cuda()
    for s in range(0, 256, 32):
        p = m.forward(x_all[torch.from_numpy(order[s:s + 32].astype(np.int64)).cuda()].contiguous()).cpu().numpy()
        assert rel_err(p, g["probs"][s:s + 32]) < 1e-4, s


def test_python_trainer_evaluates_a_general_shape_model_with_20_classes():
    """A model outside the headline shape (general step) with more classes than the fused kernels take and a
    workspace of several loader batches: Trainer.evaluate / evaluate_test must go batch by batch (there is no
    grouped launch set for this shape), keep the bookkeeping on the device for K = 20, and print the reference's
    aggregate and per-class lines (utils/trainer.hpp:128-167)."""
    from vit_amd import VTCNN, Dims
    from vit_amd.engine import DataLoader, Trainer, class_stats
    k, n, bs = 20, 72, 16
    rng = np.random.default_rng(20)
    x = (rng.integers(0, 256, size=(n, 784)).astype(np.float32) / np.float32(255.0))
    y = rng.integers(0, k, size=n).astype(np.int32)
    m = VTCNN(Dims(num_classes=k, tokens=8, token_channels=16, vt_channels=8, heads=2), max_batch=64, seed=42)
    assert not m.fused
    loader = DataLoader(x, y, bs)
    lines = []
    tr = Trainer(m, loader, loader, k, bs, test_loader=loader, out=lines.append)
    vloss, vacc = tr.evaluate()
    loss, acc, pred, truth, cm = tr.evaluate_test()
    assert int(cm.sum()) == n and np.array_equal(truth, y[loader.indices])
    # the same sweep by hand: eval mode, one forward per loader batch
    m.set_training(False)
    xs = torch.from_numpy(x.reshape(n, 1, 28, 28)).cuda()
    losses, accs, preds = [], [], []
    for s0 in range(0, n, bs):
        sel = loader.indices[s0:s0 + bs]
        p = m.forward(xs[torch.from_numpy(sel.astype(np.int64)).cuda()].contiguous()).cpu().numpy()
        losses.append(float(np.mean(-np.log(p[np.arange(len(sel)), y[sel]] + np.float32(1e-8)))))
        preds.append(p.argmax(1))
        accs.append(float(np.mean(preds[-1] == y[sel])))
    m.set_training(True)
    assert np.array_equal(pred, np.concatenate(preds))
    assert abs(loss - np.mean(losses)) < 1e-5 * max(1.0, abs(loss)) and abs(acc - np.mean(accs)) < 1e-6
    want = np.zeros((k, k), dtype=np.int64)
    np.add.at(want, (truth, pred), 1)
    assert np.array_equal(cm, want)
    text = "\n".join(lines)
    st = class_stats(cm)
    assert " - F1 Score (macro):    %.4f" % st["macro_f1"] in text and " - Recall (macro):      " in text
    assert "Per-Class Metrics" in text and text.count("\n") > 2 * k


@pytest.mark.parametrize("n", [1, 7, 256])
def test_eval_forward_other_batch_sizes_vs_oracle(n):
    """Predict at batch sizes the fixture does not hold.  Only the first BatchNorm honours eval mode
    in the reference (the projector's and the transformer's norms always use the statistics of the
    current batch, layers/batch_normalization1d.hpp), so the probabilities depend on the batch and
    the yardstick is the CPU oracle on the same batch: n=1 is Trainer::predict's single image."""
    from oracle.pyoracle import Oracle, make_dims
    from vit_amd import VTCNN, Dims
    from vit_amd.engine import load_dataset
    img, lbl, ckpt = _paths()
    ds = load_dataset(img, lbl, 10000)
    m = VTCNN(Dims(num_classes=8), max_batch=n, seed=42)
    m.load(ckpt)
    m.set_training(False)
    x = np.ascontiguousarray(ds.images[:n].reshape(n, 1, 28, 28))
    p = m.forward(torch.from_numpy(x).cuda()).cpu().numpy()
    o = Oracle(make_dims(num_classes=8), n)
    po = o.forward(m.get_arena(), x, training=False)
    assert rel_err(p, po) < 1e-4
    assert np.array_equal(p.argmax(1), po.argmax(1))


def test_cpp_cli_predict_matches_reference(tmp_path):
    """cpp/build/main bloodmnist predict: the reference's CLI (main.cpp:219-277) on the HIP path."""
    exe = os.path.join(ROOT, "cpp", "build", "main")
    assert os.access(exe, os.X_OK), "cpp/build/main is missing: run `make -C cpp` (hipcc) first"
    g = load_golden("blood_t10k_256.npz")
    img, lbl, ckpt = _paths()
    data = tmp_path / "blood_data"
    data.mkdir()
    for stem in ("train", "t10k"):
        shutil.copy(img, data / ("%s-images-idx3-ubyte" % stem))
        shutil.copy(lbl, data / ("%s-labels-idx1-ubyte" % stem))
    shutil.copy(ckpt, tmp_path / "model_bloodmnist.bin")
    # custom images: the first three test images as PNG files (grey, RGB with equal channels, and a
    # binary PGM, which this build accepts in addition), predicted one at a time by Trainer::predict
    raw = np.fromfile(img, dtype=np.uint8)[16:].reshape(256, 784)
    (tmp_path / "custom_images").mkdir()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_png_golden import encode
    (tmp_path / "custom_images" / "img_0.png").write_bytes(encode(raw[0].reshape(28, 28).astype(np.int64), 0, 8, lambda y: y % 5))
    # (77 v + 150 v + 29 v) >> 8 == v: an RGB file with equal channels decodes to the same grey bytes
    (tmp_path / "custom_images" / "img_1.png").write_bytes(
        encode(np.repeat(raw[1].reshape(28, 28, 1), 3, axis=2).reshape(28, 84).astype(np.int64), 2, 8, lambda y: 4))
    with open(tmp_path / "custom_images" / "img_2.pgm", "wb") as f:
        f.write(b"P5\n28 28\n255\n" + raw[2].tobytes())
    (tmp_path / "custom_images" / "broken.png").write_bytes(b"junk")  # reported on stderr and skipped
    from oracle.pyoracle import Oracle, make_dims
    from tests_util import arena_from_checkpoint
    o = Oracle(make_dims(num_classes=8), 1)
    arena = o.layout.pack(load_golden("step_n2_k8_full.npz"), "p0.")
    arena_from_checkpoint(ckpt, o.layout, arena)
    single = [int(o.forward(arena, (raw[k].astype(np.float32) / np.float32(255.0)).reshape(1, 1, 28, 28),
                            training=False).argmax()) for k in range(3)]
    out_labels = tmp_path / "labels.txt"
    env = dict(os.environ, VTCNN_DATA_ROOT=str(tmp_path), VTCNN_NUM_CLASSES="8", VTCNN_LABELS_OUT=str(out_labels))
    r = subprocess.run([exe, "bloodmnist", "predict"], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Modelo cargado desde" in r.stdout and "[TEST] Final Evaluation" in r.stdout
    pred = np.loadtxt(str(out_labels), dtype=np.int32)
    assert np.array_equal(pred, g["pred"])
    shown = [int(s.split(":")[1]) for s in r.stdout.splitlines() if s.startswith("Imagen predicción:")]
    assert shown == single, (shown, single)
    assert "Error al cargar" in r.stderr and "broken.png" in r.stderr
    line = [s for s in r.stdout.splitlines() if s.startswith("[TEST]")][0]
    # default stream formatting (6 significant digits), as in the reference's predict mode
    mt = re.search(r"Loss: ([0-9.eE+-]+) \| Accuracy: ([0-9.eE+-]+)%", line)
    assert mt, line
    assert abs(float(mt.group(1)) - float(g["loss"][0])) < 2e-4 * max(1.0, float(g["loss"][0])), line
    assert abs(float(mt.group(2)) - 100.0 * float(g["acc"][0])) < 1e-3, line
    # the same sweep dealt over the data-parallel replicas (one replica here): same labels, same printed numbers
    out2 = tmp_path / "labels_dp.txt"
    r2 = subprocess.run([exe, "bloodmnist", "predict"], env=dict(env, VTCNN_LABELS_OUT=str(out2), VTCNN_GPUS="1", VTCNN_DP_FORCE="1"),
                        capture_output=True, text=True, timeout=110)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert np.array_equal(np.loadtxt(str(out2), dtype=np.int32), g["pred"])
    assert [s for s in r2.stdout.splitlines() if s.startswith("[TEST]")][0] == line


def test_cpp_cli_train_one_epoch_runs_and_saves(tmp_path):
    """cpp/build/main bloodmnist train 1 on the 256-image slice: 80/20 split, one epoch, checkpoint
    written in the reference's v1 format and loadable back through the C ABI."""
    from vit_amd import VTCNN, Dims
    exe = os.path.join(ROOT, "cpp", "build", "main")
    assert os.access(exe, os.X_OK)
    img, lbl, _ = _paths()
    data = tmp_path / "blood_data"
    data.mkdir()
    for stem in ("train", "t10k"):
        shutil.copy(img, data / ("%s-images-idx3-ubyte" % stem))
        shutil.copy(lbl, data / ("%s-labels-idx1-ubyte" % stem))
    env = dict(os.environ, VTCNN_DATA_ROOT=str(tmp_path), VTCNN_NUM_CLASSES="8")
    r = subprocess.run([exe, "bloodmnist", "train", "1"], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Epoch 1/1" in r.stdout and "Train Loss:" in r.stdout and "Val Loss:" in r.stdout
    assert "  - Train: 204 muestras" in r.stdout and "  - Validation: 52 muestras" in r.stdout
    ck = tmp_path / "model_bloodmnist.bin"
    assert ck.exists() and ck.stat().st_size == 110838
    m = VTCNN(Dims(num_classes=8), max_batch=32, seed=42)
    a0 = m.get_arena().copy()
    m.load(str(ck))
    a1 = m.get_arena()
    nt = m.layout.n_trainable
    assert np.isfinite(a1).all() and not np.array_equal(a0[:nt], a1[:nt])
    # The run above used the device-resident loop (dataset on the device as bytes, batches gathered there,
    # per-batch loss / accuracy kept on the device).  The reference's own loop shape (every batch through host
    # tensors, VTCNN_HOST_LOOP=1) must train to the same bytes and print the same epoch numbers.
    first = ck.read_bytes()
    r2 = subprocess.run([exe, "bloodmnist", "train", "1"], env=dict(env, VTCNN_HOST_LOOP="1"), capture_output=True, text=True, timeout=110)
    assert r2.returncode == 0, r2.stderr[-2000:]
    assert ck.read_bytes() == first

    def numbers(out, key):
        line = [ln for ln in out.splitlines() if ln.startswith(key)][0]
        return [float(v) for v in re.findall(r"(?:Loss|Accuracy): ([0-9.]+)", line)]
    for key in ("Train Loss:", "Val Loss:", "[TEST] Final Evaluation"):
        a, b = numbers(r.stdout, key), numbers(r2.stdout, key)
        assert len(a) == 2 and abs(a[0] - b[0]) <= 2e-4 * max(1.0, b[0]) and abs(a[1] - b[1]) < 1e-3, (key, a, b)
    cm = lambda out: out[out.index("Matriz de Confusi"):]
    assert cm(r.stdout) == cm(r2.stdout)
    # The data-parallel driver with one rank (replica threads, gradient arena through an RCCL all-reduce, divisor
    # ranks * batch_size) is the single-device run, byte for byte.
    r3 = subprocess.run([exe, "bloodmnist", "train", "1"], env=dict(env, VTCNN_GPUS="1", VTCNN_DP_FORCE="1"), capture_output=True,
                        text=True, timeout=110)
    assert r3.returncode == 0, r3.stderr[-2000:]
    assert ck.read_bytes() == first
    assert numbers(r3.stdout, "Train Loss:") == numbers(r.stdout, "Train Loss:")
    assert cm(r3.stdout) == cm(r.stdout)
    # ... and so is its overlapped form: backward in two calls, the exchange in four buckets on a stream of its own
    # beside the attention backward, evaluation sweeps dealt over the replicas
    r5 = subprocess.run([exe, "bloodmnist", "train", "1"], env=dict(env, VTCNN_GPUS="1", VTCNN_DP_FORCE="1", VTCNN_DP_OVERLAP="1"),
                        capture_output=True, text=True, timeout=110)
    assert r5.returncode == 0, r5.stderr[-2000:]
    assert ck.read_bytes() == first
    assert numbers(r5.stdout, "Train Loss:") == numbers(r.stdout, "Train Loss:")
    assert numbers(r5.stdout, "Val Loss:") == numbers(r.stdout, "Val Loss:")
    assert cm(r5.stdout) == cm(r.stdout)
    # VTCNN_GPUS=1 without the test switch has nothing to exchange and never opens the collective library
    r4 = subprocess.run([exe, "bloodmnist", "train", "1"], env=dict(env, VTCNN_GPUS="1", VT_RCCL_LIBRARY="/nonexistent/librccl.so"),
                        capture_output=True, text=True, timeout=110)
    assert r4.returncode == 0, r4.stderr[-2000:]
    assert ck.read_bytes() == first


def test_cpp_cli_virtual_ranks_match_the_python_grouped_steps(tmp_path):
    """VTCNN_VIRTUAL_RANKS=3: one epoch over 204 images = two grouped steps of three loader batches (one summed update each,
    divisor 3 * 32) and the short last batch alone; the checkpoint equals, byte for byte, the arena of the same steps
    driven through vt_model_train_step_grouped from Python on the same loader order."""
    import torch
    from vit_amd import VTCNN, Dims
    from vit_amd.engine import load_dataset, loader_indices
    exe = os.path.join(ROOT, "cpp", "build", "main")
    assert os.access(exe, os.X_OK)
    img, lbl, _ = _paths()
    data = tmp_path / "blood_data"
    data.mkdir()
    for stem in ("train", "t10k"):
        shutil.copy(img, data / ("%s-images-idx3-ubyte" % stem))
        shutil.copy(lbl, data / ("%s-labels-idx1-ubyte" % stem))
    env = dict(os.environ, VTCNN_DATA_ROOT=str(tmp_path), VTCNN_NUM_CLASSES="8", VTCNN_VIRTUAL_RANKS="3")
    r = subprocess.run([exe, "bloodmnist", "train", "1"], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Epoch 1/1" in r.stdout and "Train Loss:" in r.stdout and "Val Loss:" in r.stdout
    ck = tmp_path / "model_bloodmnist.bin"
    assert ck.exists() and ck.stat().st_size == 110838
    ds = load_dataset(img, lbl, 60000)
    ntrain = int(256 * 0.8)
    order = loader_indices(ntrain, 42, 1)  # the constructor's shuffle, then the epoch's reset()
    x = torch.from_numpy(np.ascontiguousarray(ds.images[:ntrain][order].reshape(ntrain, 1, 28, 28))).cuda()
    y = torch.from_numpy(np.ascontiguousarray(ds.labels[:ntrain][order].astype(np.int32))).cuda()
    m = VTCNN(Dims(num_classes=8), max_batch=96, seed=42)
    for first in (0, 96):
        m.train_step_grouped(x[first:first + 96].contiguous(), y[first:first + 96].contiguous(), 32, 32.0)
    m.train_step(x[192:].contiguous(), y[192:].contiguous(), 32.0)
    m.save(str(tmp_path / "python.bin"))
    assert ck.read_bytes() == (tmp_path / "python.bin").read_bytes()
    # and it is not the one-batch-at-a-time run
    r2 = subprocess.run([exe, "bloodmnist", "train", "1"], env=dict(env, VTCNN_VIRTUAL_RANKS="1"), capture_output=True, text=True,
                        timeout=110)
    assert r2.returncode == 0 and ck.read_bytes() != (tmp_path / "python.bin").read_bytes()


def test_dp_entry_points_sum_over_visible_gpus():
    """cpp/build/dp_check: argument errors of vt_dp_init / vt_dp_allreduce_*, then an exact in-place sum over
    every visible GPU through both the grouped call and one call per rank from separate host threads."""
    exe = os.path.join(ROOT, "cpp", "build", "dp_check")
    assert os.access(exe, os.X_OK)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "GPU(s) OK" in r.stdout


def test_predict_batch_8192_labels_vs_oracle():
    """The predict configuration "one forward of 8192 Fashion-shaped images, 10 classes, argmax
    bit-exact vs CPU": a single semantic batch of 8192 (BatchNorm1D statistics over all 8192*L and
    8192*P rows, as the reference computes them when handed such a batch).  The reference itself
    would materialise its dead [16,8192,8192] attention scores (4.3 GB per temporary) for this
    batch, so the checker is the CPU oracle in double precision (pinned bit-exactly to the
    reference in single precision at N=2/5/32).  Labels must be equal wherever the oracle's top-2
    probabilities are further apart than 1e-5 (none closer is expected)."""
    from oracle.pyoracle import Oracle, make_dims
    from vit_amd import VTCNN, Dims
    n, k = 8192, 10
    rng = np.random.default_rng(8192)
    x = rng.integers(0, 256, size=(n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)
    m = VTCNN(Dims(num_classes=k), max_batch=n, seed=42)
    m.set_training(False)
    p = m.forward(torch.from_numpy(x).cuda())
    _, am = m.metrics()
    labels = am.cpu().numpy()
    p = p.cpu().numpy()
    o = Oracle(make_dims(num_classes=k), n, f64=True)
    po = o.forward(m.get_arena().astype(np.float64), x, training=False)
    assert rel_err(p, po) < 1e-4
    top2 = np.sort(po, axis=1)[:, -2:]
    decided = (top2[:, 1] - top2[:, 0]) > 1e-5
    assert decided.mean() > 0.999
    assert np.array_equal(labels[decided], po.argmax(1)[decided])
    assert np.array_equal(labels, p.argmax(1))  # first maximum wins, as Tensor::argmax_batch


def test_device_loader_gather_bit_exact_and_eval_bookkeeping():
    """Device-resident input pipeline: the dataset stays on the GPU as the IDX bytes, a batch is one
    gather launch with /255.0f fused in.  Bit-exact against the host decode (vt_idx_read_images) in
    the reference loader's order, ragged last batch included; then the same sweep through
    Trainer.evaluate_test (device-side loss / hit / confusion bookkeeping) must reproduce the real
    reference's predictions, loss and accuracy."""
    from vit_amd import VTCNN, Dims
    from vit_amd.engine import DataLoader, Trainer, load_dataset
    g = load_golden("blood_t10k_256.npz")
    img, lbl, ckpt = _paths()
    ds = load_dataset(img, lbl, 10000)
    assert ds.images_u8.dtype == np.uint8 and ds.images_u8.shape == (256, 784)
    dl = DataLoader(ds.images_u8, ds.labels, 48)  # 256 = 5 * 48 + 16: ragged tail
    seen = 0
    while dl.has_next():
        x, y = dl.next_batch()
        idx = dl.indices[seen:seen + x.shape[0]]
        assert np.array_equal(x.cpu().numpy().reshape(x.shape[0], -1), ds.images[idx])
        assert np.array_equal(y.cpu().numpy(), ds.labels[idx])
        seen += x.shape[0]
    assert seen == 256
    m = VTCNN(Dims(num_classes=8), max_batch=32, seed=42)
    m.load(ckpt)
    test_loader = DataLoader(ds.images_u8, ds.labels, 32)
    tr = Trainer(m, test_loader, test_loader, 8, 32, test_loader=test_loader, out=lambda s: None)
    loss, acc, pred, truth, cm = tr.evaluate_test()
    assert np.array_equal(pred, g["pred"]) and np.array_equal(truth, g["truth"])
    assert abs(loss - float(g["loss"][0])) < 1e-4 * max(1.0, abs(float(g["loss"][0])))
    assert abs(acc - float(g["acc"][0])) < 1e-5
    want = np.zeros((8, 8), dtype=np.int64)
    np.add.at(want, (g["truth"], g["pred"]), 1)
    assert np.array_equal(cm, want)
    # the same sweep with room for several loader batches per launch set (grouped forward: every
    # batch of 32 keeps its own BatchNorm1D statistics): same predictions, loss, accuracy
    m4 = VTCNN(Dims(num_classes=8), max_batch=96, seed=42)  # 3 batches per call: 256 = 2*96 + 64
    m4.load(ckpt)
    loader4 = DataLoader(ds.images_u8, ds.labels, 32)
    tr4 = Trainer(m4, loader4, loader4, 8, 32, test_loader=loader4, out=lambda s: None)
    loss4, acc4, pred4, truth4, cm4 = tr4.evaluate_test()
    assert np.array_equal(pred4, g["pred"]) and np.array_equal(truth4, g["truth"]) and np.array_equal(cm4, want)
    assert abs(loss4 - float(g["loss"][0])) < 1e-4 * max(1.0, abs(float(g["loss"][0])))
    assert abs(acc4 - float(g["acc"][0])) < 1e-5


@pytest.mark.parametrize("n", [256, 250])
def test_grouped_forward_matches_batch_by_batch(n):
    """vt_model_forward_grouped: an evaluation sweep (consecutive loader batches of 32, last one
    short when n = 250) in one launch set.  Probabilities and the final running statistics must
    equal those of forwarding the batches one after another up to rounding of the statistics (the
    projector's variance is accumulated around its running mean, which moves between batches in the
    sequential run: 1e-6 relative), labels exactly; for the full 256-image slice they must match
    the real reference's probabilities and labels."""
    from vit_amd import VTCNN, Dims
    from vit_amd.engine import load_dataset
    g = load_golden("blood_t10k_256.npz")
    img, lbl, ckpt = _paths()
    ds = load_dataset(img, lbl, 10000)
    order = torch.from_numpy(g["order"][:n].astype(np.int64))
    x = torch.from_numpy(ds.images.reshape(256, 1, 28, 28))[order].contiguous().cuda()
    a = VTCNN(Dims(num_classes=8), max_batch=256, seed=42)
    a.load(ckpt)
    a.set_training(False)
    seq = torch.cat([a.forward(x[s:s + 32]).clone() for s in range(0, n, 32)]).cpu().numpy()
    b = VTCNN(Dims(num_classes=8), max_batch=256, seed=42)
    b.load(ckpt)
    b.set_training(False)
    grp = b.forward_grouped(x, 32).cpu().numpy()
    assert rel_err(grp, seq) < 1e-6
    assert np.array_equal(grp.argmax(1), seq.argmax(1))
    nt = a.layout.n_trainable
    assert np.array_equal(a.get_arena()[:nt], b.get_arena()[:nt])
    assert np.allclose(a.get_arena()[nt:], b.get_arena()[nt:], rtol=1e-5, atol=1e-7)  # running statistics, batch order
    if n == 256:
        assert rel_err(grp, g["probs"]) < 1e-4
        assert np.array_equal(grp.argmax(1), g["pred"])
    with pytest.raises(Exception):
        b.backward(torch.zeros(n, dtype=torch.int32, device="cuda"))  # no backward after a grouped forward
    b.set_training(True)
    with pytest.raises(Exception):
        b.forward_grouped(x[:30].contiguous(), 6)  # training mode takes groups that are a multiple of 4 (tests/test_gpu_grouped_train.py)


def test_single_image_graph_replay_matches_eager():
    """Trainer::predict's single image (utils/trainer.hpp:199-243) as a captured forward graph:
    replaying it on refilled input gives exactly the eager forward's probabilities."""
    from vit_amd import VTCNN, Dims
    from vit_amd.engine import load_dataset
    img, lbl, ckpt = _paths()
    ds = load_dataset(img, lbl, 16)
    a, b = (VTCNN(Dims(num_classes=8), max_batch=1, seed=42) for _ in range(2))
    for m in (a, b):
        m.load(ckpt)
        m.set_training(False)
    xbuf = torch.zeros(1, 1, 28, 28, device="cuda")
    out = b.capture_forward(xbuf)
    for k in range(6):
        x = torch.from_numpy(np.ascontiguousarray(ds.images[k].reshape(1, 1, 28, 28))).cuda()
        pa = a.forward(x).cpu().numpy().copy()
        xbuf.copy_(x)
        b.step_graph()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), pa), k
    assert np.array_equal(a.get_arena(), b.get_arena())  # running statistics moved the same way


def _sharded_eval_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # two ranks share the one GPU here, which RCCL refuses
    from vit_amd import VTCNN, Dims
    from vit_amd.engine import DataLoader, Trainer, load_dataset
    img, lbl, ckpt = _paths()
    ds = load_dataset(img, lbl, 10000)
    m = VTCNN(Dims(num_classes=8), max_batch=96, seed=42)
    m.load(ckpt)
    if rank == 1:  # a rank whose BatchNorm2D running statistics drifted: the sweep must evaluate rank 0's model
        a = m.get_arena()
        lo = m.layout.tensors["bn.running_mean"]
        a[lo[0]:lo[0] + lo[1]] += 0.5
        m.set_arena(a)
    lines = []
    t = Trainer(m, None, None, 8, 32, test_loader=DataLoader(ds.images_u8, ds.labels, 32), out=lines.append, world=world, rank=rank)
    loss, acc, pred, truth, cm = t.evaluate_test()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), loss=loss, acc=acc, pred=pred, truth=truth, cm=cm, nlines=len(lines))
    dist.barrier()
    dist.destroy_process_group()


def test_python_trainer_sharded_evaluation(tmp_path):
    """Trainer.evaluate_test with world 2 ("replicas only": chunks of loader batches dealt to the ranks, per-batch
    numbers, labels and confusion matrix merged by one sum): every rank ends with the single-process sweep's loss,
    accuracy, labels and confusion matrix, i.e. the reference's own numbers for the shipped checkpoint; only rank 0
    prints.  Two processes on the one GPU, merged over gloo."""
    import socket
    import torch.multiprocessing as mp
    g = load_golden("blood_t10k_256.npz")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.spawn(_sharded_eval_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [dict(np.load(os.path.join(str(tmp_path), "rank%d.npz" % k))) for k in range(2)]
    for k in range(2):
        assert np.array_equal(r[k]["pred"], g["pred"]), k
        assert abs(float(r[k]["loss"]) - float(g["loss"][0])) < 2e-4 * max(1.0, float(g["loss"][0]))
        assert abs(float(r[k]["acc"]) - float(g["acc"][0])) < 1e-5
        assert int(r[k]["cm"].sum()) == 256 and np.array_equal(r[k]["cm"], r[0]["cm"])
    assert int(r[0]["nlines"]) > 10 and int(r[1]["nlines"]) == 0
