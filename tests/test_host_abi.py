"""CPU-side checks of the C ABI library: it loads without a GPU, exports every symbol that
include/vtcnn_hip.h declares, and its host-only (integer / byte exact) entry points reproduce the
reference's known answers: weight-initialisation draw order, .bin checkpoint bytes, loader
shuffling, IDX decoding.  No compute entry point is called here."""
import ctypes as C
import hashlib
import json
import os
import re
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT, load_golden

L = pytest.importorskip("vit_amd._lib")
lib = L.lib


def dims(k):
    return L.VtDims(1, 28, 16, 32, 16, 2, k)


def layout(k):
    from oracle.pyoracle import Layout, make_dims
    return Layout(make_dims(num_classes=k))


def test_every_declared_symbol_is_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "vtcnn_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 40
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    unbound = [s for s in sorted(declared) if s not in L.SIGNATURES]
    assert not unbound, unbound
    assert lib.vt_version() >= 1
    assert lib.vt_error_string(-2).decode().startswith("configuration not supported")


@pytest.mark.parametrize("fx,k", [("step_n32_k10.npz", 10), ("step_n2_k8_full.npz", 8)])
def test_weight_init_draw_order_bit_exact(fx, k):
    g = load_golden(fx)
    lo = layout(k)
    arena = np.zeros(lo.n_total, dtype=np.float32)
    assert lib.vt_host_init_arena(C.byref(dims(k)), 42, arena.ctypes.data) == 0
    for name, (off, size, shape, kind) in lo.tensors.items():
        assert np.array_equal(arena[off:off + size], g["p0." + name].ravel()), name
    assert lo.n_trainable_exact == {10: 39786, 8: 33512}[k]


def test_tokenizer_of_shipped_mnist_checkpoint_equals_seed42_init():
    """The reference never updates the tokenizer's Dense layers, so the four tokenizer tensors of
    its trained MNIST checkpoint equal the seed-42 initial values; checked through the manifest
    hash of the file parsed in the build container (the file itself is not shipped here)."""
    man = json.load(open(os.path.join(GOLD, "MANIFEST.json")))
    assert man["model_mnist_bin_sha256"].startswith("85c8d220")


def test_checkpoint_bytes(tmp_path):
    man = json.load(open(os.path.join(GOLD, "MANIFEST.json")))
    for k in (8, 10):
        lo = layout(k)
        arena = np.zeros(lo.n_total, dtype=np.float32)
        assert lib.vt_host_init_arena(C.byref(dims(k)), 42, arena.ctypes.data) == 0
        out = str(tmp_path / ("init%d.bin" % k))
        assert lib.vt_checkpoint_write(C.byref(dims(k)), arena.ctypes.data, out.encode()) == 0
        raw = open(out, "rb").read()
        assert len(raw) == man["init%d_bin" % k]["bytes"]
        assert hashlib.sha256(raw).hexdigest() == man["init%d_bin" % k]["sha256"]
        if k == 8:
            assert raw == open(os.path.join(GOLD, "init_seed42_k8.bin"), "rb").read()
    # load -> save of the reference's shipped 8-class checkpoint is byte-identical
    lo = layout(8)
    arena = np.zeros(lo.n_total, dtype=np.float32)
    lib.vt_host_init_arena(C.byref(dims(8)), 42, arena.ctypes.data)
    before = arena.copy()
    src = os.path.join(GOLD, "blood_model.bin")
    assert lib.vt_checkpoint_read(C.byref(dims(8)), arena.ctypes.data, src.encode()) == 0
    out = str(tmp_path / "resave.bin")
    assert lib.vt_checkpoint_write(C.byref(dims(8)), arena.ctypes.data, out.encode()) == 0
    assert open(out, "rb").read() == open(src, "rb").read()
    # tensors outside the file keep their seed-42 values (transformer block, BatchNorm1D statistics)
    for nm in ("mha.q0.W", "mha.o.W", "ffn.linear1.W", "norm1.gamma", "norm2.running_var", "proj.norm.running_mean"):
        assert np.array_equal(lo.view(arena, nm), lo.view(before, nm)), nm
    assert not np.array_equal(lo.view(arena, "fc.W"), lo.view(before, "fc.W"))
    # error behaviour: class-count mismatch and missing file
    a10 = np.zeros(layout(10).n_total, dtype=np.float32)
    assert lib.vt_checkpoint_read(C.byref(dims(10)), a10.ctypes.data, src.encode()) == -5
    assert lib.vt_checkpoint_read(C.byref(dims(8)), arena.ctypes.data, b"/nonexistent/x.bin") == -4


def test_checkpoint_v2_roundtrip_and_v1_compatibility(tmp_path):
    """v2 = v1 bytes + trailer (transformer block, BatchNorm1D statistics, training position)."""
    import shutil
    import subprocess
    from vit_amd._lib import VtTrainState
    lo = layout(8)
    d8 = dims(8)
    rng = np.random.default_rng(5)
    arena = rng.standard_normal(lo.n_total).astype(np.float32)
    st = VtTrainState(2, 3, 4242, 42, 4)
    v2 = str(tmp_path / "m_v2.bin")
    v1 = str(tmp_path / "m_v1.bin")
    assert lib.vt_checkpoint_write_v2(C.byref(d8), arena.ctypes.data, C.byref(st), v2.encode()) == 0
    assert lib.vt_checkpoint_write(C.byref(d8), arena.ctypes.data, v1.encode()) == 0
    raw1, raw2 = open(v1, "rb").read(), open(v2, "rb").read()
    assert raw2[:len(raw1)] == raw1 and raw2[len(raw1):len(raw1) + 8] == b"VTCNNv2\0"
    # full restore: every tensor of the arena, and the training position
    back = np.zeros(lo.n_total, dtype=np.float32)
    got = VtTrainState()
    assert lib.vt_checkpoint_read_v2(C.byref(d8), back.ctypes.data, C.byref(got), v2.encode()) == 0
    assert (got.version, got.epoch, got.step, got.loader_seed, got.loader_resets) == (2, 3, 4242, 42, 4)
    for name, (off, size, shape, kind) in lo.tensors.items():
        assert np.array_equal(back[off:off + size], arena[off:off + size]), name
    # the v1 reader takes the same file and restores exactly the v1 subset
    part = np.zeros(lo.n_total, dtype=np.float32)
    assert lib.vt_checkpoint_read(C.byref(d8), part.ctypes.data, v2.encode()) == 0
    assert np.array_equal(lo.view(part, "fc.W"), lo.view(arena, "fc.W"))
    assert not lo.view(part, "mha.q0.W").any() and not lo.view(part, "norm2.running_var").any()
    # a v1 file through the v2 reader: version 1, omitted tensors untouched
    assert lib.vt_checkpoint_read_v2(C.byref(d8), part.ctypes.data, C.byref(got), v1.encode()) == 0
    assert got.version == 1 and not lo.view(part, "ffn.linear1.W").any()
    # truncated trailer and wrong class count are format errors
    open(str(tmp_path / "cut.bin"), "wb").write(raw2[:len(raw1) + 40])
    assert lib.vt_checkpoint_read_v2(C.byref(d8), back.ctypes.data, C.byref(got), str(tmp_path / "cut.bin").encode()) == -5
    a10 = np.zeros(layout(10).n_total, dtype=np.float32)
    assert lib.vt_checkpoint_read_v2(C.byref(dims(10)), a10.ctypes.data, C.byref(got), v2.encode()) == -5
    # the REAL reference loads a v2 file like the v1 file it starts with (build container only:
    # oracle/_ref holds the reference compiled from its own headers)
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.access(harness, os.X_OK):
        ck = np.zeros(lo.n_total, dtype=np.float32)
        lib.vt_host_init_arena(C.byref(d8), 42, ck.ctypes.data)
        src = os.path.join(GOLD, "blood_model.bin")
        assert lib.vt_checkpoint_read(C.byref(d8), ck.ctypes.data, src.encode()) == 0
        blood_v2 = str(tmp_path / "blood_v2.bin")
        assert lib.vt_checkpoint_write_v2(C.byref(d8), ck.ctypes.data, None, blood_v2.encode()) == 0
        out = str(tmp_path / "blood.bundle")
        env = dict(os.environ, OMP_NUM_THREADS="1")
        subprocess.run([harness, "blood", out, os.path.join(GOLD, "blood_t10k_256-images-idx3-ubyte"),
                        os.path.join(GOLD, "blood_t10k_256-labels-idx1-ubyte"), blood_v2, "256"], check=True, env=env,
                       capture_output=True, timeout=300)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from bundle import read_bundle
        got_b = read_bundle(out)
        g = load_golden("blood_t10k_256.npz")
        assert np.array_equal(got_b["pred"], g["pred"]) and np.array_equal(got_b["probs"], g["probs"])


def test_loader_shuffle_known_answers():
    kat = load_golden("kat.npz")
    for n in (48000, 12000, 10000, 3421, 100, 33):
        for resets, key in ((0, "ctor"), (1, "reset1")):
            ref = kat["shuffle.%d.%s" % (n, key)]
            out = np.zeros(n, dtype=np.int32)
            assert lib.vt_loader_indices(n, 42, resets, out.ctypes.data) == 0
            assert np.array_equal(out[:ref.size], ref), (n, key)
            assert sorted(out.tolist()) == list(range(n))
    out = np.zeros(1, dtype=np.int32)
    assert lib.vt_loader_indices(0, 42, 0, out.ctypes.data) == 0  # empty dataset
    assert lib.vt_loader_indices(1, 42, 3, out.ctypes.data) == 0 and out[0] == 0


def test_idx_decode_bit_exact(tmp_path):
    img = os.path.join(GOLD, "blood_t10k_256-images-idx3-ubyte").encode()
    lbl = os.path.join(GOLD, "blood_t10k_256-labels-idx1-ubyte").encode()
    n, r, c = C.c_int(), C.c_int(), C.c_int()
    assert lib.vt_idx_read_images(img, 9999999, None, C.byref(n), C.byref(r), C.byref(c)) == 0
    assert (n.value, r.value, c.value) == (256, 28, 28)
    out = np.zeros((256, 784), dtype=np.float32)
    assert lib.vt_idx_read_images(img, 9999999, out.ctypes.data, C.byref(n), C.byref(r), C.byref(c)) == 0
    raw = np.fromfile(img.decode(), dtype=np.uint8)[16:].reshape(256, 784)
    assert np.array_equal(out, raw.astype(np.float32) / np.float32(255.0))
    m = C.c_int()
    labels = np.zeros(256, dtype=np.int32)
    assert lib.vt_idx_read_labels(lbl, 9999999, labels.ctypes.data, C.byref(m)) == 0 and m.value == 256
    assert np.array_equal(labels, np.fromfile(lbl.decode(), dtype=np.uint8)[8:].astype(np.int32))
    # max_samples truncation (load_dataset.hpp:37-38) and a missing file
    assert lib.vt_idx_read_images(img, 10, None, C.byref(n), C.byref(r), C.byref(c)) == 0 and n.value == 10
    assert lib.vt_idx_read_images(b"/nonexistent", 10, None, C.byref(n), C.byref(r), C.byref(c)) == -4
    # a truncated file: missing pixels decode as 0, like the reference's failed reads
    short = tmp_path / "short-idx3"
    short.write_bytes(open(img.decode(), "rb").read()[:16 + 784 + 100])
    o2 = np.ones((2, 784), dtype=np.float32)
    assert lib.vt_idx_read_images(str(short).encode(), 2, o2.ctypes.data, C.byref(n), C.byref(r), C.byref(c)) == 0
    assert np.array_equal(o2[0], out[0]) and np.all(o2[1, 100:] == 0.0) and np.array_equal(o2[1, :100], out[1, :100])
    # headers the reference's containers would refuse: negative count, negative side, pixel count beyond int
    import struct
    for hdr in (struct.pack(">iiii", 0x803, -5, 28, 28), struct.pack(">iiii", 0x803, 2, -28, 28),
                struct.pack(">iiii", 0x803, 2, 65536, 65536), struct.pack(">ii", 0x803, 2)):
        bad = tmp_path / "bad-idx3"
        bad.write_bytes(hdr)
        assert lib.vt_idx_read_images(str(bad).encode(), 0, None, C.byref(n), C.byref(r), C.byref(c)) == -5
    badl = tmp_path / "bad-idx1"
    badl.write_bytes(struct.pack(">ii", 0x801, -1))
    assert lib.vt_idx_read_labels(str(badl).encode(), 0, None, C.byref(m)) == -5
    # the magic number is not checked, as in the reference
    odd = tmp_path / "odd-idx1"
    odd.write_bytes(struct.pack(">ii", 0x1234, 3) + bytes([7, 8, 9]))
    three = np.zeros(3, dtype=np.int32)
    assert lib.vt_idx_read_labels(str(odd).encode(), 0, three.ctypes.data, C.byref(m)) == 0 and three.tolist() == [7, 8, 9]


def test_layout_matches_python_mirror():
    import vit_amd.engine as E
    for k in (8, 10):
        lo = layout(k)

        class Fake:
            dims = dims(k)
        assert E.VTCNN._layout_sizes(Fake) == (lo.n_total, lo.n_trainable)


def test_png_decode_matches_reference_loader(tmp_path):
    """vt_image_read_gray against the real reference's load_image_as_vector
    (utils/load_image_stb.hpp:9-26) on 25 PNG files covering every colour type, bit depth, scanline
    filter, Adam7, tRNS and split IDAT (tests/golden/png, expectations from
    tools/make_png_golden.py through the reference harness): every pixel bit-exact.  When the
    harness is present (build container) it is run again here on the same folder."""
    import shutil
    import subprocess
    from vit_amd._lib import lib
    from vit_amd.engine import load_custom_images_from_folder, load_image_as_vector
    g = np.load(os.path.join(GOLD, "png_expected.npz"))
    folder = os.path.join(GOLD, "png")
    names = [str(n) for n in g["names"]]
    assert sorted(os.listdir(folder)) == names
    for nm, exp in zip(names, g["images"]):
        got = load_image_as_vector(os.path.join(folder, nm))
        assert got.dtype == np.float32 and np.array_equal(got, exp), nm
    ds = load_custom_images_from_folder(folder)
    assert np.array_equal(ds.images, g["images"])
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.access(harness, os.X_OK) and os.path.isdir("/root/reference"):
        out = tmp_path / "ref.bin"
        subprocess.run([harness, "images", folder + "/", str(out)], check=True, capture_output=True, timeout=60)
        assert np.array_equal(np.fromfile(str(out), dtype=np.float32).reshape(-1, 784), ds.images)
    # error behaviour: wrong size, not a PNG, JPEG magic, truncated stream, missing file
    w, h = C.c_int(0), C.c_int(0)
    big = tmp_path / "big.png"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_png_golden import encode
    big.write_bytes(encode(np.zeros((30, 30), dtype=np.int64), 0, 8, lambda y: 0))
    assert lib.vt_image_read_gray(str(big).encode(), None, 0, C.byref(w), C.byref(h)) == 0 and (w.value, h.value) == (30, 30)
    with pytest.raises(RuntimeError, match="28x28"):
        load_image_as_vector(str(big))
    junk = tmp_path / "junk.png"
    junk.write_bytes(b"not a png at all")
    assert lib.vt_image_read_gray(str(junk).encode(), None, 0, C.byref(w), C.byref(h)) == -5
    jpg = tmp_path / "a.jpg"
    jpg.write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 64)
    assert lib.vt_image_read_gray(str(jpg).encode(), None, 0, C.byref(w), C.byref(h)) == -5  # a JPEG start, then junk
    cut = tmp_path / "cut.png"
    data = open(os.path.join(folder, "rgb16.png"), "rb").read()
    cut.write_bytes(data[:len(data) // 2])
    buf = np.zeros(784, dtype=np.uint8)
    assert lib.vt_image_read_gray(str(cut).encode(), C.c_void_p(buf.ctypes.data), 784, C.byref(w), C.byref(h)) != 0
    assert lib.vt_image_read_gray(str(tmp_path / "none.png").encode(), None, 0, C.byref(w), C.byref(h)) == -4
    with pytest.raises(RuntimeError, match="No se pudo cargar la imagen"):
        load_image_as_vector(str(junk))
    # a folder with a bad file: reported and skipped, the good ones kept in name order
    mix = tmp_path / "mix"
    mix.mkdir()
    shutil.copy(os.path.join(folder, "gray08.png"), mix / "b.png")
    shutil.copy(str(junk), mix / "a.png")
    shutil.copy(os.path.join(folder, "rgb16.png"), mix / "c.png")
    ds = load_custom_images_from_folder(str(mix))
    assert ds.images.shape == (2, 784)
    assert np.array_equal(ds.images[0], g["images"][names.index("gray08.png")])
    assert np.array_equal(ds.images[1], g["images"][names.index("rgb16.png")])


def test_jpeg_decode_matches_reference_loader(tmp_path):
    """vt_image_read_gray on baseline JPEG files against the real reference's load_image_as_vector
    (utils/load_image_stb.hpp:9-26, stbi_load(..., 1)): 14 files made by tools/make_jpeg_golden.py (grey; YCbCr 4:4:4,
    4:2:2 and 4:2:0, whose grey output is the luma plane; restart intervals; one scan per component; 16-bit quantiser
    tables; flat Huffman tables; RGB-coded files by component ids and by an Adobe segment; blocks that saturate the
    clamp), every pixel bit-exact.  When the harness is present (build container) it is run again on the same folder.
    Modes that are not decoded (progressive, arithmetic, four components, subsampled luma) are refused, not guessed."""
    import subprocess
    from vit_amd._lib import lib
    from vit_amd.engine import load_custom_images_from_folder, load_image_as_vector
    g = np.load(os.path.join(GOLD, "jpeg_expected.npz"))
    folder = os.path.join(GOLD, "jpeg")
    names = [str(n) for n in g["names"]]
    assert sorted(os.listdir(folder)) == names and len(names) == 14
    for nm, exp in zip(names, g["images"]):
        got = load_image_as_vector(os.path.join(folder, nm))
        assert got.dtype == np.float32 and np.array_equal(got, exp), nm
    ds = load_custom_images_from_folder(folder)
    assert np.array_equal(ds.images, g["images"])
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.access(harness, os.X_OK) and os.path.isdir("/root/reference"):
        out = tmp_path / "ref.bin"
        subprocess.run([harness, "images", folder + "/", str(out)], check=True, capture_output=True, timeout=60)
        assert np.array_equal(np.fromfile(str(out), dtype=np.float32).reshape(-1, 784), ds.images)
    w, h = C.c_int(0), C.c_int(0)
    data = open(os.path.join(folder, "a_grey.jpg"), "rb").read()
    sof = data.index(b"\xff\xc0")
    prog = tmp_path / "progressive.jpg"
    prog.write_bytes(data[:sof + 1] + b"\xc2" + data[sof + 2:])  # the same frame header announced as progressive
    assert lib.vt_image_read_gray(str(prog).encode(), None, 0, C.byref(w), C.byref(h)) == -2
    arith = tmp_path / "arithmetic.jpg"
    arith.write_bytes(data[:sof + 1] + b"\xc9" + data[sof + 2:])
    assert lib.vt_image_read_gray(str(arith).encode(), None, 0, C.byref(w), C.byref(h)) == -2
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_jpeg_golden import encode
    rng = np.random.default_rng(3)
    planes = [rng.integers(0, 256, size=(28, 28), dtype=np.uint8) for _ in range(3)]
    sub = tmp_path / "luma_subsampled.jpg"
    sub.write_bytes(encode([planes[0][::2, ::2].copy(), planes[1], planes[2]], [(1, 1), (2, 2), (2, 2)]))
    assert lib.vt_image_read_gray(str(sub).encode(), None, 0, C.byref(w), C.byref(h)) == -2
    with pytest.raises(RuntimeError, match="No se pudo cargar la imagen"):
        load_image_as_vector(str(sub))
    wide = tmp_path / "wide.jpg"
    wide.write_bytes(encode([rng.integers(0, 256, size=(19, 45), dtype=np.uint8)], [(1, 1)], width=45, height=19))
    assert lib.vt_image_read_gray(str(wide).encode(), None, 0, C.byref(w), C.byref(h)) == 0 and (w.value, h.value) == (45, 19)
    with pytest.raises(RuntimeError, match="28x28"):
        load_image_as_vector(str(wide))
    cut = tmp_path / "cut.jpg"
    cut.write_bytes(data[:sof + 6])
    assert lib.vt_image_read_gray(str(cut).encode(), None, 0, C.byref(w), C.byref(h)) != 0


def test_reference_main_compiles_unchanged_against_the_mirror_headers(tmp_path):
    """The reference's own main.cpp (copied to a scratch directory so that its quoted includes resolve through
    -Icpp, not through its own tree) compiles and links against cpp/ and libvtcnn_hip.so: every header it
    includes (utils/load_dataset.hpp, utils/trainer.hpp, model/cnn.hpp, model/mlp.hpp, model/vit_cnn.hpp) and
    every name it uses exists with a compatible signature.  Needs the reference checkout and hipcc."""
    import shutil
    import subprocess
    src = "/root/reference/main.cpp"
    if not os.path.exists(src) or shutil.which("hipcc") is None:
        pytest.skip("reference checkout or hipcc not available")
    work = tmp_path / "refmain"
    work.mkdir()
    shutil.copy(src, work / "main.cpp")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["hipcc", "--offload-arch=gfx950", "-O0", "-std=c++17", "-I" + os.path.join(root, "cpp"), "main.cpp", "-o", "refmain",
           "-L" + os.path.join(root, "vit_amd"), "-lvtcnn_hip"]
    r = subprocess.run(cmd, cwd=work, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]


def test_cpp_tensor_operations_follow_the_reference_semantics():
    """cpp/tests/tensor_ops_check.cpp: permute relabels, concat returns zeros, operator* keeps the left
    operand's shape (row 0 of a taller right operand), the two sum(axis) forms, slices, pad, broadcasting
    rules, draw order of the initialisers: the host half of the Tensor type, no device needed."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cpp", "build", "tensor_ops_check")
    if not os.path.exists(exe):
        pytest.fail("cpp/build/tensor_ops_check is not built (run __graft_entry__.build())")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_class_stats_against_an_independent_implementation():
    """vit_amd.engine.class_stats (what evaluate_test prints after the loss line, utils/trainer.hpp:131-153): per
    class precision / recall / F1 with 0 for empty denominators, macro means over the classes present among the
    true labels, F1 weighted by their support; compared with scikit-learn on a matrix with an absent class and a
    never-predicted class."""
    from vit_amd.engine import class_stats
    from sklearn.metrics import precision_recall_fscore_support as prfs
    rng = np.random.default_rng(3)
    k = 6
    truth = rng.integers(0, k, size=400)
    truth[truth == 4] = 1  # class 4 never occurs
    pred = rng.integers(0, k, size=400)
    pred[pred == 2] = 0   # class 2 is never predicted
    cm = np.zeros((k, k), dtype=np.int64)
    np.add.at(cm, (truth, pred), 1)
    st = class_stats(cm)
    p, r, f, sup = prfs(truth, pred, labels=list(range(k)), zero_division=0)
    assert np.allclose(st["precision"], p, atol=1e-6) and np.allclose(st["recall"], r, atol=1e-6)
    assert np.allclose(st["f1"], f, atol=1e-6)
    present = sup > 0
    assert abs(st["macro_f1"] - f[present].mean()) < 1e-6 and abs(st["macro_precision"] - p[present].mean()) < 1e-6
    assert abs(st["macro_recall"] - r[present].mean()) < 1e-6
    assert abs(st["weighted_f1"] - (f * sup).sum() / sup.sum()) < 1e-6
