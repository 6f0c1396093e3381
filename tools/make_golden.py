#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the real reference (oracle/_ref/ref_harness).

Runs only in the build container, where the reference checkout is mounted; the fixtures it
writes are committed so that the GPU box (which has no reference) can check against them.
All runs use OMP_NUM_THREADS=1 and the `-O2 -ffp-contract=off` harness build, so the reference's
thread-count dependent float reductions are deterministic.

Usage: python tools/make_golden.py [--ref /root/reference]
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bundle import read_bundle  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")


def run(*args):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.check_call([HARNESS] + [str(a) for a in args], env=env)


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def make_traj(p):
    # twelve consecutive training steps at the reference batch size, K=8: per-step loss and labels,
    # final weights (inputs kept as the u8 pixel values they were drawn as)
    run("traj", p, 32, 8, 500, 12)
    b = read_bundle(p)
    t = {"traj.loss": b["traj.loss"], "traj.argmax": b["traj.argmax"],
         "traj.x_u8": np.stack([np.rint(b["traj.x%d" % k] * 255.0).astype(np.uint8) for k in range(12)]),
         "traj.y": np.stack([b["traj.y%d" % k] for k in range(12)]).astype(np.int32)}
    assert all(np.array_equal(t["traj.x_u8"][k].astype(np.float32) / np.float32(255.0), b["traj.x%d" % k]) for k in range(12))
    t.update({k: v for k, v in b.items() if k.startswith("pT.")})
    np.savez_compressed(os.path.join(GOLD, "traj_n32_k8_t12.npz"), **t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", choices=("traj",), default=None, help="regenerate just this fixture")
    a = ap.parse_args()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "REF=" + a.ref])
    if a.only == "traj":
        tmp = tempfile.mkdtemp()
        make_traj(os.path.join(tmp, "a.bundle"))
        shutil.rmtree(tmp)
        return
    os.makedirs(GOLD, exist_ok=True)
    tmp = tempfile.mkdtemp()
    manifest = {"harness_flags": "-O2 -ffp-contract=off -fopenmp, OMP_NUM_THREADS=1", "files": {}}

    # one training step, every intermediate: N=2, K=8 (BloodMNIST head)
    p = os.path.join(tmp, "a.bundle")
    run("step", p, 2, 8, 123, 1)
    np.savez(os.path.join(GOLD, "step_n2_k8_full.npz"), **read_bundle(p))
    # one training step at the reference batch size, K=10: outputs, gradients, updated weights
    run("step", p, 32, 10, 123, 0)
    b = read_bundle(p)
    b = {k: v for k, v in b.items() if not (k.startswith("p2.") and "running" not in k)}
    np.savez(os.path.join(GOLD, "step_n32_k10.npz"), **b)
    make_traj(p)
    # a batch that is not a multiple of anything convenient, K=8
    run("step", p, 5, 8, 7, 0)
    b = read_bundle(p)
    b = {k: v for k, v in b.items() if not k.startswith(("p0.", "p2."))}
    np.savez(os.path.join(GOLD, "step_n5_k8.npz"), **b)
    # data-parallel oracle: 2 shards of 4 images, gradients per shard
    run("dp", p, 2, 4, 8, 500)
    np.savez(os.path.join(GOLD, "dp_g2_n4_k8.npz"), **read_bundle(p))
    # shuffle / init / checkpoint KATs
    run("kat", p)
    k = dict(read_bundle(p))
    for K in (10, 8):
        f = p + ".init%d.bin" % K
        manifest["init%d_bin" % K] = {"sha256": sha256(f), "bytes": os.path.getsize(f)}
    shutil.copy(p + ".init8.bin", os.path.join(GOLD, "init_seed42_k8.bin"))
    np.savez(os.path.join(GOLD, "kat.npz"), **k)

    # real-data KAT: the shipped 8-class checkpoint over (a slice of) the shipped BloodMNIST test split
    img = os.path.join(a.ref, "blood_data", "t10k-images-idx3-ubyte")
    lbl = os.path.join(a.ref, "blood_data", "t10k-labels-idx1-ubyte")
    ckpt = os.path.join(a.ref, "save_models", "blood_model.bin")
    run("blood", p, img, lbl, ckpt, 3421)
    full = read_bundle(p)
    np.savez(os.path.join(GOLD, "blood_t10k_full_labels.npz"), pred=full["pred"].astype(np.uint8),
             truth=full["truth"].astype(np.uint8), order=full["order"], loss=full["loss"], acc=full["acc"])
    # 256-image slice: data fixture + labels predicted by the reference on exactly that slice
    n = 256
    with open(img, "rb") as f:
        raw = f.read(16 + n * 784)
    raw = (2051).to_bytes(4, "big") + n.to_bytes(4, "big") + raw[8:]
    with open(os.path.join(GOLD, "blood_t10k_256-images-idx3-ubyte"), "wb") as f:
        f.write(raw)
    with open(lbl, "rb") as f:
        rawl = f.read(8 + n)
    rawl = (2049).to_bytes(4, "big") + n.to_bytes(4, "big") + rawl[8:]
    with open(os.path.join(GOLD, "blood_t10k_256-labels-idx1-ubyte"), "wb") as f:
        f.write(rawl)
    shutil.copy(ckpt, os.path.join(GOLD, "blood_model.bin"))
    run("blood", p, os.path.join(GOLD, "blood_t10k_256-images-idx3-ubyte"),
        os.path.join(GOLD, "blood_t10k_256-labels-idx1-ubyte"), ckpt, n)
    np.savez(os.path.join(GOLD, "blood_t10k_256.npz"), **read_bundle(p))
    manifest["blood_model_bin_sha256"] = sha256(ckpt)
    manifest["model_mnist_bin_sha256"] = sha256(os.path.join(a.ref, "save_models", "model_mnist.bin"))
    for fn in sorted(os.listdir(GOLD)):
        if fn != "MANIFEST.json":
            manifest["files"][fn] = os.path.getsize(os.path.join(GOLD, fn))
    with open(os.path.join(GOLD, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)
    print(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()
