#!/usr/bin/env python3
"""Generates tests/golden/selfattn_*.npz and mha_fwd_*.npz from the real reference's SelfAttention and
MultiHeadAttention layers (oracle/_ref/ref_harness selfattn / mha, OMP_NUM_THREADS=1).  Build container only."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bundle import read_bundle  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
MHA_CASES = {"tokens": (16, 32, 32, 2, 5), "ragged": (3, 45, 24, 3, 9)}
CASES = {"small": (3, 40, 24, 16, 7), "tokens": (16, 32, 32, 16, 11), "ragged": (2, 77, 10, 6, 13)}


def main():
    tmp = tempfile.mkdtemp()
    for name, (B, T, D, Dk, seed) in CASES.items():
        p = os.path.join(tmp, name + ".bundle")
        subprocess.check_call([HARNESS, "selfattn", p, str(B), str(T), str(D), str(Dk), str(seed)],
                              env=dict(os.environ, OMP_NUM_THREADS="1"))
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "selfattn_%s.npz" % name), **read_bundle(p))
        print(name, "ok")
    for name, (B, T, D, H, seed) in MHA_CASES.items():
        p = os.path.join(tmp, "mha_" + name + ".bundle")
        subprocess.check_call([HARNESS, "mha", p, str(B), str(T), str(D), str(H), str(seed)],
                              env=dict(os.environ, OMP_NUM_THREADS="1"))
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mha_fwd_%s.npz" % name), **read_bundle(p))
        print("mha", name, "ok")


if __name__ == "__main__":
    main()
