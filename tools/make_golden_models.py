#!/usr/bin/env python3
"""Generates tests/golden/models_step_n8.npz from the real reference's CNN and MLP models (model/cnn.hpp,
model/mlp.hpp; oracle/_ref/ref_harness models, OMP_NUM_THREADS=1): one training step each on a synthetic batch,
weights before / gradients / weights after / probabilities.  Build container only."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bundle import read_bundle  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def main():
    tmp = tempfile.mkdtemp()
    p = os.path.join(tmp, "models.bundle")
    subprocess.check_call([HARNESS, "models", p, "8", "321"], env=dict(os.environ, OMP_NUM_THREADS="1"))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "models_step_n8.npz"), **read_bundle(p))
    print("models ok")


if __name__ == "__main__":
    main()
