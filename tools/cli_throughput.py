"""Images/sec of the reference-style CLI (`cpp/build/main bloodmnist train E`) on a synthetic dataset: the
256-image BloodMNIST slice of tests/golden repeated to 60 000 training and 10 000 test images, as IDX files.
Prints the CLI's own epoch lines for the device-resident loop and for the reference-shaped host loop.

    python tools/cli_throughput.py [epochs] [batch_size]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batch = sys.argv[2] if len(sys.argv) > 2 else "32"

img = np.fromfile(os.path.join(GOLD, "blood_t10k_256-images-idx3-ubyte"), dtype=np.uint8)[16:].reshape(256, 784)
lbl = np.fromfile(os.path.join(GOLD, "blood_t10k_256-labels-idx1-ubyte"), dtype=np.uint8)[8:]
with tempfile.TemporaryDirectory() as tmp:
    os.mkdir(os.path.join(tmp, "blood_data"))
    for stem, n in (("train", 60000), ("t10k", 10000)):
        reps = (n + 255) // 256
        with open(os.path.join(tmp, "blood_data", stem + "-images-idx3-ubyte"), "wb") as f:
            f.write(struct.pack(">iiii", 0x803, n, 28, 28) + np.tile(img, (reps, 1))[:n].tobytes())
        with open(os.path.join(tmp, "blood_data", stem + "-labels-idx1-ubyte"), "wb") as f:
            f.write(struct.pack(">ii", 0x801, n) + np.tile(lbl, reps)[:n].tobytes())
    for mode, extra in (("device-resident loop", {}), ("host loop (reference shape)", {"VTCNN_HOST_LOOP": "1"})):
        env = dict(os.environ, VTCNN_DATA_ROOT=tmp, VTCNN_NUM_CLASSES="8", VTCNN_BATCH_SIZE=batch, **extra)
        r = subprocess.run([os.path.join(ROOT, "cpp", "build", "main"), "bloodmnist", "train", str(epochs)], env=env,
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print(mode, "FAILED", r.stderr[-500:])
            continue
        print("== %s, batch %s" % (mode, batch))
        for line in r.stdout.splitlines():
            if line.startswith(("Train Loss", "Val Loss", "[TEST]")):
                print("  ", line)
