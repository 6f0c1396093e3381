"""Writes tests/golden/png/*.png (28x28 images of every PNG colour type, bit depth, scanline filter
and interlace mode, produced by the encoder below) and tests/golden/png_expected.npz: what the REAL
reference's custom-image loader returns for each of them (utils/load_image_stb.hpp through
oracle/_ref/ref_harness `images`), in the loader's alphabetical order.  Run in the build container
(needs /root/reference for the harness)."""
import os
import struct
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "png")
CH = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def pack_rows(samples, depth):
    """samples: [h, w*channels] ints -> list of byte rows."""
    rows = []
    for r in samples:
        if depth == 8:
            rows.append(bytes(int(v) for v in r))
        elif depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in r))
        else:
            bits = "".join(format(int(v), "0%db" % depth) for v in r)
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


def filter_rows(rows, bpp, ftype_of_row):
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        ft = ftype_of_row(y)
        line = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, paeth(a, b, c))[ft]
            line[i] = (v - pred) & 0xFF
        out.append(ft)
        out += line
        prev = row
    return bytes(out)


def encode(samples, color, depth, ftype_of_row, interlace=False, palette=None, trns=None, split_idat=False):
    h, wc = samples.shape
    nc = CH[color]
    w = wc // nc
    bits = nc * depth
    bpp = max(1, bits // 8)
    body = b""
    if not interlace:
        body = filter_rows(pack_rows(samples, depth), bpp, ftype_of_row)
    else:
        px = samples.reshape(h, w, nc)
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = px[y0::dy, x0::dx]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            body += filter_rows(pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, ftype_of_row)
    z = zlib.compress(body, 6)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 1 if interlace else 0))
    png += chunk(b"gAMA", struct.pack(">I", 45455))  # ancillary chunk: must be skipped
    if palette is not None:
        png += chunk(b"PLTE", bytes(int(v) for v in palette.ravel()))
    if trns is not None:
        png += chunk(b"tRNS", bytes(trns))
    if split_idat:
        third = max(1, len(z) // 3)
        for i in range(0, len(z), third):
            png += chunk(b"IDAT", z[i:i + third])
    else:
        png += chunk(b"IDAT", z)
    return png + chunk(b"IEND", b"")


def cases():
    rng = np.random.default_rng(2024)
    w = h = 28
    mixed = lambda y: y % 5
    out = {}
    for depth in (1, 2, 4, 8, 16):
        out["gray%02d" % depth] = encode(rng.integers(0, 1 << depth, size=(h, w)), 0, depth, mixed)
    out["gray08_adam7"] = encode(rng.integers(0, 256, size=(h, w)), 0, 8, mixed, interlace=True)
    out["gray04_adam7"] = encode(rng.integers(0, 16, size=(h, w)), 0, 4, mixed, interlace=True)
    out["gray08_trns"] = encode(rng.integers(0, 256, size=(h, w)), 0, 8, mixed, trns=[0, 17])
    for ft in range(5):
        out["rgb08_filter%d" % ft] = encode(rng.integers(0, 256, size=(h, w * 3)), 2, 8, lambda y, ft=ft: ft)
    out["rgb16"] = encode(rng.integers(0, 65536, size=(h, w * 3)), 2, 16, mixed)
    out["rgb08_adam7_split"] = encode(rng.integers(0, 256, size=(h, w * 3)), 2, 8, mixed, interlace=True, split_idat=True)
    out["rgba08"] = encode(rng.integers(0, 256, size=(h, w * 4)), 6, 8, mixed)
    out["rgba16"] = encode(rng.integers(0, 65536, size=(h, w * 4)), 6, 16, mixed)
    out["graya08"] = encode(rng.integers(0, 256, size=(h, w * 2)), 4, 8, mixed)
    out["graya16"] = encode(rng.integers(0, 65536, size=(h, w * 2)), 4, 16, mixed)
    for depth in (1, 2, 4, 8):
        n = 1 << depth
        pal = rng.integers(0, 256, size=(n, 3))
        out["pal%02d" % depth] = encode(rng.integers(0, n, size=(h, w)), 3, depth, mixed, palette=pal)
    pal = rng.integers(0, 256, size=(256, 3))
    out["pal08_trns"] = encode(rng.integers(0, 256, size=(h, w)), 3, 8, mixed, palette=pal,
                               trns=list(rng.integers(0, 256, size=200)))
    # an MNIST-like digit: dark background, bright stroke (what the frontend uploads)
    digit = np.zeros((h, w), dtype=np.int64)
    digit[4:24, 13:16] = 255
    digit[4:7, 10:13] = 180
    out["digit_one"] = encode(digit, 0, 8, lambda y: 0)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, data in cases().items():
        with open(os.path.join(OUT, name + ".png"), "wb") as f:
            f.write(data)
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    with tempfile.TemporaryDirectory() as td:
        dst = os.path.join(td, "out.bin")
        r = subprocess.run([harness, "images", OUT + "/", dst], capture_output=True, text=True, check=True)
        print(r.stdout.strip(), r.stderr.strip())
        flat = np.fromfile(dst, dtype=np.float32)
    names = sorted(n + ".png" for n in cases())
    assert flat.size == 784 * len(names), (flat.size, len(names))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "png_expected.npz"), names=np.array(names),
                        images=flat.reshape(len(names), 784))
    print("wrote", len(names), "png files and png_expected.npz")


if __name__ == "__main__":
    sys.exit(main())
