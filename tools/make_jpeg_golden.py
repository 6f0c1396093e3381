"""Writes tests/golden/jpeg/*.jpg (28x28 baseline JPEG files made by the small encoder below: grey, YCbCr 4:4:4 and
4:2:0, restart intervals, one scan per component, 16-bit quantiser tables, RGB-coded files, saturating blocks) and
tests/golden/jpeg_expected.npz: what the REAL reference's custom-image loader returns for each of them
(utils/load_image_stb.hpp through oracle/_ref/ref_harness `images`), in the loader's alphabetical order.
Run in the build container (needs /root/reference for the harness)."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np
from scipy.fft import dctn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "jpeg")

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
          62, 63]
QLUM = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87,
        80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92,
        95, 98, 112, 100, 103, 99]
# ITU T.81 Annex K.3 luminance tables (code counts per length 1..16, then the symbols)
DC_BITS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
DC_VALS = list(range(12))
AC_BITS = [0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d]
AC_VALS = [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14,
           0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09,
           0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a,
           0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65,
           0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88,
           0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9,
           0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
           0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea,
           0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa]
assert sum(AC_BITS) == len(AC_VALS) == 162 and len(set(AC_VALS)) == 162
# a second, flat pair of tables: every symbol on its own fixed-length code (DC 4 bits, AC 8 bits)
FLAT_DC_BITS = [0, 0, 0, 12] + [0] * 12
FLAT_AC_VALS = [0x00, 0xf0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]
FLAT_AC_BITS = [0] * 7 + [len(FLAT_AC_VALS)] + [0] * 8


def codes_of(bits, vals):
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            out[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return out


class Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, code, length):
        self.acc = (self.acc << length) | (code & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)  # pad with ones
        data, self.out = bytes(self.out), bytearray()
        return data


def category(v):
    return 0 if v == 0 else int(abs(v)).bit_length()


def amplitude(v, cat):
    return v if v >= 0 else v + (1 << cat) - 1


def encode_block(bw, coef, pred, dc_codes, ac_codes):
    diff = int(coef[0]) - pred
    cat = category(diff)
    bw.put(*dc_codes[cat])
    if cat:
        bw.put(amplitude(diff, cat), cat)
    run = 0
    last = max([k for k in range(1, 64) if coef[k] != 0], default=0)
    for k in range(1, last + 1):
        v = int(coef[k])
        if v == 0:
            run += 1
            continue
        while run > 15:
            bw.put(*ac_codes[0xF0])
            run -= 16
        cat = category(v)
        bw.put(*ac_codes[(run << 4) | cat])
        bw.put(amplitude(v, cat), cat)
        run = 0
    if last < 63:
        bw.put(*ac_codes[0x00])
    return int(coef[0])


def quantised_blocks(plane, q):
    """plane [h8, w8] uint8 (multiples of 8) -> [h8/8, w8/8, 64] quantised coefficients in zig-zag order."""
    h, w = plane.shape
    out = np.zeros((h // 8, w // 8, 64), dtype=np.int64)
    qn = np.asarray(q, dtype=np.float64).reshape(8, 8)
    for by in range(h // 8):
        for bx in range(w // 8):
            blk = plane[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8].astype(np.float64) - 128.0
            c = np.rint(dctn(blk, norm="ortho") / qn).astype(np.int64).ravel()
            c = np.clip(c, -1023, 1023)
            out[by, bx] = c[ZIGZAG]
    return out


def pad_plane(p, h2, w2):
    out = np.zeros((h2, w2), dtype=np.uint8)
    h, w = p.shape
    out[:h, :w] = p
    out[h:, :w] = p[-1:, :]
    out[:, w:] = out[:, w - 1:w]
    return out


def seg(marker, payload):
    return bytes([0xFF, marker]) + struct.pack(">H", len(payload) + 2) + payload


def encode(planes, sampling, quality_scale=1.0, restart=0, interleaved=True, q16=False, flat=False, ids=None, jfif=True,
           adobe=None, width=28, height=28):
    """planes: list of [h_c, w_c] uint8 arrays already at their sampled size; sampling: [(h, v)] per component."""
    n = len(planes)
    ids = ids or list(range(1, n + 1))
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    mcux, mcuy = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    q = [max(1, min(65535 if q16 else 255, int(round(v * quality_scale)))) for v in QLUM]
    dc_bits, dc_vals, ac_bits, ac_vals = (FLAT_DC_BITS, DC_VALS, FLAT_AC_BITS, FLAT_AC_VALS) if flat else (DC_BITS, DC_VALS, AC_BITS, AC_VALS)
    dc_codes, ac_codes = codes_of(dc_bits, dc_vals), codes_of(ac_bits, ac_vals)
    blocks = []
    for p, (h, v) in zip(planes, sampling):
        blocks.append(quantised_blocks(pad_plane(p, mcuy * v * 8, mcux * h * 8), q))
    out = bytearray(b"\xFF\xD8")
    if jfif:
        out += seg(0xE0, b"JFIF\0" + bytes([1, 1, 0, 0, 1, 0, 1, 0, 0]))
    if adobe is not None:
        out += seg(0xEE, b"Adobe\0" + bytes([100, 0, 0, 0, 0, adobe]))
    qt = bytes([0x10]) + b"".join(struct.pack(">H", q[ZIGZAG[i]]) for i in range(64)) if q16 else bytes([0x00]) + bytes(q[ZIGZAG[i]] for i in range(64))
    out += seg(0xDB, qt)
    out += seg(0xC0, bytes([8]) + struct.pack(">HH", height, width) + bytes([n]) +
               b"".join(bytes([ids[k], (sampling[k][0] << 4) | sampling[k][1], 0]) for k in range(n)))
    out += seg(0xC4, bytes([0x00]) + bytes(dc_bits) + bytes(dc_vals) + bytes([0x10]) + bytes(ac_bits) + bytes(ac_vals))
    if restart:
        out += seg(0xDD, struct.pack(">H", restart))

    def scan(comps):
        nonlocal out
        out += seg(0xDA, bytes([len(comps)]) + b"".join(bytes([ids[k], 0x00]) for k in comps) + bytes([0, 63, 0]))
        bw = Bits()
        pred = {k: 0 for k in comps}
        units, rst = 0, 0

        def unit_done():
            nonlocal units, rst, out
            units += 1
            if restart and units % restart == 0:
                out += bw.flush()
                return True
            return False

        def restart_marker(more):
            nonlocal rst, out
            if more:
                out += bytes([0xFF, 0xD0 + (rst & 7)])
                rst += 1
                for k in pred:
                    pred[k] = 0

        if len(comps) == 1:
            k = comps[0]
            h, v = sampling[k]
            cx, cy = -(-(width * h) // hmax), -(-(height * v) // vmax)
            bwid, bhei = (cx + 7) // 8, (cy + 7) // 8
            for j in range(bhei):
                for i in range(bwid):
                    pred[k] = encode_block(bw, blocks[k][j, i], pred[k], dc_codes, ac_codes)
                    if unit_done():
                        restart_marker(not (j == bhei - 1 and i == bwid - 1))
        else:
            for j in range(mcuy):
                for i in range(mcux):
                    for k in comps:
                        h, v = sampling[k]
                        for y in range(v):
                            for x in range(h):
                                pred[k] = encode_block(bw, blocks[k][j * v + y, i * h + x], pred[k], dc_codes, ac_codes)
                    if unit_done():
                        restart_marker(not (j == mcuy - 1 and i == mcux - 1))
        out += bw.flush()

    if interleaved or n == 1:
        scan(list(range(n)))
    else:
        for k in range(n):
            scan([k])
    out += b"\xFF\xD9"
    return bytes(out)


def main():
    os.makedirs(OUT, exist_ok=True)
    raw = np.fromfile(os.path.join(ROOT, "tests", "golden", "blood_t10k_256-images-idx3-ubyte"), dtype=np.uint8)[16:].reshape(-1, 28, 28)
    rng = np.random.default_rng(7)
    g0, g1, g2 = raw[0], raw[5], raw[9]
    noise = rng.integers(0, 256, size=(28, 28), dtype=np.uint8)
    stripes = np.where((np.arange(28)[None, :] // 2 + np.arange(28)[:, None]) % 2 == 0, 255, 0).astype(np.uint8)  # saturating blocks
    cb = rng.integers(0, 256, size=(28, 28), dtype=np.uint8)
    cr = np.roll(g1, 3, axis=1)
    half = lambda p: p[::2, ::2].copy()  # noqa: E731
    files = {
        "a_grey.jpg": encode([g0], [(1, 1)]),
        "b_grey_fine.jpg": encode([g1], [(1, 1)], quality_scale=0.1),
        "c_grey_coarse_flat_tables.jpg": encode([g2], [(1, 1)], quality_scale=3.0, flat=True),
        "d_grey_noise_restart2.jpg": encode([noise], [(1, 1)], quality_scale=0.25, restart=2),
        "e_grey_stripes_saturating.jpg": encode([stripes], [(1, 1)], quality_scale=0.05),
        "f_grey_q16.jpeg": encode([g0], [(1, 1)], quality_scale=0.5, q16=True),
        "g_ycc444.jpg": encode([g1, cb, cr], [(1, 1)] * 3),
        "h_ycc420.jpg": encode([g2, half(cb), half(cr)], [(2, 2), (1, 1), (1, 1)], quality_scale=0.5),
        "i_ycc422_restart1.jpg": encode([noise, cb[:, ::2].copy(), cr[:, ::2].copy()], [(2, 1), (1, 1), (1, 1)], quality_scale=0.3, restart=1),
        "j_ycc444_three_scans.jpg": encode([g0, cb, cr], [(1, 1)] * 3, interleaved=False, restart=3),
        "k_rgb_ids.jpg": encode([g0, g1, g2], [(1, 1)] * 3, quality_scale=0.2, ids=[ord("R"), ord("G"), ord("B")]),
        "l_rgb_adobe.jpg": encode([noise, g1, cb], [(1, 1)] * 3, quality_scale=0.2, jfif=False, adobe=0),
        "m_ycc_adobe_transform1.jpg": encode([g2, cb, cr], [(1, 1)] * 3, jfif=False, adobe=1),
        "n_no_jfif_no_adobe.jpg": encode([g1, cb, cr], [(1, 1)] * 3, jfif=False),
    }
    for old in os.listdir(OUT):
        os.remove(os.path.join(OUT, old))
    for name, data in files.items():
        with open(os.path.join(OUT, name), "wb") as f:
            f.write(data)
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    tmp = tempfile.mkdtemp()
    outbin = os.path.join(tmp, "images.bin")
    res = subprocess.run([harness, "images", OUT, outbin], env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True)
    sys.stderr.write(res.stderr)
    got = np.fromfile(outbin, dtype=np.float32).reshape(-1, 784)
    names = sorted(files)
    rejected = [ln.split()[3].rstrip(":") for ln in res.stderr.splitlines() if ln.startswith("Error al cargar")]
    kept = [n for n in names if not any(r.endswith(n) for r in rejected)]
    assert got.shape[0] == len(kept), (got.shape, kept, res.stderr)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "jpeg_expected.npz"), names=np.array(kept), images=got)
    print("reference loaded %d of %d files; rejected: %s" % (len(kept), len(names), rejected))


if __name__ == "__main__":
    main()
