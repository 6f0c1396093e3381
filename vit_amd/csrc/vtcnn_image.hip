// Host-only: the custom-image loader of the predict path.
//
// The reference reads `custom_images/*.png|jpg|jpeg` through stbi_load(path, &w, &h, &n, 1)
// (utils/load_image_stb.hpp:9-26): an 8-bit, one-channel image, then pixel / 255.0f.  What the
// caller sees is therefore fixed by how that decoder maps each PNG colour type to one grey byte:
//   grey          -> the sample (1/2/4-bit samples scaled by 255/(2^depth - 1));
//   grey + alpha  -> the grey sample, alpha dropped;
//   RGB / RGBA / palette -> (77 r + 150 g + 29 b) >> 8 on the (palette-expanded) colour, alpha and
//                    tRNS dropped;
//   16-bit        -> the same formula on 16-bit samples, truncated to 16 bits, then the high byte.
// PNG is decoded here (zlib inflate + the five scanline filters + Adam7); baseline JPEG in vtcnn_jpeg.hip.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/vtcnn_hip.h"

namespace vt {
int jpeg_decode_gray(const std::vector<unsigned char> &file, std::vector<unsigned char> &out, int *width, int *height);
}

namespace {

uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Undoes the scanline filters of one (sub)image in place.  raw: h rows of (1 + stride) bytes.
bool unfilter(unsigned char *raw, int h, size_t stride, int bpp) {
  std::vector<unsigned char> zero(stride, 0);
  const unsigned char *prev = zero.data();
  for (int y = 0; y < h; ++y) {
    unsigned char *row = raw + (size_t)y * (stride + 1);
    const int ft = row[0];
    unsigned char *cur = row + 1;
    if (ft > 4) return false;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
      int add = 0;
      switch (ft) {
        case 1: add = a; break;
        case 2: add = b; break;
        case 3: add = (a + b) >> 1; break;
        case 4: add = paeth(a, b, c); break;
        default: break;
      }
      cur[i] = (unsigned char)(cur[i] + add);
    }
    prev = cur;
  }
  return true;
}

struct Png {
  int w = 0, h = 0, depth = 0, color = 0, interlace = 0;
  unsigned char pal[256][3];
  int npal = 0;
};

int channels(int color) { return color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4; }

// sample s (0-based, in samples) of an unfiltered scanline
inline unsigned sample(const unsigned char *line, size_t s, int depth) {
  switch (depth) {
    case 8: return line[s];
    case 16: return ((unsigned)line[2 * s] << 8) | line[2 * s + 1];
    default: {
      const size_t bit = s * depth;
      return (line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
    }
  }
}

// One grey byte from pixel x of an unfiltered scanline.
unsigned char grey_of(const Png &p, const unsigned char *line, int x) {
  const int nc = channels(p.color);
  if (p.color == 3) {
    const unsigned idx = sample(line, x, p.depth);
    if ((int)idx >= p.npal) return 0;  // an index past the palette reads as black
    return (unsigned char)((p.pal[idx][0] * 77 + p.pal[idx][1] * 150 + 29 * p.pal[idx][2]) >> 8);
  }
  if (p.color == 0 || p.color == 4) {
    const unsigned v = sample(line, (size_t)x * nc, p.depth);
    if (p.depth == 16) return (unsigned char)(v >> 8);
    if (p.depth == 8) return (unsigned char)v;
    static const unsigned scale[5] = {0, 0xff, 0x55, 0, 0x11};
    return (unsigned char)(v * scale[p.depth]);
  }
  const unsigned r = sample(line, (size_t)x * nc, p.depth), g = sample(line, (size_t)x * nc + 1, p.depth),
                 b = sample(line, (size_t)x * nc + 2, p.depth);
  if (p.depth == 16) return (unsigned char)(((uint16_t)((r * 77 + g * 150 + 29 * b) >> 8)) >> 8);
  return (unsigned char)((r * 77 + g * 150 + 29 * b) >> 8);
}

}  // namespace

extern "C" int vt_image_read_gray(const char *path, unsigned char *out, int capacity, int *width, int *height) {
  if (!path || !width || !height) return VT_EINVAL;
  FILE *f = std::fopen(path, "rb");
  if (!f) return VT_EIO;
  std::vector<unsigned char> file;
  unsigned char buf[65536];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
  std::fclose(f);
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() >= 2 && file[0] == 0xff && file[1] == 0xd8) {  // JPEG
    std::vector<unsigned char> grey;
    const int rc = vt::jpeg_decode_gray(file, grey, width, height);
    if (rc != 0 || !out) return rc;
    if ((long)grey.size() > capacity) return VT_EINVAL;
    std::memcpy(out, grey.data(), grey.size());
    return 0;
  }
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return VT_EFORMAT;
  Png p;
  std::vector<unsigned char> idat;
  bool have_ihdr = false, done = false;
  size_t pos = 8;
  while (!done && pos + 8 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const unsigned char *type = &file[pos + 4], *data = &file[pos + 8];
    if (pos + 12 + (size_t)len > file.size()) return VT_EFORMAT;
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13 || have_ihdr) return VT_EFORMAT;
      have_ihdr = true;
      p.w = (int)be32(data); p.h = (int)be32(data + 4);
      p.depth = data[8]; p.color = data[9]; p.interlace = data[12];
      const bool depth_ok = p.depth == 1 || p.depth == 2 || p.depth == 4 || p.depth == 8 || p.depth == 16;
      const bool color_ok = p.color == 0 || p.color == 2 || p.color == 3 || p.color == 4 || p.color == 6;
      if (p.w < 1 || p.h < 1 || p.w > (1 << 24) || p.h > (1 << 24) || !depth_ok || !color_ok || data[10] || data[11] ||
          p.interlace > 1)
        return VT_EFORMAT;
      if ((p.color == 3 && p.depth == 16) || ((p.color == 2 || p.color == 4 || p.color == 6) && p.depth < 8)) return VT_EFORMAT;
    } else if (!have_ihdr) {
      return VT_EFORMAT;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      if (len > 768 || len % 3) return VT_EFORMAT;
      p.npal = (int)len / 3;
      std::memcpy(p.pal, data, len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      done = true;
    } else if (!(type[0] & 0x20)) {
      return VT_EFORMAT;  // unknown critical chunk
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || idat.empty() || (p.color == 3 && p.npal == 0)) return VT_EFORMAT;
  *width = p.w;
  *height = p.h;
  if (!out) return 0;
  if ((long)p.w * p.h > capacity) return VT_EINVAL;

  const int bits = channels(p.color) * p.depth, bpp = bits >= 8 ? bits / 8 : 1;
  // Adam7 passes (x0, y0, dx, dy); a non-interlaced image is the single pass (0, 0, 1, 1)
  static const int ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1};
  static const int adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
  const int npass = p.interlace ? 7 : 1;
  size_t need = 0;
  int pw[7], ph[7];
  for (int k = 0; k < npass; ++k) {
    pw[k] = p.interlace ? (p.w - ax0[k] + adx[k] - 1) / adx[k] : p.w;
    ph[k] = p.interlace ? (p.h - ay0[k] + ady[k] - 1) / ady[k] : p.h;
    if (pw[k] > 0 && ph[k] > 0) need += (size_t)ph[k] * (1 + ((size_t)pw[k] * bits + 7) / 8);
  }
  std::vector<unsigned char> raw(need);
  uLongf rawlen = (uLongf)need;
  const int zrc = uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size());
  if ((zrc != Z_OK && zrc != Z_BUF_ERROR) || rawlen < need) return VT_EFORMAT;
  size_t off = 0;
  for (int k = 0; k < npass; ++k) {
    if (pw[k] <= 0 || ph[k] <= 0) continue;
    const size_t stride = ((size_t)pw[k] * bits + 7) / 8;
    if (!unfilter(&raw[off], ph[k], stride, bpp)) return VT_EFORMAT;
    for (int y = 0; y < ph[k]; ++y) {
      const unsigned char *line = &raw[off + (size_t)y * (stride + 1) + 1];
      for (int x = 0; x < pw[k]; ++x) {
        const int X = p.interlace ? ax0[k] + x * adx[k] : x, Y = p.interlace ? ay0[k] + y * ady[k] : y;
        out[(size_t)Y * p.w + X] = grey_of(p, line, x);
      }
    }
    off += (size_t)ph[k] * (stride + 1);
  }
  return 0;
}
