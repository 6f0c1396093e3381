// Host-only: baseline JPEG decode to one grey byte per pixel, for the custom-image loader of the predict path.
//
// The reference reads `custom_images/*.jpg|jpeg` through stbi_load(path, &w, &h, &n, 1)
// (utils/load_image_stb.hpp:9-26; the decoder is the vendored utils/stb_image.h).  What the caller sees is fixed by
// that decoder's arithmetic, which this file restates for the sequential Huffman modes (SOF0 / SOF1, 8-bit samples):
//   * entropy decoding per ITU T.81 Annex F (any number of scans, interleaved or one component each, restart
//     intervals), coefficient * quantiser truncated to 16 bits (stb_image.h:2210-2262);
//   * the 8x8 inverse DCT in its integer form: the Loeffler-Ligtenberg-Moschytz factorisation with 12-bit constants
//     round(c * 4096), a column pass that keeps two extra bits ((x + 512) >> 10) and a row pass that removes 17 bits
//     with rounding and adds the level shift ((x + 65536 + (128 << 17)) >> 17), clamped to 0..255
//     (stb_image.h:2425-2520; its SIMD form is built to give the same bits);
//   * a one-channel request on a three-component YCbCr file returns the luma plane itself (no colour conversion, the
//     chroma planes are not even resampled: stb_image.h:3876-3890, 3998-4005); on an RGB-coded file (component ids
//     'R','G','B', or an Adobe segment with transform 0 and no JFIF segment) the grey byte is (77 r + 150 g + 29 b) >> 8.
// Not decoded (VT_EUNSUPPORTED, reported to the user as the reference reports an unreadable file): progressive and
// arithmetic-coded files, four-component (CMYK / YCCK) files, files whose luma (or, for RGB, any) plane is subsampled.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/vtcnn_hip.h"

namespace vt {
int jpeg_decode_gray(const std::vector<unsigned char> &file, std::vector<unsigned char> &out, int *width, int *height);
}

namespace {

// zig-zag position -> natural (row-major) index; positions past 63, reachable only through a corrupt run
// length, all land on the last coefficient
const unsigned char kNatural[64 + 15] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
    6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
    39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman {
  bool defined = false;
  int mincode[17], maxcode[18], valptr[17];  // per code length 1..16 (Annex F.2.2.3)
  unsigned char values[256];
  bool build(const int *counts, const unsigned char *vals, int nvals) {
    std::memcpy(values, vals, nvals);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      valptr[len] = k;
      mincode[len] = code;
      code += counts[len - 1];
      k += counts[len - 1];
      maxcode[len] = counts[len - 1] ? code - 1 : -1;
      if (code > (1 << len)) return false;  // more codes of this length than the prefix property allows
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    defined = true;
    return true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0;
  int x = 0, y = 0, w2 = 0, h2 = 0;  // size in samples, and padded to whole MCUs
  int dc_pred = 0, hd = 0, ha = 0;
  std::vector<unsigned char> data;
};

struct BitReader {
  const unsigned char *p, *end;
  uint32_t buf = 0;
  int bits = 0;
  int marker = -1;  // a marker met inside the entropy-coded data: from then on the stream reads as zeros
  BitReader(const unsigned char *b, const unsigned char *e) : p(b), end(e) {}
  void reset() {
    buf = 0;
    bits = 0;
    marker = -1;
  }
  void fill() {
    while (bits <= 24) {
      int byte = 0;
      if (marker < 0 && p < end) {
        byte = *p++;
        if (byte == 0xff) {
          int c = p < end ? *p++ : 0;
          while (c == 0xff) c = p < end ? *p++ : 0;  // fill bytes
          if (c != 0) {
            marker = c;
            byte = 0;
          }
        }
      }
      buf |= (uint32_t)byte << (24 - bits);
      bits += 8;
    }
  }
  int get(int n) {  // n <= 16
    if (n == 0) return 0;
    if (bits < n) fill();
    const int v = (int)(buf >> (32 - n));
    buf <<= n;
    bits -= n;
    return v;
  }
  int bit() { return get(1); }
};

int huff_decode(BitReader &br, const Huffman &h) {
  int code = 0;
  for (int len = 1; len <= 16; ++len) {
    code = (code << 1) | br.bit();
    if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.values[h.valptr[len] + code - h.mincode[len]];
  }
  return -1;
}

// n bits as the signed value of category n (Annex F.2.2.1)
int receive_extend(BitReader &br, int n) {
  const int v = br.get(n);
  return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
}

// Even / odd decomposition of one 8-point inverse DCT pass in 12-bit fixed point (constants: round(c * 4096) for the
// rotation constants of the factorisation).  Outputs: even part x0..x3, odd part t0..t3; sample k = x_k + t_(3-k),
// sample 7-k = x_k - t_(3-k).
struct Pass8 {
  int x0, x1, x2, x3, t0, t1, t2, t3;
};
inline Pass8 idct_pass(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
  Pass8 r;
  int p1 = (s2 + s6) * 2217;
  const int e2 = p1 + s6 * -7567, e3 = p1 + s2 * 3135;
  const int e0 = (s0 + s4) * 4096, e1 = (s0 - s4) * 4096;
  r.x0 = e0 + e3;
  r.x3 = e0 - e3;
  r.x1 = e1 + e2;
  r.x2 = e1 - e2;
  int t0 = s7, t1 = s5, t2 = s3, t3 = s1;
  int p3 = t0 + t2, p4 = t1 + t3, p2 = t1 + t2;
  p1 = t0 + t3;
  const int p5 = (p3 + p4) * 4816;
  t0 *= 1223;
  t1 *= 8410;
  t2 *= 12586;
  t3 *= 6149;
  p1 = p5 + p1 * -3685;
  p2 = p5 + p2 * -10497;
  p3 *= -8034;
  p4 *= -1597;
  r.t3 = t3 + p1 + p4;
  r.t2 = t2 + p2 + p3;
  r.t1 = t1 + p2 + p4;
  r.t0 = t0 + p1 + p3;
  return r;
}
inline unsigned char clamp255(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void idct_block(unsigned char *out, int stride, const short *d) {
  int val[64];
  for (int i = 0; i < 8; ++i) {  // columns; a column whose AC terms are all zero is its DC term * 4
    const short *c = d + i;
    int *v = val + i;
    if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
      const int dc = c[0] * 4;
      for (int k = 0; k < 8; ++k) v[8 * k] = dc;
      continue;
    }
    Pass8 p = idct_pass(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
    p.x0 += 512; p.x1 += 512; p.x2 += 512; p.x3 += 512;
    v[0] = (p.x0 + p.t3) >> 10;  v[56] = (p.x0 - p.t3) >> 10;
    v[8] = (p.x1 + p.t2) >> 10;  v[48] = (p.x1 - p.t2) >> 10;
    v[16] = (p.x2 + p.t1) >> 10; v[40] = (p.x2 - p.t1) >> 10;
    v[24] = (p.x3 + p.t0) >> 10; v[32] = (p.x3 - p.t0) >> 10;
  }
  for (int i = 0; i < 8; ++i) {  // rows: 12 bits of the constants + 2 kept above + 3 of the two sqrt(8) scalings
    const int *v = val + 8 * i;
    unsigned char *o = out + (size_t)i * stride;
    Pass8 p = idct_pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    const int bias = 65536 + (128 << 17);
    p.x0 += bias; p.x1 += bias; p.x2 += bias; p.x3 += bias;
    o[0] = clamp255((p.x0 + p.t3) >> 17); o[7] = clamp255((p.x0 - p.t3) >> 17);
    o[1] = clamp255((p.x1 + p.t2) >> 17); o[6] = clamp255((p.x1 - p.t2) >> 17);
    o[2] = clamp255((p.x2 + p.t1) >> 17); o[5] = clamp255((p.x2 - p.t1) >> 17);
    o[3] = clamp255((p.x3 + p.t0) >> 17); o[4] = clamp255((p.x3 - p.t0) >> 17);
  }
}

struct Decoder {
  const unsigned char *p, *end;
  int img_x = 0, img_y = 0, ncomp = 0, h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
  int restart_interval = 0, jfif = 0, app14 = -1, rgb = 0;
  bool have_frame = false;
  Component comp[4];
  Huffman dc[4], ac[4];
  unsigned short dequant[4][64];

  int get8() { return p < end ? *p++ : 0; }
  int get16() {
    const int a = get8();
    return (a << 8) | get8();
  }
  bool eof() const { return p >= end; }

  int frame_header() {
    const int len = get16(), prec = get8();
    if (len < 11 || prec != 8) return VT_EFORMAT;
    img_y = get16();
    img_x = get16();
    ncomp = get8();
    if (img_y == 0 || img_x == 0) return VT_EFORMAT;
    if (ncomp != 1 && ncomp != 3 && ncomp != 4) return VT_EFORMAT;
    if (len != 8 + 3 * ncomp) return VT_EFORMAT;
    if (ncomp == 4) return VT_EUNSUPPORTED;
    static const char rgb_ids[3] = {'R', 'G', 'B'};
    rgb = 0;
    for (int i = 0; i < ncomp; ++i) {
      Component &c = comp[i];
      c.id = get8();
      if (ncomp == 3 && c.id == rgb_ids[i]) ++rgb;
      const int q = get8();
      c.h = q >> 4;
      c.v = q & 15;
      c.tq = get8();
      if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return VT_EFORMAT;
      h_max = c.h > h_max ? c.h : h_max;
      v_max = c.v > v_max ? c.v : v_max;
    }
    for (int i = 0; i < ncomp; ++i)
      if (h_max % comp[i].h != 0 || v_max % comp[i].v != 0) return VT_EFORMAT;
    if ((long)img_x * img_y > (1L << 28)) return VT_EFORMAT;
    mcu_x = (img_x + 8 * h_max - 1) / (8 * h_max);
    mcu_y = (img_y + 8 * v_max - 1) / (8 * v_max);
    for (int i = 0; i < ncomp; ++i) {
      Component &c = comp[i];
      c.x = (img_x * c.h + h_max - 1) / h_max;
      c.y = (img_y * c.v + v_max - 1) / v_max;
      c.w2 = mcu_x * c.h * 8;
      c.h2 = mcu_y * c.v * 8;
      c.data.assign((size_t)c.w2 * c.h2, 0);
    }
    have_frame = true;
    return 0;
  }

  int tables(int m) {  // DQT, DHT, DRI, APPn, COM
    if (m == 0xdd) {
      if (get16() != 4) return VT_EFORMAT;
      restart_interval = get16();
      return 0;
    }
    if (m == 0xdb) {
      int len = get16() - 2;
      while (len > 0) {
        const int q = get8(), sixteen = (q >> 4) != 0, t = q & 15;
        if ((q >> 4) > 1 || t > 3) return VT_EFORMAT;
        for (int i = 0; i < 64; ++i) dequant[t][kNatural[i]] = (unsigned short)(sixteen ? get16() : get8());
        len -= sixteen ? 129 : 65;
      }
      return len == 0 ? 0 : VT_EFORMAT;
    }
    if (m == 0xc4) {
      int len = get16() - 2;
      while (len > 0) {
        const int q = get8(), tc = q >> 4, th = q & 15;
        if (tc > 1 || th > 3) return VT_EFORMAT;
        int counts[16], n = 0;
        for (int i = 0; i < 16; ++i) n += counts[i] = get8();
        if (n > 256) return VT_EFORMAT;
        unsigned char vals[256];
        for (int i = 0; i < n; ++i) vals[i] = (unsigned char)get8();
        if (!(tc == 0 ? dc[th] : ac[th]).build(counts, vals, n)) return VT_EFORMAT;
        len -= 17 + n;
      }
      return len == 0 ? 0 : VT_EFORMAT;
    }
    if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) {
      int len = get16();
      if (len < 2) return VT_EFORMAT;
      len -= 2;
      if (m == 0xe0 && len >= 5) {
        static const unsigned char tag[5] = {'J', 'F', 'I', 'F', 0};
        bool ok = true;
        for (int i = 0; i < 5; ++i) ok &= get8() == tag[i];
        len -= 5;
        if (ok) jfif = 1;
      } else if (m == 0xee && len >= 12) {
        static const unsigned char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
        bool ok = true;
        for (int i = 0; i < 6; ++i) ok &= get8() == tag[i];
        len -= 6;
        if (ok) {
          get8();
          get16();
          get16();
          app14 = get8();
          len -= 6;
        }
      }
      if (p + len > end) return VT_EFORMAT;
      p += len;
      return 0;
    }
    return VT_EFORMAT;  // unknown marker
  }

  int decode_block(BitReader &br, Component &c, short *data) {
    const Huffman &hd = dc[c.hd], &ha = ac[c.ha];
    if (!hd.defined || !ha.defined) return VT_EFORMAT;
    std::memset(data, 0, 64 * sizeof(short));
    const int t = huff_decode(br, hd);
    if (t < 0 || t > 15) return VT_EFORMAT;
    const int diff = t ? receive_extend(br, t) : 0;
    c.dc_pred += diff;
    const unsigned short *dq = dequant[c.tq];
    data[0] = (short)(c.dc_pred * dq[0]);
    int k = 1;
    do {
      const int rs = huff_decode(br, ha);
      if (rs < 0) return VT_EFORMAT;
      const int s = rs & 15, r = rs >> 4;
      if (s == 0) {
        if (rs != 0xf0) break;  // end of block
        k += 16;
      } else {
        k += r;
        const int zig = kNatural[k++];
        data[zig] = (short)(receive_extend(br, s) * dq[zig]);
      }
    } while (k < 64);
    return 0;
  }

  // One scan.  On return *pending is a marker met at its end (or -1).
  int scan(int *pending) {
    const int len = get16(), n = get8();
    if (n < 1 || n > 4 || n > ncomp || len != 6 + 2 * n) return VT_EFORMAT;
    int order[4];
    for (int i = 0; i < n; ++i) {
      const int id = get8(), q = get8();
      int which = -1;
      for (int k = 0; k < ncomp; ++k)
        if (comp[k].id == id) which = k;
      if (which < 0) return VT_EFORMAT;
      comp[which].hd = q >> 4;
      comp[which].ha = q & 15;
      if (comp[which].hd > 3 || comp[which].ha > 3) return VT_EFORMAT;
      order[i] = which;
    }
    const int ss = get8(), se = get8(), a = get8();
    if (ss != 0 || (a >> 4) != 0 || (a & 15) != 0) return VT_EFORMAT;  // sequential scans only
    (void)se;
    BitReader br(p, end);
    for (int k = 0; k < 4; ++k) comp[k].dc_pred = 0;
    int todo = restart_interval ? restart_interval : 0x7fffffff;
    short data[64];
    // after `restart_interval` units a restart marker must follow; anything else ends the scan
    auto unit_done = [&]() -> int {
      if (--todo > 0) return 0;
      if (br.bits < 24) br.fill();
      if (br.marker < 0xd0 || br.marker > 0xd7) return 1;
      br.reset();
      for (int k = 0; k < 4; ++k) comp[k].dc_pred = 0;
      todo = restart_interval ? restart_interval : 0x7fffffff;
      return 0;
    };
    bool stop = false;
    if (n == 1) {  // one component: its blocks in raster order, only those that hold image samples
      Component &c = comp[order[0]];
      const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
      for (int j = 0; j < h && !stop; ++j)
        for (int i = 0; i < w && !stop; ++i) {
          const int rc = decode_block(br, c, data);
          if (rc != 0) return rc;
          idct_block(&c.data[(size_t)c.w2 * j * 8 + i * 8], c.w2, data);
          stop = unit_done() != 0;
        }
    } else {  // interleaved: per MCU, every component's h x v blocks
      for (int j = 0; j < mcu_y && !stop; ++j)
        for (int i = 0; i < mcu_x && !stop; ++i) {
          for (int k = 0; k < n; ++k) {
            Component &c = comp[order[k]];
            for (int y = 0; y < c.v; ++y)
              for (int x = 0; x < c.h; ++x) {
                const int rc = decode_block(br, c, data);
                if (rc != 0) return rc;
                idct_block(&c.data[(size_t)c.w2 * ((j * c.v + y) * 8) + (i * c.h + x) * 8], c.w2, data);
              }
          }
          stop = unit_done() != 0;
        }
    }
    if (br.bits < 24 && br.marker < 0) br.fill();  // a marker right behind the data is met here, not skipped over
    p = br.p;
    *pending = br.marker;
    return 0;
  }
};

}  // namespace

int vt::jpeg_decode_gray(const std::vector<unsigned char> &file, std::vector<unsigned char> &out, int *width, int *height) {
  Decoder d;
  d.p = file.data();
  d.end = file.data() + file.size();
  if (d.get8() != 0xff || d.get8() != 0xd8) return VT_EFORMAT;
  int pending = -1;
  auto next_marker = [&]() -> int {
    if (pending >= 0) {
      const int m = pending;
      pending = -1;
      return m;
    }
    int x = d.get8();
    if (x != 0xff) return -1;
    while (x == 0xff) x = d.get8();
    return x;
  };
  bool progressive_or_arith = false;
  int m = next_marker();
  // tables and segments up to the frame header
  while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
    if (m == -1) {  // bytes that are not a marker: skip to the next one
      while (m == -1) {
        if (d.eof()) return VT_EFORMAT;
        m = next_marker();
      }
      continue;
    }
    if (m == 0xc9 || m == 0xca || m == 0xcb) return VT_EUNSUPPORTED;  // arithmetic coding
    const int rc = d.tables(m);
    if (rc != 0) return rc;
    m = next_marker();
  }
  if (m == 0xc2) progressive_or_arith = true;
  {
    const int rc = d.frame_header();
    if (rc != 0) return rc;
  }
  if (progressive_or_arith) return VT_EUNSUPPORTED;
  *width = d.img_x;
  *height = d.img_y;
  // scans up to the end-of-image marker
  m = next_marker();
  while (m != 0xd9) {
    if (m == 0xda) {
      const int rc = d.scan(&pending);
      if (rc != 0) return rc;
      if (pending < 0) {  // padding behind the entropy-coded data: the next 0xff starts the next marker
        while (!d.eof()) {
          if (d.get8() == 0xff) {
            pending = d.get8();
            break;
          }
        }
      }
    } else if (m == 0xdc) {  // DNL: the line count of the frame header is replaced; only an equal one is accepted
      const int ld = d.get16(), nl = d.get16();
      if (ld != 4 || nl != d.img_y) return VT_EFORMAT;
    } else if (m == -1) {
      if (d.eof()) return VT_EFORMAT;
    } else {
      const int rc = d.tables(m);
      if (rc != 0) return rc;
    }
    m = next_marker();
  }
  const bool is_rgb = d.ncomp == 3 && (d.rgb == 3 || (d.app14 == 0 && !d.jfif));
  const int planes = is_rgb ? 3 : 1;
  for (int k = 0; k < planes; ++k)
    if (d.comp[k].h != d.h_max || d.comp[k].v != d.v_max) return VT_EUNSUPPORTED;  // a subsampled plane would need resampling
  out.resize((size_t)d.img_x * d.img_y);
  for (int y = 0; y < d.img_y; ++y)
    for (int x = 0; x < d.img_x; ++x) {
      if (!is_rgb) {
        out[(size_t)y * d.img_x + x] = d.comp[0].data[(size_t)y * d.comp[0].w2 + x];
      } else {
        const int r = d.comp[0].data[(size_t)y * d.comp[0].w2 + x], g = d.comp[1].data[(size_t)y * d.comp[1].w2 + x],
                  b = d.comp[2].data[(size_t)y * d.comp[2].w2 + x];
        out[(size_t)y * d.img_x + x] = (unsigned char)((r * 77 + g * 150 + 29 * b) >> 8);
      }
    }
  return 0;
}
