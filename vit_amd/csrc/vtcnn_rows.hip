// Token-row kernels of the VTCNN hot path on the gfx950 matrix cores: everything that works on
// the flat [N*L, D] token matrix (transformer FFN forward / backward, the projector's token side,
// the weight-gradient part of the attention backward).  D = 32, so every product is a 32-wide
// contraction: v_mfma_f32_16x16x4_f32 tiles of 16 token rows, one tile per wave, weights loaded
// once per wave as MFMA operands (they are 4 KB and live in L2).
//
// Layouts (lane = 16*g + j; see vtcnn_pix.hip for the instruction's register layout):
//   T tile: lane j = token row, registers t[tt][v] = feature 16*tt + 4*g + v   ("D order": what
//           an accumulator holds after D'[feature][row] products, and what two 16-byte loads at
//           columns 4g and 16+4g of a row give).  It is the B operand of a product that
//           contracts over features and keeps rows as columns, or, unchanged, the A operand of
//           the same product with rows as output ROWS (which yields the N layout).
//   N tile: lane j = feature (column j + 16*tile), registers = rows 4*g + v.  Operand of products
//           that contract over rows (weight gradients).
// Weight operands: for y = x.W (W row-major [in][ldw]) the A operand of output tile n0 is
// W[(16*tt + 4*g + v)*ldw + n0 + j]; for dx = dy.W^T it is W[(i0 + j)*ldw + 16*tt + 4*g + v]
// (one 16-byte load per tt).
#include "vtcnn_common.h"
#include "vtcnn_tail.h"

namespace vt {

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 f4zero() { return f4{0.f, 0.f, 0.f, 0.f}; }

__device__ __forceinline__ float gsum(float x) {  // over the four lane groups g
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}
__device__ __forceinline__ float jsum(float x) {  // over the 16 lanes j of a group
  x += __shfl_xor(x, 1);
  x += __shfl_xor(x, 2);
  x += __shfl_xor(x, 4);
  x += __shfl_xor(x, 8);
  return x;
}

struct Lane {
  int j, g;
  __device__ __forceinline__ Lane() : j(threadIdx.x & 15), g((threadIdx.x & 63) >> 4) {}
};

// T tile of a [rows][ld] matrix with 16*TT feature columns starting at col0; rows past nvalid are 0.
template <int TT>
struct TTile {
  f4 t[TT];
  __device__ __forceinline__ void load(const float *__restrict__ base, long row0, int nvalid, int ld, int col0, Lane ln) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
      t[tt] = ln.j < nvalid ? *reinterpret_cast<const f4 *>(base + (row0 + ln.j) * ld + col0 + 16 * tt + 4 * ln.g) : f4zero();
  }
  __device__ __forceinline__ void store(float *__restrict__ base, long row0, int nvalid, int ld, int col0, Lane ln) const {
    if (ln.j < nvalid) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) *reinterpret_cast<f4 *>(base + (row0 + ln.j) * ld + col0 + 16 * tt + 4 * ln.g) = t[tt];
    }
  }
};

// per-lane constants of a per-feature vector in D order: c[tt][v] = vec[16*tt + 4*g + v]
template <int TT>
__device__ __forceinline__ void load_fvec(const float *__restrict__ vec, f4 (&c)[TT], Lane ln) {
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) c[tt] = *reinterpret_cast<const f4 *>(vec + 16 * tt + 4 * ln.g);
}

// A operand of y = x.W for output columns n0..n0+15 (W row-major [16*TT][ldw]); zero past ncols.
template <int TT>
__device__ __forceinline__ void load_w(const float *__restrict__ W, int ldw, int n0, int ncols, f4 (&a)[TT], Lane ln) {
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int v = 0; v < 4; ++v) a[tt][v] = (n0 + ln.j < ncols) ? W[(16 * tt + 4 * ln.g + v) * ldw + n0 + ln.j] : 0.f;
}
// A operand of dx = dy.W^T for output features i0..i0+15 (W row-major [..][ldw], 16*TT columns used from col0)
template <int TT>
__device__ __forceinline__ void load_wt(const float *__restrict__ W, int ldw, int i0, int nrows, int col0, f4 (&a)[TT], Lane ln) {
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
    a[tt] = (i0 + ln.j < nrows) ? *reinterpret_cast<const f4 *>(W + (size_t)(i0 + ln.j) * ldw + col0 + 16 * tt + 4 * ln.g) : f4zero();
}
// acc[feature][row] += sum over the tile's features  a . x   (T layout out)
template <int TT>
__device__ __forceinline__ f4 mm_t(const f4 (&a)[TT], const f4 (&x)[TT], f4 acc) {
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc = mfma16(a[tt][v], x[tt][v], acc);
  return acc;
}
// acc[row][feature] += the same product with the roles swapped (N layout out)
template <int TT>
__device__ __forceinline__ f4 mm_n(const f4 (&x)[TT], const f4 (&w)[TT], f4 acc) {
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc = mfma16(x[tt][v], w[tt][v], acc);
  return acc;
}
// N tile from memory: n[v] = base[(row0 + 4g + v)*ld + col]; rows past nvalid are 0
__device__ __forceinline__ f4 load_n(const float *__restrict__ base, long row0, int nvalid, int ld, int col, Lane ln) {
  f4 n;
#pragma unroll
  for (int v = 0; v < 4; ++v) n[v] = (4 * ln.g + v < nvalid) ? base[(row0 + 4 * ln.g + v) * ld + col] : 0.f;
  return n;
}
// acc[i][e] += sum over the 16 rows  a_n[i][row] . b_n[row][e]
__device__ __forceinline__ f4 mm_rows(f4 a_n, f4 b_n, f4 acc) {
#pragma unroll
  for (int v = 0; v < 4; ++v) acc = mfma16(a_n[v], b_n[v], acc);
  return acc;
}

// Cross-wave sums of D tiles (rows i = 4g+v, columns j), one LDS phase for all of a kernel's tiles:
// every wave stages its copy of tile `slot`, ONE workgroup barrier follows, then tile_flush writes
// the workgroup totals ((w0 + w1) + (w2 + w3)) to dst[i*ld + j].  sbuf: slots x TILE_SLOT floats.
constexpr int TILE_SLOT = NW * 256;
__device__ __forceinline__ void tile_stage(f4 acc, float *sbuf, int slot) {
  const Lane ln;
  float *b = sbuf + slot * TILE_SLOT + (threadIdx.x >> 6) * 256;
#pragma unroll
  for (int v = 0; v < 4; ++v) b[(4 * ln.g + v) * 16 + ln.j] = acc[v];
}
__device__ __forceinline__ void tile_flush(const float *sbuf, int slot, float *__restrict__ dst, int ld) {
  const int tid = threadIdx.x;
  const float *b = sbuf + slot * TILE_SLOT;
  dst[(tid >> 4) * ld + (tid & 15)] = (b[tid] + b[256 + tid]) + (b[512 + tid] + b[768 + tid]);
}

// ---------------------------------------------------------------------------------------------
// TransformerLayer::forward (reference layers/transformer_layer.hpp:43-120) over the flat
// [N*L, D] token matrix (Tensor::permute does not move data, tensor_cpu.hpp:719-747).
// The attention branch is identically zero in the reference (Tensor::concat returns zeros,
// tensor_cpu.hpp:251; no bias on the output projection), so residual = tokens.
//   a = BN1(tok);  hdn = a.W1 + b1;  s2 = a + relu(hdn).W2 + b2   (+ partial stats of s2)
// RB = 64 rows per workgroup = one 16-row tile per wave; 32 MFMAs per tile.
// ---------------------------------------------------------------------------------------------
template <bool TAIL>  // TAIL: the last workgroup closes norm2's statistics (vtcnn_tail.h)
__global__ __launch_bounds__(TPB) void k_tr_ffn_fwd(const float *__restrict__ tok, const float *__restrict__ st,
                                                    const float *__restrict__ g, const float *__restrict__ b,
                                                    const float *__restrict__ W1, const float *__restrict__ b1,
                                                    const float *__restrict__ W2, const float *__restrict__ b2,
                                                    float *__restrict__ a_out, float *__restrict__ hdn_out,
                                                    float *__restrict__ s2_out, float *__restrict__ part_n2,
                                                    long rows, long group_rows, const BnFwdTail tail) {
  __shared__ float sred[NW][D];
  const Lane ln;
  const int wave = threadIdx.x >> 6;
  const long row0 = ((long)blockIdx.x * NW + wave) * 16;
  const int nv = (int)max(0L, min(16L, rows - row0));
  // statistics group of this workgroup's rows (group_rows is a multiple of RB; 0 = one group)
  if (group_rows > 0) st += ((long)blockIdx.x * RB / group_rows) * 2 * D;
  f4 mean[2], inv[2], gam[2], bet[2], bb1[2], bb2[2];
  load_fvec<2>(st, mean, ln);
  load_fvec<2>(st + D, inv, ln);
  load_fvec<2>(g, gam, ln);
  load_fvec<2>(b, bet, ln);
  load_fvec<2>(b1, bb1, ln);
  load_fvec<2>(b2, bb2, ln);
  TTile<2> x;
  x.load(tok, row0, nv, D, 0, ln);
  TTile<2> a;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 4; ++v) a.t[tt][v] = ((x.t[tt][v] - mean[tt][v]) * inv[tt][v]) * gam[tt][v] + bet[tt][v];
  a.store(a_out, row0, nv, D, 0, ln);
  TTile<2> h, r, s2;
#pragma unroll
  for (int to = 0; to < 2; ++to) {
    f4 w[2];
    load_w<2>(W1, D, 16 * to, D, w, ln);
    h.t[to] = mm_t<2>(w, a.t, bb1[to]);
#pragma unroll
    for (int v = 0; v < 4; ++v) r.t[to][v] = h.t[to][v] > 0.f ? h.t[to][v] : 0.f;
  }
  h.store(hdn_out, row0, nv, D, 0, ln);
#pragma unroll
  for (int to = 0; to < 2; ++to) {
    f4 w[2];
    load_w<2>(W2, D, 16 * to, D, w, ln);
    const f4 o = mm_t<2>(w, r.t, bb2[to]);
#pragma unroll
    for (int v = 0; v < 4; ++v) s2.t[to][v] = a.t[to][v] + o[v];
  }
  s2.store(s2_out, row0, nv, D, 0, ln);
  if (part_n2 == nullptr) return;
  // per-workgroup (sum, centred M2) of s2 per feature over the workgroup's valid rows
  // (a group boundary never falls inside a workgroup, so "rows left" is also right for the last
  // workgroup of a short last group: the short group is the last one)
  const long left = rows - (long)blockIdx.x * RB;
  const float cnt = (float)(left < RB ? left : RB);
  const bool ok = ln.j < nv;
  f4 sm[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 4; ++v) sm[tt][v] = jsum(ok ? s2.t[tt][v] : 0.f);
  if (ln.j == 0) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int v = 0; v < 4; ++v) sred[wave][16 * tt + 4 * ln.g + v] = sm[tt][v];
  }
  __syncthreads();
  f4 mu[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int f = 16 * tt + 4 * ln.g + v;
      sm[tt][v] = (sred[0][f] + sred[1][f]) + (sred[2][f] + sred[3][f]);
      mu[tt][v] = sm[tt][v] / cnt;
    }
  __syncthreads();
  f4 q[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float dlt = s2.t[tt][v] - mu[tt][v];
      q[tt][v] = jsum(ok ? dlt * dlt : 0.f);
    }
  if (ln.j == 0) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int v = 0; v < 4; ++v) sred[wave][16 * tt + 4 * ln.g + v] = q[tt][v];
  }
  __syncthreads();
  if (threadIdx.x < D) {
    const int f = threadIdx.x;
    // every lane holds the workgroup sums of its own features; lanes (wave 0, j == 0) write them
    const float rq = (sred[0][f] + sred[1][f]) + (sred[2][f] + sred[3][f]);
    if constexpr (!TAIL) part_n2[(size_t)blockIdx.x * 2 * D + D + f] = rq;
    else tail_store(part_n2 + (size_t)blockIdx.x * 2 * D + D + f, rq);
  }
  if (wave == 0 && ln.j == 0) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float *dst = part_n2 + (size_t)blockIdx.x * 2 * D + 16 * tt + 4 * ln.g + v;
        if constexpr (!TAIL) *dst = sm[tt][v];
        else tail_store(dst, sm[tt][v]);
      }
  }
  if constexpr (TAIL) bn_fwd_tail(tail, part_n2, blockIdx.x, gridDim.x, D);
}

// ---------------------------------------------------------------------------------------------
// Transformer output o = BN2(s2) and the token side of ProjectorLayer::forward
// (reference layers/projector_layer.hpp:72-88): k = o.Wk, v = o.Wv + bv, plus the folded score
// matrix Mq[c][l] = sum_d Wq[c][d] k[l][d], so that the per-pixel score x.Wq.k^T becomes x.Mq
// (8 MACs per token instead of 8*32 + 32).  The L = 16 tokens of a sample are exactly one tile:
// one wave per image.  pk record per image: Mq[C1][L] | v[L][C2] | k[L][D].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_proj_tokens_fwd(const float *__restrict__ s2, const float *__restrict__ st,
                                                         const float *__restrict__ g, const float *__restrict__ b,
                                                         const float *__restrict__ Wq, const float *__restrict__ Wk,
                                                         const float *__restrict__ Wv, const float *__restrict__ bv,
                                                         float *__restrict__ o_out, float *__restrict__ pk, int N,
                                                         int group) {
  static_assert(L == 16, "one 16-row tile per image");
  const Lane ln;
  const int n = blockIdx.x * NW + (threadIdx.x >> 6);
  if (n >= N) return;
  if (group > 0) st += (size_t)(n / group) * 2 * D;  // statistics group of this image
  const long row0 = (long)n * L;
  f4 mean[2], inv[2], gam[2], bet[2];
  load_fvec<2>(st, mean, ln);
  load_fvec<2>(st + D, inv, ln);
  load_fvec<2>(g, gam, ln);
  load_fvec<2>(b, bet, ln);
  TTile<2> x, o;
  x.load(s2, row0, 16, D, 0, ln);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 4; ++v) o.t[tt][v] = ((x.t[tt][v] - mean[tt][v]) * inv[tt][v]) * gam[tt][v] + bet[tt][v];
  o.store(o_out, row0, 16, D, 0, ln);
  float *rec = pk + (size_t)n * PK_STRIDE;
  TTile<2> k;
#pragma unroll
  for (int to = 0; to < 2; ++to) {
    f4 w[2];
    load_w<2>(Wk, D, 16 * to, D, w, ln);
    k.t[to] = mm_t<2>(w, o.t, f4zero());
  }
  k.store(rec + C1 * L + L * C2, 0, 16, D, 0, ln);
  {
    f4 w[2], bias;
    load_w<2>(Wv, C2, 0, C2, w, ln);
    bias = *reinterpret_cast<const f4 *>(bv + 4 * ln.g);
    TTile<1> v;
    v.t[0] = mm_t<2>(w, o.t, bias);
    v.store(rec + C1 * L, 0, 16, C2, 0, ln);
  }
  {
    // Mq[c][l]: A = Wq rows c (8 of the 16 output rows are used), B = k in T layout
    f4 w[2];
    load_wt<2>(Wq, D, 0, C1, 0, w, ln);
    const f4 mq = mm_t<2>(w, k.t, f4zero());
    if (ln.g < 2) {
#pragma unroll
      for (int v = 0; v < 4; ++v) rec[(4 * ln.g + v) * L + ln.j] = mq[v];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ProjectorLayer::backward, token side (reference projector_layer.hpp:186-202): Dense backward of
// linear3 (values) and linear2 (keys) over the N*L token rows.
//   gWv = o^T dv, gbv = sum dv, gWk = o^T dk, token delta = dv.Wv^T + dk.Wk^T
// Per-workgroup partials: gWk[D*D] | gWv[D*C2] | gbv[C2]  (PT_STRIDE floats), summed afterwards.
// ---------------------------------------------------------------------------------------------
constexpr int PT_STRIDE = D * D + D * C2 + C2;  // 1552
constexpr int PT_SLOTS = 6 + 1;                  // staged tiles + one slot holding the bias sums
// sbuf: PT_SLOTS * TILE_SLOT floats of LDS.  `out` returns this wave's tile of the token delta (T layout).
__device__ __forceinline__ void projtok_bwd_body(const float *__restrict__ o, const float *__restrict__ dv,
                                                 const float *__restrict__ dk, const float *__restrict__ Wv,
                                                 const float *__restrict__ Wk, float *__restrict__ pdt,
                                                 float *__restrict__ scratch, long rows, float *sbuf, TTile<2> &out) {
  const Lane ln;
  const int wave = threadIdx.x >> 6;
  const long row0 = ((long)blockIdx.x * NW + wave) * 16;
  const int nv = (int)max(0L, min(16L, rows - row0));
  // token delta, T layout
  TTile<1> dvt;
  TTile<2> dkt;
  dvt.load(dv, row0, nv, C2, 0, ln);
  dkt.load(dk, row0, nv, D, 0, ln);
#pragma unroll
  for (int to = 0; to < 2; ++to) {
    f4 wv[1], wk[2];
    load_wt<1>(Wv, C2, 16 * to, D, 0, wv, ln);
    load_wt<2>(Wk, D, 16 * to, D, 0, wk, ln);
    out.t[to] = mm_t<2>(wk, dkt.t, mm_t<1>(wv, dvt.t, f4zero()));
  }
  out.store(pdt, row0, nv, D, 0, ln);
  // weight gradients: contraction over the tile's rows, N-layout operands from memory
  f4 on[2], dkn[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    on[t] = load_n(o, row0, nv, D, ln.j + 16 * t, ln);
    dkn[t] = load_n(dk, row0, nv, D, ln.j + 16 * t, ln);
  }
  const f4 dvn = load_n(dv, row0, nv, C2, ln.j, ln);
  float *row = scratch + (size_t)blockIdx.x * PT_STRIDE;
  // slots 0..3: gWk tiles, 4..5: gWv tiles, 6: the bias sums of the four waves
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
    for (int te = 0; te < 2; ++te) tile_stage(mm_rows(on[ti], dkn[te], f4zero()), sbuf, 2 * ti + te);
    tile_stage(mm_rows(on[ti], dvn, f4zero()), sbuf, 4 + ti);
  }
  const float sbv = gsum((dvn[0] + dvn[1]) + (dvn[2] + dvn[3]));  // lane j: sum over the tile's rows of dv[.][j]
  float *bs = sbuf + 6 * TILE_SLOT;
  if (ln.g == 0) bs[wave * 16 + ln.j] = sbv;
  __syncthreads();
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
    for (int te = 0; te < 2; ++te) tile_flush(sbuf, 2 * ti + te, row + (16 * ti) * D + 16 * te, D);
    tile_flush(sbuf, 4 + ti, row + D * D + (16 * ti) * C2, C2);
  }
  if (threadIdx.x < C2) row[D * D + D * C2 + threadIdx.x] = (bs[threadIdx.x] + bs[16 + threadIdx.x]) + (bs[32 + threadIdx.x] + bs[48 + threadIdx.x]);
}
__global__ __launch_bounds__(TPB) void k_projtok_bwd(const float *__restrict__ o, const float *__restrict__ dv,
                                                     const float *__restrict__ dk, const float *__restrict__ Wv,
                                                     const float *__restrict__ Wk, float *__restrict__ pdt,
                                                     float *__restrict__ scratch, long rows) {
  __shared__ float sbuf[PT_SLOTS * TILE_SLOT];
  TTile<2> out;
  projtok_bwd_body(o, dv, dk, Wv, Wk, pdt, scratch, rows, sbuf, out);
}

// ---------------------------------------------------------------------------------------------
// TransformerLayer::backward, step A (reference layers/transformer_layer.hpp:142-161):
//   d0 = last_output - targets  with targets = projector token delta (visual_transformer.hpp:156-157)
//   partial sums of the second BatchNorm1D backward.
// Also the tokenizer's linear2 input delta dT0 = token_delta . W2tok^T (tokenizer.hpp:73-75), which
// reads the same rows.
// ---------------------------------------------------------------------------------------------
// pt: this wave's tile of the projector token delta (T layout).  sred: NW x 2D floats of LDS.
template <bool TAIL>
__device__ __forceinline__ void tr_bwd_a_body(const float *__restrict__ o, const TTile<2> &pt,
                                              const float *__restrict__ s2, const float *__restrict__ st2,
                                              const float *__restrict__ W2tok, float *__restrict__ part,
                                              float *__restrict__ dT0, long rows, float (*sred)[2 * D],
                                              const BnBwdTail &tail) {
  const Lane ln;
  const int wave = threadIdx.x >> 6;
  const long row0 = ((long)blockIdx.x * NW + wave) * 16;
  const int nv = (int)max(0L, min(16L, rows - row0));
  f4 mean[2], inv[2];
  load_fvec<2>(st2, mean, ln);
  load_fvec<2>(st2 + D, inv, ln);
  TTile<2> ot, xt;
  ot.load(o, row0, nv, D, 0, ln);
  xt.load(s2, row0, nv, D, 0, ln);
  const bool ok = ln.j < nv;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float d0 = ok ? ot.t[tt][v] - pt.t[tt][v] : 0.f;
      const float xh = (xt.t[tt][v] - mean[tt][v]) * inv[tt][v];
      const float s1 = jsum(d0), s2v = jsum(ok ? d0 * xh : 0.f);
      if (ln.j == 0) {
        sred[wave][16 * tt + 4 * ln.g + v] = s1;
        sred[wave][D + 16 * tt + 4 * ln.g + v] = s2v;
      }
    }
  {
    // dT0[row][c] = sum_d pdt[row][d] W2tok[c][d]: 8 output features
    f4 w[2];
    load_wt<2>(W2tok, D, 0, C1, 0, w, ln);
    const f4 r = mm_t<2>(w, pt.t, f4zero());
    if (ok && ln.g < 2) *reinterpret_cast<f4 *>(dT0 + (row0 + ln.j) * C1 + 4 * ln.g) = r;
  }
  __syncthreads();
  if (threadIdx.x < 2 * D) {
    const int f = threadIdx.x;
    const float r = (sred[0][f] + sred[1][f]) + (sred[2][f] + sred[3][f]);
    if constexpr (!TAIL) part[(size_t)blockIdx.x * 2 * D + f] = r;
    else tail_store(part + (size_t)blockIdx.x * 2 * D + f, r);
  }
  if constexpr (TAIL) bn_bwd_tail(tail, part, blockIdx.x, gridDim.x, D);
}
__global__ __launch_bounds__(TPB) void k_tr_bwd_a(const float *__restrict__ o, const float *__restrict__ pdt,
                                                  const float *__restrict__ s2, const float *__restrict__ st2,
                                                  const float *__restrict__ W2tok, float *__restrict__ part,
                                                  float *__restrict__ dT0, long rows) {
  __shared__ float sred[NW][2 * D];
  const Lane ln;
  const long row0 = ((long)blockIdx.x * NW + (threadIdx.x >> 6)) * 16;
  TTile<2> pt;
  pt.load(pdt, row0, (int)max(0L, min(16L, rows - row0)), D, 0, ln);
  tr_bwd_a_body<false>(o, pt, s2, st2, W2tok, part, dT0, rows, sred, BnBwdTail{});
}
// Both of the above in one launch: step A only needs the token delta of its own rows, which the
// projector's token-side backward has just formed in registers, so no device-wide ordering sits
// between them.  Same arithmetic in the same order: bit-identical to the two separate launches.
template <bool TAIL>  // TAIL: the last workgroup closes norm2's backward reduction
__global__ __launch_bounds__(TPB) void k_projtok_bwd_tr_a(const float *__restrict__ o, const float *__restrict__ dv,
                                                          const float *__restrict__ dk, const float *__restrict__ Wv,
                                                          const float *__restrict__ Wk, float *__restrict__ pdt,
                                                          float *__restrict__ scratch, const float *__restrict__ s2,
                                                          const float *__restrict__ st2, const float *__restrict__ W2tok,
                                                          float *__restrict__ part, float *__restrict__ dT0, long rows,
                                                          long group_rows, const BnBwdTail tail) {
  __shared__ float sbuf[PT_SLOTS * TILE_SLOT];
  __shared__ float sred[NW][2 * D];
  // statistics group of this workgroup's rows (group_rows is a multiple of RB; 0 = one group)
  if (group_rows > 0) st2 += ((long)blockIdx.x * RB / group_rows) * 2 * D;
  TTile<2> pt;
  projtok_bwd_body(o, dv, dk, Wv, Wk, pdt, scratch, rows, sbuf, pt);
  tr_bwd_a_body<TAIL>(o, pt, s2, st2, W2tok, part, dT0, rows, sred, tail);
}

// ---------------------------------------------------------------------------------------------
// TransformerLayer::backward, step B (reference transformer_layer.hpp:161-189):
//   ds2 = BN2.backward(d0);  linear2 backward (its cached input is relu(hdn));
//   linear1 backward WITHOUT the ReLU derivative (:174-181);  da = dh.W1^T + ds2;
//   partial sums of the first BatchNorm1D backward.
// Per-workgroup partials: gW1[D*D] | gb1[D] | gW2[D*D] | gb2[D]  (FF_STRIDE, arena order).
// dh is produced twice, in T layout (feeds da) and in N layout (feeds gW1), by the same MFMAs with
// the operand roles swapped; nothing is transposed through memory.
// ---------------------------------------------------------------------------------------------
constexpr int FF_STRIDE = 2 * (D * D + D);  // 2112
template <bool TAIL>  // TAIL: the last workgroup closes norm1's backward reduction
__global__ __launch_bounds__(TPB) void k_tr_bwd_b(const float *__restrict__ o, const float *__restrict__ pdt,
                                                  const float *__restrict__ coef2, const float *__restrict__ hdn,
                                                  const float *__restrict__ a, const float *__restrict__ tok,
                                                  const float *__restrict__ st1, const float *__restrict__ W1,
                                                  const float *__restrict__ W2, float *__restrict__ ds2_out,
                                                  float *__restrict__ da_out, float *__restrict__ scratch,
                                                  float *__restrict__ part, long rows, long group_rows,
                                                  const BnBwdTail tail) {
  __shared__ float sbuf[9 * TILE_SLOT];
  __shared__ float sred[NW][2 * D];
  if (group_rows > 0) {  // statistics group of this workgroup's rows (group_rows is a multiple of RB; 0 = one group)
    const long grp = (long)blockIdx.x * RB / group_rows;
    coef2 += grp * 2 * D;
    st1 += grp * 2 * D;
  }
  const Lane ln;
  const int wave = threadIdx.x >> 6;
  const long row0 = ((long)blockIdx.x * NW + wave) * 16;
  const int nv = (int)max(0L, min(16L, rows - row0));
  const bool ok = ln.j < nv;
  f4 c0[2], c1[2];
  load_fvec<2>(coef2, c0, ln);
  load_fvec<2>(coef2 + D, c1, ln);
  TTile<2> ot, pt, ds2;
  ot.load(o, row0, nv, D, 0, ln);
  pt.load(pdt, row0, nv, D, 0, ln);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 4; ++v) ds2.t[tt][v] = ok ? (ot.t[tt][v] - pt.t[tt][v]) * c0[tt][v] + c1[tt][v] : 0.f;
  ds2.store(ds2_out, row0, nv, D, 0, ln);
  // dh = ds2 . W2^T in both layouts; da = dh . W1^T + ds2
  TTile<2> dh, da;
  f4 dhn[2];
#pragma unroll
  for (int to = 0; to < 2; ++to) {
    f4 w[2];
    load_wt<2>(W2, D, 16 * to, D, 0, w, ln);
    dh.t[to] = mm_t<2>(w, ds2.t, f4zero());
    dhn[to] = mm_n<2>(ds2.t, w, f4zero());
  }
#pragma unroll
  for (int to = 0; to < 2; ++to) {
    f4 w[2];
    load_wt<2>(W1, D, 16 * to, D, 0, w, ln);
    da.t[to] = mm_t<2>(w, dh.t, ds2.t[to]);
  }
  da.store(da_out, row0, nv, D, 0, ln);
  // partial sums of the first BatchNorm1D backward: sum da, sum da * xhat(tok)
  {
    f4 mean[2], inv[2];
    load_fvec<2>(st1, mean, ln);
    load_fvec<2>(st1 + D, inv, ln);
    TTile<2> xt;
    xt.load(tok, row0, nv, D, 0, ln);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float dv = ok ? da.t[tt][v] : 0.f;
        const float xh = (xt.t[tt][v] - mean[tt][v]) * inv[tt][v];
        const float s1 = jsum(dv), s2v = jsum(ok ? dv * xh : 0.f);
        if (ln.j == 0) {
          sred[wave][16 * tt + 4 * ln.g + v] = s1;
          sred[wave][D + 16 * tt + 4 * ln.g + v] = s2v;
        }
      }
  }
  // weight gradients over the tile's rows (N-layout operands: lane = feature, registers = rows)
  f4 an[2], rn[2], dsn[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = ln.j + 16 * t;
    an[t] = load_n(a, row0, nv, D, col, ln);
    const f4 hn = load_n(hdn, row0, nv, D, col, ln);
    const f4 on = load_n(o, row0, nv, D, col, ln), pn = load_n(pdt, row0, nv, D, col, ln);
    const float k0 = coef2[col], k1 = coef2[D + col];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      rn[t][v] = hn[v] > 0.f ? hn[v] : 0.f;
      dsn[t][v] = (4 * ln.g + v < nv) ? (on[v] - pn[v]) * k0 + k1 : 0.f;
    }
  }
  float *row = scratch + (size_t)blockIdx.x * FF_STRIDE;
  // slots 0..3: gW1 = a^T dh, 4..7: gW2 = relu(hdn)^T ds2, 8: the bias sums of the four waves
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int te = 0; te < 2; ++te) {
      tile_stage(mm_rows(an[ti], dhn[te], f4zero()), sbuf, 2 * ti + te);
      tile_stage(mm_rows(rn[ti], dsn[te], f4zero()), sbuf, 4 + 2 * ti + te);
    }
  float *bs = sbuf + 8 * TILE_SLOT;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float sb1 = gsum((dhn[t][0] + dhn[t][1]) + (dhn[t][2] + dhn[t][3]));  // gb1[j+16t] over the tile's rows
    const float sb2 = gsum((dsn[t][0] + dsn[t][1]) + (dsn[t][2] + dsn[t][3]));  // gb2
    if (ln.g == 0) {
      bs[wave * 64 + 16 * t + ln.j] = sb1;
      bs[wave * 64 + 32 + 16 * t + ln.j] = sb2;
    }
  }
  __syncthreads();
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int te = 0; te < 2; ++te) {
      tile_flush(sbuf, 2 * ti + te, row + (16 * ti) * D + 16 * te, D);
      tile_flush(sbuf, 4 + 2 * ti + te, row + D * D + D + (16 * ti) * D + 16 * te, D);
    }
  if (threadIdx.x < 2 * D) {
    const int f = threadIdx.x;
    const float sm = (bs[f] + bs[64 + f]) + (bs[128 + f] + bs[192 + f]);
    if (f < D) row[D * D + f] = sm;
    else row[2 * D * D + D + (f - D)] = sm;
    const float r = (sred[0][f] + sred[1][f]) + (sred[2][f] + sred[3][f]);
    if constexpr (!TAIL) part[(size_t)blockIdx.x * 2 * D + f] = r;
    else tail_store(part + (size_t)blockIdx.x * 2 * D + f, r);
  }
  if constexpr (TAIL) bn_bwd_tail(tail, part, blockIdx.x, gridDim.x, D);
}

// ---------------------------------------------------------------------------------------------
// TransformerLayer::backward, step C (reference transformer_layer.hpp:189-200 and the first part
// of MultiHeadAttention::backward, multi_head_attention.hpp:131-152):
//   dr = BN1.backward(da);  dc = dr . Wo^T  (delta of the zero "concat");
//   per head q = X.Wq, k = X.Wk, v = X.Wv recomputed from the layer input X = token rows.
// qkv layout: three [rows][D] matrices (q | k | v), head h in columns h*HD..h*HD+HD-1.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_tr_bwd_c(const float *__restrict__ da, const float *__restrict__ coef1,
                                                  const float *__restrict__ Wo, const float *__restrict__ tok,
                                                  const float *__restrict__ Wqkv, float *__restrict__ dr_out,
                                                  float *__restrict__ dc, float *__restrict__ qkv, long rows,
                                                  long group_rows) {
  static_assert(HD == 16, "one output tile per head");
  if (group_rows > 0) coef1 += ((long)blockIdx.x * RB / group_rows) * 2 * D;  // this workgroup's statistics group
  const Lane ln;
  const int wave = threadIdx.x >> 6;
  const long row0 = ((long)blockIdx.x * NW + wave) * 16;
  const int nv = (int)max(0L, min(16L, rows - row0));
  if (nv == 0) return;
  f4 c0[2], c1[2];
  load_fvec<2>(coef1, c0, ln);
  load_fvec<2>(coef1 + D, c1, ln);
  TTile<2> dat, dr, x;
  dat.load(da, row0, nv, D, 0, ln);
  x.load(tok, row0, nv, D, 0, ln);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int v = 0; v < 4; ++v) dr.t[tt][v] = dat.t[tt][v] * c0[tt][v] + c1[tt][v];
  dr.store(dr_out, row0, nv, D, 0, ln);
  TTile<2> out;
#pragma unroll
  for (int to = 0; to < 2; ++to) {
    f4 w[2];
    load_wt<2>(Wo, D, 16 * to, D, 0, w, ln);
    out.t[to] = mm_t<2>(w, dr.t, f4zero());
  }
  out.store(dc, row0, nv, D, 0, ln);
  // arena order of the six per-head matrices: q0,k0,v0,q1,k1,v1, each [D][HD]
#pragma unroll
  for (int m = 0; m < 3; ++m) {
#pragma unroll
    for (int h = 0; h < HEADS; ++h) {
      f4 w[2];
      load_w<2>(Wqkv + (size_t)(h * 3 + m) * D * HD, HD, 0, HD, w, ln);
      out.t[h] = mm_t<2>(w, x.t, f4zero());
    }
    out.store(qkv + (size_t)m * rows * D, row0, nv, D, 0, ln);
  }
}

}  // namespace vt

// ---------------------------------------------------------------------------------------------
// extern "C" launchers
// ---------------------------------------------------------------------------------------------
using namespace vt;

static int ffn_fwd_launch(const float *tokens, const float *n1_stat, const float *n1_gamma, const float *n1_beta,
                          const float *w1, const float *b1, const float *w2, const float *b2, float *a, float *hdn, float *s2,
                          float *part_n2, long rows, long group_rows, const BnFwdTail &tail, hipStream_t s) {
  if (rows < 1 || group_rows < 0 || group_rows % RB != 0) return VT_EINVAL;
  const int nb = (int)((rows + RB - 1) / RB);
  if (group_rows == 0 && part_n2 && tail.ticket)
    hipLaunchKernelGGL(k_tr_ffn_fwd<true>, dim3(nb), dim3(TPB), 0, s, tokens, n1_stat, n1_gamma, n1_beta, w1, b1, w2, b2, a, hdn,
                       s2, part_n2, rows, group_rows, tail);
  else
    hipLaunchKernelGGL(k_tr_ffn_fwd<false>, dim3(nb), dim3(TPB), 0, s, tokens, n1_stat, n1_gamma, n1_beta, w1, b1, w2, b2, a, hdn,
                       s2, part_n2, rows, group_rows, BnFwdTail{});
  VT_CHECK_LAUNCH();
  return 0;
}
int vt::transformer_ffn_fwd_tail(const float *tokens, const float *n1_stat, const float *n1_gamma, const float *n1_beta,
                                 const float *w1, const float *b1, const float *w2, const float *b2, float *a, float *hdn,
                                 float *s2, float *part_n2, long rows, const BnFwdTail &tail, hipStream_t s) {
  return ffn_fwd_launch(tokens, n1_stat, n1_gamma, n1_beta, w1, b1, w2, b2, a, hdn, s2, part_n2, rows, 0, tail, s);
}
extern "C" int vt_transformer_ffn_fwd_grouped(const float *tokens, const float *n1_stat, const float *n1_gamma,
                                              const float *n1_beta, const float *w1, const float *b1, const float *w2,
                                              const float *b2, float *a, float *hdn, float *s2, float *part_n2,
                                              long rows, long group_rows, vt_stream s) {
  return ffn_fwd_launch(tokens, n1_stat, n1_gamma, n1_beta, w1, b1, w2, b2, a, hdn, s2, part_n2, rows, group_rows, BnFwdTail{},
                        (hipStream_t)s);
}
extern "C" int vt_transformer_ffn_fwd(const float *tokens, const float *n1_stat, const float *n1_gamma,
                                      const float *n1_beta, const float *w1, const float *b1, const float *w2,
                                      const float *b2, float *a, float *hdn, float *s2, float *part_n2, long rows,
                                      vt_stream s) {
  return vt_transformer_ffn_fwd_grouped(tokens, n1_stat, n1_gamma, n1_beta, w1, b1, w2, b2, a, hdn, s2, part_n2, rows, 0, s);
}

extern "C" int vt_projector_tokens_fwd_grouped(const float *s2, const float *n2_stat, const float *n2_gamma,
                                               const float *n2_beta, const float *wq, const float *wk, const float *wv,
                                               const float *bv, float *o, float *pk, int N, int group, vt_stream s) {
  if (N < 1 || group < 0) return VT_EINVAL;
  hipLaunchKernelGGL(k_proj_tokens_fwd, dim3((N + NW - 1) / NW), dim3(TPB), 0, (hipStream_t)s, s2, n2_stat, n2_gamma,
                     n2_beta, wq, wk, wv, bv, o, pk, N, group);
  VT_CHECK_LAUNCH();
  return 0;
}
extern "C" int vt_projector_tokens_fwd(const float *s2, const float *n2_stat, const float *n2_gamma,
                                       const float *n2_beta, const float *wq, const float *wk, const float *wv,
                                       const float *bv, float *o, float *pk, int N, vt_stream s) {
  return vt_projector_tokens_fwd_grouped(s2, n2_stat, n2_gamma, n2_beta, wq, wk, wv, bv, o, pk, N, 0, s);
}

// Token side of the projector backward; scratch >= ceil(rows/64) * 1552 floats.
extern "C" int vt_projector_tokens_bwd(const float *o, const float *dv, const float *dk, const float *wv,
                                       const float *wk, float *pdt, float *g_wk_wv_bv, float *scratch, long rows,
                                       vt_stream s) {
  if (rows < 1) return VT_EINVAL;
  const int nb = (int)((rows + RB - 1) / RB);
  hipLaunchKernelGGL(k_projtok_bwd, dim3(nb), dim3(TPB), 0, (hipStream_t)s, o, dv, dk, wv, wk, pdt, scratch, rows);
  VT_CHECK_LAUNCH();
  return reduce_rows(scratch, nb, PT_STRIDE, PT_STRIDE, g_wk_wv_bv, (hipStream_t)s);
}

extern "C" int vt_transformer_bwd_a(const float *o, const float *pdt, const float *s2, const float *n2_stat,
                                    const float *tok_w2, float *part, float *dT0, long rows, vt_stream s) {
  if (rows < 1) return VT_EINVAL;
  const int nb = (int)((rows + RB - 1) / RB);
  hipLaunchKernelGGL(k_tr_bwd_a, dim3(nb), dim3(TPB), 0, (hipStream_t)s, o, pdt, s2, n2_stat, tok_w2, part, dT0, rows);
  VT_CHECK_LAUNCH();
  return 0;
}

// vt_projector_tokens_bwd followed by vt_transformer_bwd_a as one launch (same results, bit for bit).
int vt::projector_tokens_bwd_transformer_a_tail(const float *o, const float *dv, const float *dk, const float *wv,
                                                const float *wk, float *pdt, float *g_wk_wv_bv, float *scratch,
                                                const float *s2, const float *n2_stat, const float *tok_w2, float *part,
                                                float *dT0, long rows, const BnBwdTail &tail, hipStream_t s,
                                                long group_rows) {
  if (rows < 1 || group_rows < 0 || group_rows % RB != 0) return VT_EINVAL;
  const int nb = (int)((rows + RB - 1) / RB);
  if (tail.ticket && group_rows == 0)
    VT_LAUNCH_SIGNALLING(k_projtok_bwd_tr_a<true>, dim3(nb), dim3(TPB), 0, s, o, dv, dk, wv, wk, pdt, scratch, s2, n2_stat, tok_w2,
                         part, dT0, rows, 0L, tail);
  else
    VT_LAUNCH_SIGNALLING(k_projtok_bwd_tr_a<false>, dim3(nb), dim3(TPB), 0, s, o, dv, dk, wv, wk, pdt, scratch, s2, n2_stat,
                         tok_w2, part, dT0, rows, group_rows, BnBwdTail{});
  VT_CHECK_LAUNCH();
  return reduce_rows(scratch, nb, PT_STRIDE, PT_STRIDE, g_wk_wv_bv, s);
}
extern "C" int vt_projector_tokens_bwd_transformer_a(const float *o, const float *dv, const float *dk, const float *wv,
                                                     const float *wk, float *pdt, float *g_wk_wv_bv, float *scratch,
                                                     const float *s2, const float *n2_stat, const float *tok_w2,
                                                     float *part, float *dT0, long rows, vt_stream s) {
  return projector_tokens_bwd_transformer_a_tail(o, dv, dk, wv, wk, pdt, g_wk_wv_bv, scratch, s2, n2_stat, tok_w2, part, dT0,
                                                 rows, BnBwdTail{}, (hipStream_t)s, 0);
}

// g_ffn: gradient block [gW1 | gb1 | gW2 | gb2] (2112 floats, arena order); scratch >= nb*2112.
int vt::transformer_bwd_b_tail(const float *o, const float *pdt, const float *n2_coef, const float *hdn, const float *a,
                               const float *tok, const float *n1_stat, const float *w1, const float *w2, float *ds2, float *da,
                               float *g_ffn, float *scratch, float *part, long rows, const BnBwdTail &tail, hipStream_t s,
                               long group_rows) {
  if (rows < 1 || group_rows < 0 || group_rows % RB != 0) return VT_EINVAL;
  const int nb = (int)((rows + RB - 1) / RB);
  if (tail.ticket && group_rows == 0)
    hipLaunchKernelGGL(k_tr_bwd_b<true>, dim3(nb), dim3(TPB), 0, s, o, pdt, n2_coef, hdn, a, tok, n1_stat, w1, w2, ds2, da, scratch,
                       part, rows, 0L, tail);
  else
    hipLaunchKernelGGL(k_tr_bwd_b<false>, dim3(nb), dim3(TPB), 0, s, o, pdt, n2_coef, hdn, a, tok, n1_stat, w1, w2, ds2, da, scratch,
                       part, rows, group_rows, BnBwdTail{});
  VT_CHECK_LAUNCH();
  return reduce_rows(scratch, nb, FF_STRIDE, FF_STRIDE, g_ffn, s);
}
extern "C" int vt_transformer_bwd_b(const float *o, const float *pdt, const float *n2_coef, const float *hdn,
                                    const float *a, const float *tok, const float *n1_stat, const float *w1,
                                    const float *w2, float *ds2, float *da, float *g_ffn, float *scratch, float *part,
                                    long rows, vt_stream s) {
  return transformer_bwd_b_tail(o, pdt, n2_coef, hdn, a, tok, n1_stat, w1, w2, ds2, da, g_ffn, scratch, part, rows, BnBwdTail{},
                                (hipStream_t)s, 0);
}

// group_rows: token rows per statistics group (n1_coef then holds one record per group; a multiple of 64), 0 = one group
int vt::transformer_bwd_c_grouped(const float *da, const float *n1_coef, const float *wo, const float *tok, const float *wqkv,
                                  float *dr, float *dc, float *qkv, long rows, long group_rows, hipStream_t s) {
  if (rows < 1 || group_rows < 0 || group_rows % RB != 0) return VT_EINVAL;
  const int nb = (int)((rows + RB - 1) / RB);
  hipLaunchKernelGGL(k_tr_bwd_c, dim3(nb), dim3(TPB), 0, s, da, n1_coef, wo, tok, wqkv, dr, dc, qkv, rows, group_rows);
  VT_CHECK_LAUNCH();
  return 0;
}
extern "C" int vt_transformer_bwd_c(const float *da, const float *n1_coef, const float *wo, const float *tok,
                                    const float *wqkv, float *dr, float *dc, float *qkv, long rows, vt_stream s) {
  return transformer_bwd_c_grouped(da, n1_coef, wo, tok, wqkv, dr, dc, qkv, rows, 0, (hipStream_t)s);
}
