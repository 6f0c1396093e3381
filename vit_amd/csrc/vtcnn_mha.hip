// MultiHeadAttention::backward, per-head part (reference layers/multi_head_attention.hpp:139-181),
// for gfx950, on the matrix cores: fp32 products formed from bf16 hi / lo operand halves
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation) in the row sweep, exact-f32 MFMA in the weight gradients.
//
// The layer's "batch" axis is the token index and its "sequence" axis is the mini-batch, because
// Tensor::permute relabels the shape without moving data (utils/tensor/tensor_cpu.hpp:719-747):
// attention group b = rows b*T .. b*T+T-1 of the flat [N*L, D] token matrix, T = N.  Only the
// weight gradients of the per-head query / key / value Dense layers survive: the forward output
// is multiplied by a zero concat (tensor_cpu.hpp:251) and the layer's input delta is discarded
// (layers/visual_transformer.hpp:160).  So the attention backward is restated row-wise:
//   scores[i][j] = K_i.Q_j / sqrt(d);  attn = softmax_j;  dattn[i][j] = dhead_i . V_j
//   dot_i = sum_j dattn*attn;  dsc = attn * (dattn - dot_i)     (no 1/sqrt(d) in backward, :165-170)
//   Y_i = sum_j attn[i][j] X_j;  Z_i = sum_j dsc[i][j] X_j
//   gWv = sum_i Y_i^T dhead_i;  gWq = sum_i Z_i^T K_i;  gWk = sum_i X_i^T (Z_i Wq)
// which equals X^T dV, X^T dQ, X^T dK of the reference without ever forming the T x T matrices
// in memory (flash-style: at T = 1024 they would be 4 MB per group and head).
//
// Kernel 1 (k_mha_rows): one wave = 32 rows i of one (head, group), sweeping j in tiles of 32.
// All five contractions are 32x32 MFMA tiles.  The score tile is computed TRANSPOSED,
//   S^T[j][i] = sum_e Q[j][e] K[i][e],   dA^T[j][i] = sum_e V[j][e] dhead[i][e],
// so that its accumulator layout (column i on the lane, rows j in the 16 registers) is directly
// the B operand of the next product, which contracts over j:
//   Y^T[d][i] += sum_j X[j][d] E^T[j][i],   W^T[d][i] += sum_j X[j][d] (E*dA)^T[j][i].
// No lane shuffles in the loop; per-row softmax quantities are per-lane scalars.
// Softmax is shift invariant: the sweep subtracts a per-row running shift that trails the row maximum
// (rescaling its sums when a tile overshoots it), so no pass over j comes before the sweep.
// Kernel 2 (k_mha_wgrad): the three [D x HD] gradient products over row blocks, slab + reduce.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "vtcnn_common.h"

namespace vt {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int MH_STRIDE = 3 * D * HD;  // per-workgroup partials: gq[D*HD] | gk[D*HD] | gv[D*HD]

// accumulator register v of lane half hh holds row (v&3) + 8*(v>>2) + 4*hh of a 32x32 tile
__device__ __forceinline__ int acc_row(int v, int hh) { return (v & 3) + 8 * (v >> 2) + 4 * hh; }

// Partial, unnormalised row sums of one j range ("split"), each relative to the split's own final shift r:
// the gradient kernel brings the splits of a row to their common largest shift before adding them.
//   yz [head][split][2 (Y~, W~)][D][rows]      st [head][split][3 (sum e, sum e*dattn, shift r)][rows]
// (row index fastest: coalesced for both kernels)

// fp32 products on the bf16 matrix pipe.  Every operand x is carried as two bf16 numbers,
// hi = bf16(x) and lo = bf16(x - hi), so hi + lo holds 16 mantissa bits of x, and a product of two
// tiles is the sum of the partial products hi.hi + hi.lo + lo.hi (+ lo.lo where the result feeds an
// exponential), each one v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The bf16 pipe runs 16
// times the rate of the fp32 one, so three or four partial products still cost a fifth of the cycles;
// the relative error per product is 2^-17, random in sign, far inside the 1e-4 of the parity tests
// after the sums over the mini-batch.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// (a, b) -> packed bf16 pair of the high parts and of the low parts
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo) {
  const floatx2 f = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
  const floatx2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
__device__ __forceinline__ void split2(floatx2 f, unsigned &hi, unsigned &lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
  const floatx2 r = f - floatx2{__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
// third piece: what the pair (hi, lo) still misses of (a, b); hi + lo + lo2 holds all 24 mantissa bits
__device__ __forceinline__ void split3(floatx2 f, unsigned &hi, unsigned &lo, unsigned &lo2) {
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
  const floatx2 r = f - floatx2{__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  const floatx2 r2 = r - floatx2{__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u)};
  lo2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}
__device__ __forceinline__ void split3(float a, float b, unsigned &hi, unsigned &lo, unsigned &lo2) {
  split3(floatx2{a, b}, hi, lo, lo2);
}
__device__ __forceinline__ bf16x8 as_bf8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ floatx16 mfma_bf(bf16x8 a, bf16x8 b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int MR_WAVES = 4;  // row tiles (waves) per workgroup sharing the staged j tiles
constexpr int XT = 36;       // LDS row stride (bf16 elements) of the transposed token tile: 72 bytes, conflict-free b64 reads

// One staged j tile: 32 token rows.  Q and V head slices as [row][16] bf16 (hi and lo arrays); the
// token rows transposed, [feature][row], so that a lane owning feature d reads four consecutive rows
// with one 8-byte load.
struct StageBuf {
  unsigned short qh[32 * HD], ql[32 * HD], vh[32 * HD], vl[32 * HD];
  unsigned short q3[32 * HD];  // third bf16 piece of Q (read by the high-precision sweep only)
  unsigned short xh[D * XT], xl[D * XT];
};

// FULL: T is a multiple of 32, so no row of any tile needs a mask.
// Workgroup = MR_WAVES row tiles of one (head, group, j range).  The j tile is fetched once per
// workgroup, two 16-byte loads per thread, split into bf16 hi / lo on its way into LDS (once, for
// all four waves) and double-buffered: the global loads of tile k+1 are issued before the arithmetic
// of tile k and land in LDS after it.
template <bool FULL>
__global__ __launch_bounds__(64 * MR_WAVES) void k_mha_rows(const float *__restrict__ qkv, const float *__restrict__ dc,
                                                            const float *__restrict__ tok, float *__restrict__ yz,
                                                            float *__restrict__ st, float *__restrict__ xbar, int T,
                                                            long rows, int groups, int nsplit, int prec_mode) {
  static_assert(HD == 16 && D == 32 && MR_WAVES == 4, "tile shapes are written for head_dim 16, token channels 32");
  __shared__ __attribute__((aligned(16))) StageBuf sb[3];
  __shared__ float sred[MR_WAVES];
  const int itiles = (T + 31) / 32, iblocks = (itiles + MR_WAVES - 1) / MR_WAVES;
  // Workgroups go to the eight XCDs round-robin, each XCD with an L2 of its own: the row tiles of one
  // (head, group) sweep the same j tiles, so they are given indices that land on one XCD.
  int bi = blockIdx.x;
  if ((gridDim.x & 7) == 0) bi = (bi & 7) * (gridDim.x >> 3) + (bi >> 3);
  const int sp = bi % nsplit; bi /= nsplit;
  const int ib = bi % iblocks; bi /= iblocks;
  const int bgrp = bi % groups, h = bi / groups;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hh = lane >> 5;
  const long gbase = (long)bgrp * T;
  const float *__restrict__ Qm = qkv, *__restrict__ Km = qkv + rows * D, *__restrict__ Vm = qkv + 2 * rows * D;
  const float scale = 1.0f / sqrtf((float)HD);
  const int i = (ib * MR_WAVES + wave) * 32 + c;
  const bool ok_i = i < T;
  const long row_i = gbase + (ok_i ? i : T - 1);
  // this split's j range, in whole tiles
  const int tps = (itiles + nsplit - 1) / nsplit;
  const int jbeg = sp * tps * 32, jend = min(T, (sp + 1) * tps * 32);
  const int ntile = jbeg < jend ? (jend - jbeg + 31) / 32 : 0;

  // B operands of the score products (this wave's own rows, constant over the sweep): lane half hh
  // carries head columns e = 8*hh .. 8*hh+7, the k slots of its half of the 32x32x16 product
  bf16x8 kbh, kbl, kb3, dbh, dbl;
  {
    const float4 k0 = *reinterpret_cast<const float4 *>(Km + row_i * D + h * HD + 8 * hh);
    const float4 k1 = *reinterpret_cast<const float4 *>(Km + row_i * D + h * HD + 8 * hh + 4);
    const float4 d0 = *reinterpret_cast<const float4 *>(dc + row_i * D + h * HD + 8 * hh);
    const float4 d1 = *reinterpret_cast<const float4 *>(dc + row_i * D + h * HD + 8 * hh + 4);
    uint4 ph, pl, p3;
    split3(k0.x, k0.y, ph.x, pl.x, p3.x); split3(k0.z, k0.w, ph.y, pl.y, p3.y); split3(k1.x, k1.y, ph.z, pl.z, p3.z); split3(k1.z, k1.w, ph.w, pl.w, p3.w);
    kbh = as_bf8(ph); kbl = as_bf8(pl); kb3 = as_bf8(p3);
    split2(d0.x, d0.y, ph.x, pl.x); split2(d0.z, d0.w, ph.y, pl.y); split2(d1.x, d1.y, ph.z, pl.z); split2(d1.z, d1.w, ph.w, pl.w);
    dbh = as_bf8(ph); dbl = as_bf8(pl);
  }
  // staging roles: thread -> one float4 of the token tile and one of the Q (threads 0..127) or V tile
  const int xr = tid >> 3, xc = (tid & 7) * 4, t2 = tid & 127, qr = t2 >> 2, qc = (t2 & 3) * 4;
  const float *__restrict__ gx = tok + gbase * D + xc;
  const float *__restrict__ gq = (tid < 128 ? Qm : Vm) + gbase * D + h * HD + qc;
  // Pivot of the group: the mean token row and the mean value row of its first 32 rows (any per-group constant
  // is exact algebraically; a sample mean removes the common component to within a fifth of the spread and
  // costs one tile of loads instead of a pass over the group).  The sweep works on token
  // rows and value rows RELATIVE to it: softmax_backward subtracts the attention-weighted mean of dattn, so a
  // constant added to every dattn of a row (dhead_i . V_mean) drops out exactly, and its row sums of dsc are
  // zero, so a constant token row drops out of Z = sum_j dsc X_j as well.  Without this, tokens with a large
  // component common to all rows (any trained model) make Z the small difference of two large sums, each
  // carrying the 2^-17 relative error of the bf16 hi / lo products; with near-uniform attention and the mean
  // as pivot both sums are small themselves.  The gradient kernel adds the mean back where the un-centred Y
  // is needed (it reads it from xbar, written by the first workgroup of each group).
  __shared__ float spiv[MR_WAVES][D + HD];
  float4 px, pq;
  {
    float4 ax = make_float4(0.f, 0.f, 0.f, 0.f), aq = ax;
    {  // the same loads and the same order in every workgroup of the group: the same bits
      const int j0 = 0;
      const float4 vx = *reinterpret_cast<const float4 *>(gx + (long)min(j0 + xr, T - 1) * D);
      const float4 vq = *reinterpret_cast<const float4 *>(gq + (long)min(j0 + qr, T - 1) * D);
      const float mx_ = j0 + xr < T ? 1.f : 0.f, mq_ = j0 + qr < T ? 1.f : 0.f;
      ax.x += mx_ * vx.x; ax.y += mx_ * vx.y; ax.z += mx_ * vx.z; ax.w += mx_ * vx.w;
      aq.x += mq_ * vq.x; aq.y += mq_ * vq.y; aq.z += mq_ * vq.z; aq.w += mq_ * vq.w;
    }
    // token sums: lanes with equal (tid & 7) hold the same four columns (rows xr = tid >> 3)
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      ax.x += __shfl_xor(ax.x, o); ax.y += __shfl_xor(ax.y, o); ax.z += __shfl_xor(ax.z, o); ax.w += __shfl_xor(ax.w, o);
    }
    // value sums (threads 128..255; the others summed Q rows, unused): equal (tid & 3) = same four columns
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
      aq.x += __shfl_xor(aq.x, o); aq.y += __shfl_xor(aq.y, o); aq.z += __shfl_xor(aq.z, o); aq.w += __shfl_xor(aq.w, o);
    }
    if (lane < 8) *reinterpret_cast<float4 *>(&spiv[wave][4 * lane]) = ax;
    if (lane < 4) *reinterpret_cast<float4 *>(&spiv[wave][D + 4 * lane]) = aq;
    __syncthreads();
    const float invT = 1.0f / (float)min(T, 32);
    const int xq = (tid & 7) * 4, vq_ = D + (tid & 3) * 4;
    px.x = (spiv[0][xq] + spiv[1][xq] + spiv[2][xq] + spiv[3][xq]) * invT;
    px.y = (spiv[0][xq + 1] + spiv[1][xq + 1] + spiv[2][xq + 1] + spiv[3][xq + 1]) * invT;
    px.z = (spiv[0][xq + 2] + spiv[1][xq + 2] + spiv[2][xq + 2] + spiv[3][xq + 2]) * invT;
    px.w = (spiv[0][xq + 3] + spiv[1][xq + 3] + spiv[2][xq + 3] + spiv[3][xq + 3]) * invT;
    pq.x = tid < 128 ? 0.f : (spiv[2][vq_] + spiv[3][vq_]) * invT;
    pq.y = tid < 128 ? 0.f : (spiv[2][vq_ + 1] + spiv[3][vq_ + 1]) * invT;
    pq.z = tid < 128 ? 0.f : (spiv[2][vq_ + 2] + spiv[3][vq_ + 2]) * invT;
    pq.w = tid < 128 ? 0.f : (spiv[2][vq_ + 3] + spiv[3][vq_ + 3]) * invT;
    if (h == 0 && ib == 0 && sp == 0 && tid < 8) *reinterpret_cast<float4 *>(xbar + (size_t)bgrp * D + xq) = px;
  }
  // two register sets: a tile's global loads are issued two sweep steps before it is needed in LDS
  // (one step of arithmetic is shorter than a trip to memory)
  float4 rxa, rqa, rxb, rqb;
  auto gload = [&](int jt, float4 &rx, float4 &rq) {
    rx = *reinterpret_cast<const float4 *>(gx + (long)min(jt + xr, T - 1) * D);
    rq = *reinterpret_cast<const float4 *>(gq + (long)min(jt + qr, T - 1) * D);
  };
  auto sstore = [&](int buf, const float4 &rx_, const float4 &rq_) {
    StageBuf &B = sb[buf];
    // relative to the pivot; here and not at the load, whose data should stay in flight until now
    const float4 rx = make_float4(rx_.x - px.x, rx_.y - px.y, rx_.z - px.z, rx_.w - px.w);
    const float4 rq = make_float4(rq_.x - pq.x, rq_.y - pq.y, rq_.z - pq.z, rq_.w - pq.w);
    uint2 hv, lv, tv;
    split3(rq.x, rq.y, hv.x, lv.x, tv.x);
    split3(rq.z, rq.w, hv.y, lv.y, tv.y);
    unsigned short *dh = tid < 128 ? B.qh : B.vh, *dl = tid < 128 ? B.ql : B.vl;
    *reinterpret_cast<uint2 *>(dh + qr * HD + qc) = hv;
    *reinterpret_cast<uint2 *>(dl + qr * HD + qc) = lv;
    if (tid < 128) *reinterpret_cast<uint2 *>(B.q3 + qr * HD + qc) = tv;
    split2(rx.x, rx.y, hv.x, lv.x);
    split2(rx.z, rx.w, hv.y, lv.y);
    B.xh[(xc + 0) * XT + xr] = (unsigned short)hv.x; B.xh[(xc + 1) * XT + xr] = (unsigned short)(hv.x >> 16);
    B.xh[(xc + 2) * XT + xr] = (unsigned short)hv.y; B.xh[(xc + 3) * XT + xr] = (unsigned short)(hv.y >> 16);
    B.xl[(xc + 0) * XT + xr] = (unsigned short)lv.x; B.xl[(xc + 1) * XT + xr] = (unsigned short)(lv.x >> 16);
    B.xl[(xc + 2) * XT + xr] = (unsigned short)lv.y; B.xl[(xc + 3) * XT + xr] = (unsigned short)(lv.y >> 16);
  };
  if (ntile > 0) {
    gload(jbeg, rxa, rqa);
    gload(jbeg + 32 * min(1, ntile - 1), rxb, rqb);
  }

  if (ntile > 0) {
    sstore(0, rxa, rqa);
    sstore(1, rxb, rqb);
    gload(jbeg + 32 * min(2, ntile - 1), rxa, rqa);  // stored by sweep step 0
  }
  __syncthreads();
  // e = exp(scale * s - r) as one fused multiply-add and one exp2, with a per-row running shift r (below)
  const float sc2 = scale * 1.44269504088896341f;
  floatx16 accY, accW;
#pragma unroll
  for (int v = 0; v < 16; ++v) { accY[v] = 0.f; accW[v] = 0.f; }
  floatx2 ssum2 = {0.f, 0.f}, tsum2 = {0.f, 0.f};
  const floatx2 sc2v = {sc2, sc2};
  // Running shift of row i (binary orders), the flash-attention way but lazy: it follows the row maximum
  // only when a tile's maximum exceeds it by more than LAZY orders, so numerators stay below 2^LAZY (fine in
  // fp32 and in the bf16 pieces) and the accumulators are rescaled a few times per sweep at most.  No bound
  // on the scores is needed (a bound can sit hundreds of orders above the true maximum once the weights
  // have grown, and every numerator of the row then flushes to zero), and no pass over the group up front.
  constexpr float LAZY = 8.0f;
  float r = -INFINITY;

  // With one j range per row tile a SIMD holds a single wave, so the matrix pipe and the vector pipe
  // only overlap if independent work of both kinds sits next to each other in that wave's instruction
  // stream.  The sweep is therefore software-pipelined over three staged tiles: while the vector
  // pipe turns tile k's scores into softmax numerators, the matrix pipe forms tile k+1's scores, and
  // the second-stage products of one half tile run beside the vector work of the other half.
  // Transposed score and dattn tiles of one staged tile: rows j (this lane: j = c) x columns i,
  // contraction over the 16 head columns = one matrix instruction per partial product.
  auto scores = [&](auto prec, const StageBuf &B, floatx16 &aS, floatx16 &aD) {
    constexpr bool P = decltype(prec)::value;
    const bf16x8 qh = as_bf8(*reinterpret_cast<const uint4 *>(B.qh + c * HD + 8 * hh));
    const bf16x8 ql = as_bf8(*reinterpret_cast<const uint4 *>(B.ql + c * HD + 8 * hh));
    const bf16x8 vh = as_bf8(*reinterpret_cast<const uint4 *>(B.vh + c * HD + 8 * hh));
    const bf16x8 vl = as_bf8(*reinterpret_cast<const uint4 *>(B.vl + c * HD + 8 * hh));
#pragma unroll
    for (int v = 0; v < 16; ++v) { aS[v] = 0.f; aD[v] = 0.f; }
    if constexpr (P) {
      const bf16x8 q3 = as_bf8(*reinterpret_cast<const uint4 *>(B.q3 + c * HD + 8 * hh));
      aS = mfma_bf(q3, kbh, aS);  // smallest partial products first
      aS = mfma_bf(qh, kb3, aS);
    }
    aS = mfma_bf(ql, kbl, aS);
    aD = mfma_bf(vl, dbh, aD);
    aS = mfma_bf(ql, kbh, aS);
    aD = mfma_bf(vh, dbl, aD);
    aS = mfma_bf(qh, kbl, aS);
    aD = mfma_bf(vh, dbh, aD);
    aS = mfma_bf(qh, kbh, aS);
  };
  // one sweep step: tile k's scores are in (aS, aD); tile k+1's go to (nS, nD)
  // (rxs, rqs): tile k+2, loaded one step ago, goes to LDS at the end of this step; (rxl, rql) receives tile k+3
  auto step = [&](auto prec, int k, const floatx16 &aS, const floatx16 &aD, floatx16 &nS, floatx16 &nD, const float4 &rxs,
                  const float4 &rqs, float4 &rxl, float4 &rql) {
    constexpr bool P = decltype(prec)::value;
    const int jt = jbeg + 32 * k;
    gload(jbeg + 32 * min(k + 3, ntile - 1), rxl, rql);  // unconditional: past the end it re-reads the final tile
    scores(prec, sb[(k + 1) % 3], nS, nD);     // (past the end: the final tile again, result unused)
    const StageBuf &B = sb[k % 3];
    // softmax numerators e and e * dattn, then their hi / lo halves as the B operands of the second
    // stage: registers 0..7 of lane half hh are tile rows {4hh..4hh+3, 8+4hh..}, registers 8..15 the
    // same + 16; the token operand presents its rows in that same order
    {
      float tm = -INFINITY;
#pragma unroll
      for (int v = 0; v < 16; ++v) tm = fmaxf(tm, (FULL || jt + acc_row(v, hh) < T) ? aS[v] : -INFINITY);
      tm = fmaxf(tm, __shfl_xor(tm, 32)) * sc2;  // both lane halves of column i agree
      if (__any(tm > r + LAZY)) {
        const float rn = tm > r + LAZY ? tm : r;
        const float alpha = __builtin_amdgcn_exp2f(r - rn);  // first tile: r = -inf, alpha = 0 on zero sums
#pragma unroll
        for (int v = 0; v < 16; ++v) { accY[v] *= alpha; accW[v] *= alpha; }
        ssum2 *= alpha;
        tsum2 *= alpha;
        r = rn;
      }
    }
    const floatx2 nmxv = {-r, -r};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      unsigned ph[4], pl[4], qh4[4], ql4[4], p3[4] = {0, 0, 0, 0}, q34[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // two adjacent accumulator registers at a time: the multiply-adds, products, sums and the
        // hi / lo remainders are packed fp32 instructions (two lanes' worth of work per issue slot)
        const int v = 8 * half + 2 * q;
        const floatx2 t = floatx2{aS[v], aS[v + 1]} * sc2v + nmxv;
        floatx2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        if (!FULL) {
          e.x = (jt + acc_row(v, hh) < T) ? e.x : 0.f;
          e.y = (jt + acc_row(v + 1, hh) < T) ? e.y : 0.f;
        }
        const floatx2 ed = e * floatx2{aD[v], aD[v + 1]};
        ssum2 += e;
        tsum2 += ed;
        if constexpr (P) {
          split3(e, ph[q], pl[q], p3[q]);
          split3(ed, qh4[q], ql4[q], q34[q]);
        } else {
          split2(e, ph[q], pl[q]);
          split2(ed, qh4[q], ql4[q]);
        }
      }
      const uint4 eh = make_uint4(ph[0], ph[1], ph[2], ph[3]), el = make_uint4(pl[0], pl[1], pl[2], pl[3]);
      const uint4 fh = make_uint4(qh4[0], qh4[1], qh4[2], qh4[3]), fl = make_uint4(ql4[0], ql4[1], ql4[2], ql4[3]);
      // Y^T[d][i] += sum_j X[j][d] e^T[j][i],  W^T[d][i] += sum_j X[j][d] (e*dattn)^T[j][i]: this lane owns
      // feature d = c; one instruction (16 rows) per partial product and half
      const unsigned short *xph = B.xh + c * XT + 16 * half + 4 * hh, *xpl = B.xl + c * XT + 16 * half + 4 * hh;
      const uint2 h0 = *reinterpret_cast<const uint2 *>(xph), h1 = *reinterpret_cast<const uint2 *>(xph + 8);
      const uint2 l0 = *reinterpret_cast<const uint2 *>(xpl), l1 = *reinterpret_cast<const uint2 *>(xpl + 8);
      const bf16x8 xh = as_bf8(make_uint4(h0.x, h0.y, h1.x, h1.y)), xl = as_bf8(make_uint4(l0.x, l0.y, l1.x, l1.y));
      if constexpr (P) {
        const uint4 e3 = make_uint4(p3[0], p3[1], p3[2], p3[3]), f3 = make_uint4(q34[0], q34[1], q34[2], q34[3]);
        accY = mfma_bf(xh, as_bf8(e3), accY);
        accW = mfma_bf(xh, as_bf8(f3), accW);
        accY = mfma_bf(xl, as_bf8(el), accY);
        accW = mfma_bf(xl, as_bf8(fl), accW);
      }
      accY = mfma_bf(xl, as_bf8(eh), accY);
      accW = mfma_bf(xl, as_bf8(fh), accW);
      accY = mfma_bf(xh, as_bf8(el), accY);
      accW = mfma_bf(xh, as_bf8(fl), accW);
      accY = mfma_bf(xh, as_bf8(eh), accY);
      accW = mfma_bf(xh, as_bf8(fh), accW);
    }
    sstore((k + 2) % 3, rxs, rqs);  // the buffer of tile k-1: every wave left it before the last barrier
    __syncthreads();
  };
  // Precision of the sweep, one decision per workgroup.  Two bf16 pieces per operand carry 16 mantissa bits:
  // a score is then off by about 2^-17 of |K_i||Q_j|, and the softmax numerator by that many nats.  Harmless
  // while the scores are of order one (initialisation scale: gradients within 1e-6 of exact arithmetic), but
  // the query / key weights never influence the forward pass (the zero concat), so nothing holds them back
  // during training: their gradients are unnormalised sums over the whole mini-batch, and within tens of
  // steps the scores reach the tens and hundreds: saturated rows, where the softmax backward is a difference
  // of nearly equal terms.  When the first staged tile shows scaled scores beyond PRECISE_ABOVE binary
  // orders, the workgroup carries Q, K and both softmax numerators with three pieces (24 bits): six partial
  // products for the scores instead of four, five instead of three in the second stage.
  constexpr float PRECISE_ABOVE = 1.5f;
  bool precise = prec_mode == 2;
  if (prec_mode == 0) {
    float ma = 0.f;
    if (ntile > 0) {
      floatx16 ts, td;
      scores(std::false_type{}, sb[0], ts, td);
#pragma unroll
      for (int v = 0; v < 16; ++v) ma = fmaxf(ma, fabsf(ts[v]));
    }
    ma = wave_max(ma);
    if (lane == 0) sred[wave] = ma;
    __syncthreads();
    precise = sc2 * fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3])) > PRECISE_ABOVE;
  }
  auto sweep = [&](auto prec) {
    floatx16 s0, d0, s1, d1;
    if (ntile > 0) scores(prec, sb[0], s0, d0);
    for (int k = 0; k < ntile; k += 2) {
      step(prec, k, s0, d0, s1, d1, rxa, rqa, rxb, rqb);
      if (k + 1 < ntile) step(prec, k + 1, s1, d1, s0, d0, rxb, rqb, rxa, rqa);
    }
  };
  if (precise) sweep(std::true_type{});
  else sweep(std::false_type{});
  float ssum = ssum2.x + ssum2.y, tsum = tsum2.x + tsum2.y;
  ssum += __shfl_xor(ssum, 32);
  tsum += __shfl_xor(tsum, 32);
  if (ok_i) {
    const size_t hs = (size_t)h * nsplit + sp;
    float *yb = yz + (hs * 2 + 0) * D * rows + gbase + i;
    float *wb = yz + (hs * 2 + 1) * D * rows + gbase + i;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int d = acc_row(v, hh);
      yb[(size_t)d * rows] = accY[v];
      wb[(size_t)d * rows] = accW[v];
    }
    if (hh == 0) {
      st[(hs * 3 + 0) * rows + gbase + i] = ssum;
      st[(hs * 3 + 1) * rows + gbase + i] = tsum;
      st[(hs * 3 + 2) * rows + gbase + i] = r;
    }
  }
}

// Merges the splits (Y = Y~/s, Z = W~/s - (t/s) Y) and forms, over a block of RB rows of one head,
//   gq = Z^T K,  gk = X^T (Z Wq),  gv = Y^T dhead.
// One 16-row tile per wave, v_mfma_f32_16x16x4_f32.  The three products contract over rows, so
// their operands are wanted with the feature on the lane and the rows in registers: that is how
// yz is stored (row index fastest), one 16-byte load per feature and split.  dK = Z.Wq contracts
// over features instead and takes Z with the row on the lane (plain coalesced loads of the same
// array); with Z as the A operand its result lands in the registers-are-rows layout that gk needs.
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// NSPLIT: the number of j ranges, a compile-time constant so that the loops over them unroll and every load
// of the kernel can be in flight at once (with one range its shift factors are exactly 1 and drop out).
template <int NSPLIT>
__global__ __launch_bounds__(TPB) void k_mha_wgrad(const float *__restrict__ yz, const float *__restrict__ st,
                                                   const float *__restrict__ qkv, const float *__restrict__ dc,
                                                   const float *__restrict__ tok, const float *__restrict__ Wqkv,
                                                   float *__restrict__ slab, const float *__restrict__ xbar, long rows,
                                                   int single_key, int T) {
  constexpr int nsplit = NSPLIT;
  __shared__ float sbuf[6][NW * 256];  // the six gradient tiles of the four waves, summed in one phase
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4, h = blockIdx.y;
  const long row0 = ((long)blockIdx.x * NW + wave) * 16;
  const int nv = (int)max(0L, min(16L, rows - row0));
  const float *__restrict__ Km = qkv + rows * D;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  // per-row softmax normalisers: for the registers-are-rows operands (rows 4g+v) and for row j
  f4 inv4 = zero, dot4 = zero;
  float invj = 0.f, dotj = 0.f;
  // every split's sums are relative to its own shift r: factor exp2(r_split - r_max) brings them together
  // (exactly 1 with a single split; 0 for a split that saw no tile, whose r is -inf)
  f4 rm4 = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  float rmj = -INFINITY;
  if constexpr (NSPLIT > 1) {
#pragma unroll
    for (int sp = 0; sp < nsplit; ++sp) {
      const float *s2 = st + (((size_t)h * nsplit + sp) * 3 + 2) * rows;
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (4 * g + v < nv) rm4[v] = fmaxf(rm4[v], s2[row0 + 4 * g + v]);
      if (j < nv) rmj = fmaxf(rmj, s2[row0 + j]);
    }
  }
  auto fac4 = [&](int sp) {
    f4 f = zero;
    if constexpr (NSPLIT == 1) {  // exp2(r - r) = 1 exactly
#pragma unroll
      for (int v = 0; v < 4; ++v) f[v] = 4 * g + v < nv ? 1.f : 0.f;
    } else {
      const float *s2 = st + (((size_t)h * nsplit + sp) * 3 + 2) * rows;
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (4 * g + v < nv) f[v] = exp2f(s2[row0 + 4 * g + v] - rm4[v]);
    }
    return f;
  };
  auto facj = [&](int sp) {
    if constexpr (NSPLIT == 1) return j < nv ? 1.f : 0.f;
    else return j < nv ? exp2f(st[(((size_t)h * nsplit + sp) * 3 + 2) * rows + row0 + j] - rmj) : 0.f;
  };
  {
    f4 s4 = zero, t4 = zero;
    float sj = 0.f, tj = 0.f;
#pragma unroll
    for (int sp = 0; sp < nsplit; ++sp) {
      const float *s0 = st + (((size_t)h * nsplit + sp) * 3 + 0) * rows, *s1 = s0 + rows;
      const f4 f = fac4(sp);
      const float fj = facj(sp);
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (4 * g + v < nv) {
          s4[v] += s0[row0 + 4 * g + v] * f[v];
          t4[v] += s1[row0 + 4 * g + v] * f[v];
        }
      if (j < nv) {
        sj += s0[row0 + j] * fj;
        tj += s1[row0 + j] * fj;
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (4 * g + v < nv) {
        inv4[v] = s4[v] > 0.f ? 1.0f / s4[v] : 0.f;  // a row whose numerators all flushed to zero contributes nothing
        dot4[v] = t4[v] * inv4[v];
      }
    if (j < nv) {
      invj = sj > 0.f ? 1.0f / sj : 0.f;
      dotj = tj * invj;
    }
  }
  // Y, Z with the feature on the lane (feature j + 16t, rows 4g+v), and Z with the row on the lane
  f4 yn[2], zn[2], zt[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f4 y = zero, w = zero;
#pragma unroll
    for (int sp = 0; sp < nsplit; ++sp) {
      const float *yb = yz + ((((size_t)h * nsplit + sp) * 2 + 0) * D + (j + 16 * t)) * rows + row0 + 4 * g;
      const float *wb = yb + (size_t)D * rows;
      const f4 f = fac4(sp);
      if (4 * g + 3 < nv) {  // the four rows of this lane group are consecutive floats, 16-byte aligned
        const f4 y4 = *reinterpret_cast<const f4 *>(yb), w4 = *reinterpret_cast<const f4 *>(wb);
        y += y4 * f;
        w += w4 * f;
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (4 * g + v < nv) {
            y[v] += yb[v] * f[v];
            w[v] += wb[v] * f[v];
          }
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      yn[t][v] = y[v] * inv4[v];
      // a group of one row attends only to itself: its softmax is exactly 1 and softmax_backward
      // exactly 0 (tensor_cpu.hpp:946-950), which the difference of two rounded terms would miss
      zn[t][v] = single_key ? 0.f : w[v] * inv4[v] - dot4[v] * yn[t][v];
      // the row sums came in relative to the group's mean token row (k_mha_rows); Z is independent of it, the
      // attention-weighted token mean Y (for the value gradient) is not: the weights sum to one
      if (4 * g + v < nv) yn[t][v] += xbar[((row0 + 4 * g + v) / T) * D + j + 16 * t];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float y1 = 0.f, w1 = 0.f;
      if (j < nv)
#pragma unroll
        for (int sp = 0; sp < nsplit; ++sp) {
          const size_t o = ((((size_t)h * nsplit + sp) * 2 + 0) * D + (16 * t + 4 * g + v)) * rows + row0 + j;
          const float fj = facj(sp);
          y1 += yz[o] * fj;
          w1 += yz[o + (size_t)D * rows] * fj;
        }
      y1 *= invj;
      zt[t][v] = single_key ? 0.f : w1 * invj - dotj * y1;
    }
  }
  // dK[row][e] = sum_d Z[d][row] Wq[d][e]  (rows in registers)
  f4 dkn = zero;
  {
    const float *__restrict__ Wq = Wqkv + (size_t)(h * 3 + 0) * D * HD;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) dkn = mfma16(zt[t][v], Wq[(16 * t + 4 * g + v) * HD + j], dkn);
  }
  f4 kn, dhn, xn[2];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const bool ok = 4 * g + v < nv;
    const long r = row0 + 4 * g + v;
    kn[v] = ok ? Km[r * D + h * HD + j] : 0.f;
    dhn[v] = ok ? dc[r * D + h * HD + j] : 0.f;
    xn[0][v] = ok ? tok[r * D + j] : 0.f;
    xn[1][v] = ok ? tok[r * D + 16 + j] : 0.f;
  }
  float *row = slab + ((size_t)h * gridDim.x + blockIdx.x) * MH_STRIDE;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f4 gq = zero, gk = zero, gv = zero;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      gq = mfma16(zn[t][v], kn[v], gq);
      gk = mfma16(xn[t][v], dkn[v], gk);
      gv = mfma16(yn[t][v], dhn[v], gv);
    }
    const f4 acc[3] = {gq, gk, gv};
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int v = 0; v < 4; ++v) sbuf[3 * t + m][wave * 256 + (4 * g + v) * 16 + j] = acc[m][v];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const float *b = sbuf[3 * t + m];
      row[m * D * HD + (16 * t) * HD + tid] = (b[tid] + b[256 + tid]) + (b[512 + tid] + b[768 + tid]);
    }
}

}  // namespace vt

using namespace vt;

// How many j ranges each row tile is split into so that the launch has one wave per SIMD (1024).
// A second resident wave buys the sweep 10 % and costs the gradient kernel more than that: every
// split adds a set of partial sums that it must read and merge.
static int mha_splits(int N, int groups) {
  const int itiles = (N + 31) / 32;
  static const int forced = [] {  // measurement aid: VT_MHA_SPLITS=n forces the number of j ranges (1..4)
    const char *e = getenv("VT_MHA_SPLITS");
    return e ? atoi(e) : 0;
  }();
  if (forced >= 1 && forced <= 4) return forced < itiles ? forced : itiles;
  const long waves = (long)HEADS * groups * ((itiles + MR_WAVES - 1) / MR_WAVES) * MR_WAVES;
  int ns = (int)((1024 + waves - 1) / waves);
  if (ns > 4) ns = 4;
  if (ns > itiles) ns = itiles;
  return ns < 1 ? 1 : ns;
}

// The group means of the token rows [groups][D] sit behind the largest layout of the other scratch regions
// (four j splits): heads*4*(2*D + 3)*rows + heads*ceil(rows/64)*1536 floats.
static float *mha_xbar(float *scratch, long rows) {
  return scratch + (size_t)HEADS * 4 * (2 * D + 3) * rows + (size_t)HEADS * ((rows + RB - 1) / RB) * MH_STRIDE;
}

// Measurement aid: VT_MHA_PRECISION=two / three forces the sweep's operand precision (default: chosen per
// workgroup from the score magnitudes).
static int mha_precision_mode() {
  static const int mode = [] {
    const char *e = getenv("VT_MHA_PRECISION");
    if (!e) return 0;
    return strcmp(e, "two") == 0 ? 1 : (strcmp(e, "three") == 0 ? 2 : 0);
  }();
  return mode;
}

// Step 1: per-row attention sums (k_mha_rows).  scratch as for vt_mha_bwd.
extern "C" int vt_mha_bwd_rows(const float *qkv, const float *dc, const float *tok, float *scratch, int N, int groups,
                               vt_stream s) {
  if (N < 1 || groups < 1) return VT_EINVAL;
  const int itiles = (N + 31) / 32, ns = mha_splits(N, groups);
  const long rows = (long)N * groups;
  float *yz = scratch;
  float *st = yz + (size_t)HEADS * ns * 2 * D * rows;
  float *xbar = mha_xbar(scratch, rows);
  const int iblocks = (itiles + MR_WAVES - 1) / MR_WAVES;
  if (N % 32 == 0)
    hipLaunchKernelGGL(k_mha_rows<true>, dim3(HEADS * groups * iblocks * ns), dim3(64 * MR_WAVES), 0, (hipStream_t)s,
                       qkv, dc, tok, yz, st, xbar, N, rows, groups, ns, mha_precision_mode());
  else
    hipLaunchKernelGGL(k_mha_rows<false>, dim3(HEADS * groups * iblocks * ns), dim3(64 * MR_WAVES), 0, (hipStream_t)s,
                       qkv, dc, tok, yz, st, xbar, N, rows, groups, ns, mha_precision_mode());
  VT_CHECK_LAUNCH();
  return 0;
}

// Step 2: merge the j splits and form the six gradient tensors from the row sums of step 1.
extern "C" int vt_mha_bwd_wgrad(const float *qkv, const float *dc, const float *tok, const float *wqkv, float *g_qkv,
                                float *scratch, int N, int groups, vt_stream s) {
  if (N < 1 || groups < 1) return VT_EINVAL;
  const int ns = mha_splits(N, groups);
  const long rows = (long)N * groups;
  const int nrb = (int)((rows + RB - 1) / RB);
  float *yz = scratch;
  float *st = yz + (size_t)HEADS * ns * 2 * D * rows;
  float *slab = st + (size_t)HEADS * ns * 3 * rows;
#define VT_WGRAD(NS)                                                                                               \
  hipLaunchKernelGGL(k_mha_wgrad<NS>, dim3(nrb, HEADS), dim3(TPB), 0, (hipStream_t)s, yz, st, qkv, dc, tok, wqkv, slab, \
                     mha_xbar(scratch, rows), rows, N == 1 ? 1 : 0, N)
  switch (ns) {
    case 1: VT_WGRAD(1); break;
    case 2: VT_WGRAD(2); break;
    case 3: VT_WGRAD(3); break;
    default: VT_WGRAD(4); break;
  }
#undef VT_WGRAD
  VT_CHECK_LAUNCH();
  for (int h = 0; h < HEADS; ++h) {
    const int rc = reduce_rows(slab + (size_t)h * nrb * MH_STRIDE, nrb, MH_STRIDE, MH_STRIDE,
                               g_qkv + (size_t)h * MH_STRIDE, (hipStream_t)s);
    if (rc) return rc;
  }
  return 0;
}

// g_qkv: [q0 | k0 | v0 | q1 | k1 | v1] gradients (arena order);
// scratch >= heads*4*(2*D + 3)*rows + heads*ceil(rows/64)*1536 + groups*D floats with rows = N*groups.
extern "C" int vt_mha_bwd(const float *qkv, const float *dc, const float *tok, const float *wqkv, float *g_qkv,
                          float *scratch, int N, int groups, vt_stream s) {
  const int rc = vt_mha_bwd_rows(qkv, dc, tok, scratch, N, groups, s);
  if (rc) return rc;
  return vt_mha_bwd_wgrad(qkv, dc, tok, wqkv, g_qkv, scratch, N, groups, s);
}
