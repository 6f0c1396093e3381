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
// Softmax uses a per-row upper bound of the scores instead of the exact row maximum
// (|K_i.Q_j| <= |K_i| max_j |Q_j|): softmax is shift invariant, so only the rounding changes, and
// a whole pass over j disappears.
// Kernel 2 (k_mha_wgrad): the three [D x HD] gradient products over row blocks, slab + reduce.
#include "vtcnn_common.h"

namespace vt {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int MH_STRIDE = 3 * D * HD;  // per-workgroup partials: gq[D*HD] | gk[D*HD] | gv[D*HD]

// accumulator register v of lane half hh holds row (v&3) + 8*(v>>2) + 4*hh of a 32x32 tile
__device__ __forceinline__ int acc_row(int v, int hh) { return (v & 3) + 8 * (v >> 2) + 4 * hh; }

// Partial, unnormalised row sums of one j range ("split"): because every split uses the same
// per-row score bound, partials from different splits simply add.
//   yz [head][split][2 (Y~, W~)][D][rows]      st [head][split][2 (sum e, sum e*dattn)][rows]
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
                                                            float *__restrict__ st, int T, long rows, int groups,
                                                            int nsplit) {
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
  bf16x8 kbh, kbl, dbh, dbl;
  float kn2 = 0.f;
  {
    const float4 k0 = *reinterpret_cast<const float4 *>(Km + row_i * D + h * HD + 8 * hh);
    const float4 k1 = *reinterpret_cast<const float4 *>(Km + row_i * D + h * HD + 8 * hh + 4);
    const float4 d0 = *reinterpret_cast<const float4 *>(dc + row_i * D + h * HD + 8 * hh);
    const float4 d1 = *reinterpret_cast<const float4 *>(dc + row_i * D + h * HD + 8 * hh + 4);
    uint4 ph, pl;
    split2(k0.x, k0.y, ph.x, pl.x); split2(k0.z, k0.w, ph.y, pl.y); split2(k1.x, k1.y, ph.z, pl.z); split2(k1.z, k1.w, ph.w, pl.w);
    kbh = as_bf8(ph); kbl = as_bf8(pl);
    split2(d0.x, d0.y, ph.x, pl.x); split2(d0.z, d0.w, ph.y, pl.y); split2(d1.x, d1.y, ph.z, pl.z); split2(d1.z, d1.w, ph.w, pl.w);
    dbh = as_bf8(ph); dbl = as_bf8(pl);
    kn2 = k0.x * k0.x + k0.y * k0.y + k0.z * k0.z + k0.w * k0.w + k1.x * k1.x + k1.y * k1.y + k1.z * k1.z + k1.w * k1.w;
  }
  // staging roles: thread -> one float4 of the token tile and one of the Q (threads 0..127) or V tile
  const int xr = tid >> 3, xc = (tid & 7) * 4, t2 = tid & 127, qr = t2 >> 2, qc = (t2 & 3) * 4;
  const float *__restrict__ gx = tok + gbase * D + xc;
  const float *__restrict__ gq = (tid < 128 ? Qm : Vm) + gbase * D + h * HD + qc;
  // two register sets: a tile's global loads are issued two sweep steps before it is needed in LDS
  // (one step of arithmetic is shorter than a trip to memory)
  float4 rxa, rqa, rxb, rqb;
  auto gload = [&](int jt, float4 &rx, float4 &rq) {
    rx = *reinterpret_cast<const float4 *>(gx + (long)min(jt + xr, T - 1) * D);
    rq = *reinterpret_cast<const float4 *>(gq + (long)min(jt + qr, T - 1) * D);
  };
  auto sstore = [&](int buf, const float4 &rx, const float4 &rq) {
    StageBuf &B = sb[buf];
    uint2 hv, lv;
    split2(rq.x, rq.y, hv.x, lv.x);
    split2(rq.z, rq.w, hv.y, lv.y);
    unsigned short *dh = tid < 128 ? B.qh : B.vh, *dl = tid < 128 ? B.ql : B.vl;
    *reinterpret_cast<uint2 *>(dh + qr * HD + qc) = hv;
    *reinterpret_cast<uint2 *>(dl + qr * HD + qc) = lv;
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

  kn2 += __shfl_xor(kn2, 32);
  // largest |Q_j|^2 of the whole group (identical in every split and row tile)
  float qn2 = 0.f;
  for (int j = tid; j < T; j += 64 * MR_WAVES) {
    float sq = 0.f;
#pragma unroll
    for (int q = 0; q < HD / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(Qm + (gbase + j) * D + h * HD + 4 * q);
      sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    qn2 = fmaxf(qn2, sq);
  }
  qn2 = wave_max(qn2);
  if (lane == 0) sred[wave] = qn2;
  if (ntile > 0) {
    sstore(0, rxa, rqa);
    sstore(1, rxb, rqb);
    gload(jbeg + 32 * min(2, ntile - 1), rxa, rqa);  // stored by sweep step 0
  }
  __syncthreads();
  qn2 = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
  // e = exp(scale * s - mx) with mx = scale |K_i| max|Q_j| >= every scaled score of row i, evaluated
  // as one fused multiply-add and one exp2
  const float sc2 = scale * 1.44269504088896341f;
  const float nmx = -sc2 * sqrtf(kn2) * sqrtf(qn2);

  floatx16 accY, accW;
#pragma unroll
  for (int v = 0; v < 16; ++v) { accY[v] = 0.f; accW[v] = 0.f; }
  floatx2 ssum2 = {0.f, 0.f}, tsum2 = {0.f, 0.f};
  const floatx2 sc2v = {sc2, sc2}, nmxv = {nmx, nmx};

  // With one j range per row tile a SIMD holds a single wave, so the matrix pipe and the vector pipe
  // only overlap if independent work of both kinds sits next to each other in that wave's instruction
  // stream.  The sweep is therefore software-pipelined over three staged tiles: while the vector
  // pipe turns tile k's scores into softmax numerators, the matrix pipe forms tile k+1's scores, and
  // the second-stage products of one half tile run beside the vector work of the other half.
  // Transposed score and dattn tiles of one staged tile: rows j (this lane: j = c) x columns i,
  // contraction over the 16 head columns = one matrix instruction per partial product.
  auto scores = [&](const StageBuf &B, floatx16 &aS, floatx16 &aD) {
    const bf16x8 qh = as_bf8(*reinterpret_cast<const uint4 *>(B.qh + c * HD + 8 * hh));
    const bf16x8 ql = as_bf8(*reinterpret_cast<const uint4 *>(B.ql + c * HD + 8 * hh));
    const bf16x8 vh = as_bf8(*reinterpret_cast<const uint4 *>(B.vh + c * HD + 8 * hh));
    const bf16x8 vl = as_bf8(*reinterpret_cast<const uint4 *>(B.vl + c * HD + 8 * hh));
#pragma unroll
    for (int v = 0; v < 16; ++v) { aS[v] = 0.f; aD[v] = 0.f; }
    aS = mfma_bf(ql, kbl, aS);  // smallest partial product first
    aD = mfma_bf(vl, dbh, aD);
    aS = mfma_bf(ql, kbh, aS);
    aD = mfma_bf(vh, dbl, aD);
    aS = mfma_bf(qh, kbl, aS);
    aD = mfma_bf(vh, dbh, aD);
    aS = mfma_bf(qh, kbh, aS);
  };
  // one sweep step: tile k's scores are in (aS, aD); tile k+1's go to (nS, nD)
  // (rxs, rqs): tile k+2, loaded one step ago, goes to LDS at the end of this step; (rxl, rql) receives tile k+3
  auto step = [&](int k, const floatx16 &aS, const floatx16 &aD, floatx16 &nS, floatx16 &nD, const float4 &rxs,
                  const float4 &rqs, float4 &rxl, float4 &rql) {
    const int jt = jbeg + 32 * k;
    gload(jbeg + 32 * min(k + 3, ntile - 1), rxl, rql);  // unconditional: past the end it re-reads the final tile
    scores(sb[(k + 1) % 3], nS, nD);           // (past the end: the final tile again, result unused)
    const StageBuf &B = sb[k % 3];
    // softmax numerators e and e * dattn, then their hi / lo halves as the B operands of the second
    // stage: registers 0..7 of lane half hh are tile rows {4hh..4hh+3, 8+4hh..}, registers 8..15 the
    // same + 16; the token operand presents its rows in that same order
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      unsigned ph[4], pl[4], qh4[4], ql4[4];
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
        split2(e, ph[q], pl[q]);
        split2(ed, qh4[q], ql4[q]);
      }
      const uint4 eh = make_uint4(ph[0], ph[1], ph[2], ph[3]), el = make_uint4(pl[0], pl[1], pl[2], pl[3]);
      const uint4 fh = make_uint4(qh4[0], qh4[1], qh4[2], qh4[3]), fl = make_uint4(ql4[0], ql4[1], ql4[2], ql4[3]);
      // Y^T[d][i] += sum_j X[j][d] e^T[j][i],  W^T[d][i] += sum_j X[j][d] (e*dattn)^T[j][i]: this lane owns
      // feature d = c; one instruction (16 rows) per partial product and half
      const unsigned short *xph = B.xh + c * XT + 16 * half + 4 * hh, *xpl = B.xl + c * XT + 16 * half + 4 * hh;
      const uint2 h0 = *reinterpret_cast<const uint2 *>(xph), h1 = *reinterpret_cast<const uint2 *>(xph + 8);
      const uint2 l0 = *reinterpret_cast<const uint2 *>(xpl), l1 = *reinterpret_cast<const uint2 *>(xpl + 8);
      const bf16x8 xh = as_bf8(make_uint4(h0.x, h0.y, h1.x, h1.y)), xl = as_bf8(make_uint4(l0.x, l0.y, l1.x, l1.y));
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
  floatx16 s0, d0, s1, d1;
  if (ntile > 0) scores(sb[0], s0, d0);
  for (int k = 0; k < ntile; k += 2) {
    step(k, s0, d0, s1, d1, rxa, rqa, rxb, rqb);
    if (k + 1 < ntile) step(k + 1, s1, d1, s0, d0, rxb, rqb, rxa, rqa);
  }
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
      st[(hs * 2 + 0) * rows + gbase + i] = ssum;
      st[(hs * 2 + 1) * rows + gbase + i] = tsum;
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

__global__ __launch_bounds__(TPB) void k_mha_wgrad(const float *__restrict__ yz, const float *__restrict__ st,
                                                   const float *__restrict__ qkv, const float *__restrict__ dc,
                                                   const float *__restrict__ tok, const float *__restrict__ Wqkv,
                                                   float *__restrict__ slab, long rows, int nsplit, int single_key) {
  __shared__ float sbuf[6][NW * 256];  // the six gradient tiles of the four waves, summed in one phase
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4, h = blockIdx.y;
  const long row0 = ((long)blockIdx.x * NW + wave) * 16;
  const int nv = (int)max(0L, min(16L, rows - row0));
  const float *__restrict__ Km = qkv + rows * D;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  // per-row softmax normalisers: for the registers-are-rows operands (rows 4g+v) and for row j
  f4 inv4 = zero, dot4 = zero;
  float invj = 0.f, dotj = 0.f;
  {
    f4 s4 = zero, t4 = zero;
    float sj = 0.f, tj = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) {
      const float *s0 = st + (((size_t)h * nsplit + sp) * 2 + 0) * rows, *s1 = s0 + rows;
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (4 * g + v < nv) {
          s4[v] += s0[row0 + 4 * g + v];
          t4[v] += s1[row0 + 4 * g + v];
        }
      if (j < nv) {
        sj += s0[row0 + j];
        tj += s1[row0 + j];
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (4 * g + v < nv) {
        inv4[v] = 1.0f / s4[v];
        dot4[v] = t4[v] * inv4[v];
      }
    if (j < nv) {
      invj = 1.0f / sj;
      dotj = tj * invj;
    }
  }
  // Y, Z with the feature on the lane (feature j + 16t, rows 4g+v), and Z with the row on the lane
  f4 yn[2], zn[2], zt[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f4 y = zero, w = zero;
    for (int sp = 0; sp < nsplit; ++sp) {
      const float *yb = yz + ((((size_t)h * nsplit + sp) * 2 + 0) * D + (j + 16 * t)) * rows + row0 + 4 * g;
      const float *wb = yb + (size_t)D * rows;
      if (4 * g + 3 < nv) {  // the four rows of this lane group are consecutive floats, 16-byte aligned
        const f4 y4 = *reinterpret_cast<const f4 *>(yb), w4 = *reinterpret_cast<const f4 *>(wb);
        y += y4;
        w += w4;
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (4 * g + v < nv) {
            y[v] += yb[v];
            w[v] += wb[v];
          }
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      yn[t][v] = y[v] * inv4[v];
      // a group of one row attends only to itself: its softmax is exactly 1 and softmax_backward
      // exactly 0 (tensor_cpu.hpp:946-950), which the difference of two rounded terms would miss
      zn[t][v] = single_key ? 0.f : w[v] * inv4[v] - dot4[v] * yn[t][v];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float y1 = 0.f, w1 = 0.f;
      if (j < nv)
        for (int sp = 0; sp < nsplit; ++sp) {
          const size_t o = ((((size_t)h * nsplit + sp) * 2 + 0) * D + (16 * t + 4 * g + v)) * rows + row0 + j;
          y1 += yz[o];
          w1 += yz[o + (size_t)D * rows];
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
  const long waves = (long)HEADS * groups * ((itiles + MR_WAVES - 1) / MR_WAVES) * MR_WAVES;
  int ns = (int)((1024 + waves - 1) / waves);
  if (ns > 4) ns = 4;
  if (ns > itiles) ns = itiles;
  return ns < 1 ? 1 : ns;
}

// Step 1: per-row attention sums (k_mha_rows).  scratch as for vt_mha_bwd.
extern "C" int vt_mha_bwd_rows(const float *qkv, const float *dc, const float *tok, float *scratch, int N, int groups,
                               vt_stream s) {
  if (N < 1 || groups < 1) return VT_EINVAL;
  const int itiles = (N + 31) / 32, ns = mha_splits(N, groups);
  const long rows = (long)N * groups;
  float *yz = scratch;
  float *st = yz + (size_t)HEADS * ns * 2 * D * rows;
  const int iblocks = (itiles + MR_WAVES - 1) / MR_WAVES;
  if (N % 32 == 0)
    hipLaunchKernelGGL(k_mha_rows<true>, dim3(HEADS * groups * iblocks * ns), dim3(64 * MR_WAVES), 0, (hipStream_t)s,
                       qkv, dc, tok, yz, st, N, rows, groups, ns);
  else
    hipLaunchKernelGGL(k_mha_rows<false>, dim3(HEADS * groups * iblocks * ns), dim3(64 * MR_WAVES), 0, (hipStream_t)s,
                       qkv, dc, tok, yz, st, N, rows, groups, ns);
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
  float *slab = st + (size_t)HEADS * ns * 2 * rows;
  hipLaunchKernelGGL(k_mha_wgrad, dim3(nrb, HEADS), dim3(TPB), 0, (hipStream_t)s, yz, st, qkv, dc, tok, wqkv, slab, rows,
                     ns, N == 1 ? 1 : 0);
  VT_CHECK_LAUNCH();
  for (int h = 0; h < HEADS; ++h) {
    const int rc = reduce_rows(slab + (size_t)h * nrb * MH_STRIDE, nrb, MH_STRIDE, MH_STRIDE,
                               g_qkv + (size_t)h * MH_STRIDE, (hipStream_t)s);
    if (rc) return rc;
  }
  return 0;
}

// g_qkv: [q0 | k0 | v0 | q1 | k1 | v1] gradients (arena order);
// scratch >= heads*4*(2*D + 2)*rows + heads*ceil(rows/64)*1536 floats with rows = N*groups.
extern "C" int vt_mha_bwd(const float *qkv, const float *dc, const float *tok, const float *wqkv, float *g_qkv,
                          float *scratch, int N, int groups, vt_stream s) {
  const int rc = vt_mha_bwd_rows(qkv, dc, tok, scratch, N, groups, s);
  if (rc) return rc;
  return vt_mha_bwd_wgrad(qkv, dc, tok, wqkv, g_qkv, scratch, N, groups, s);
}
