// Strided-batched fp32 GEMM on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products and
// accumulation) for the shapes whose contractions are real matrix products: every DenseLayer
// forward / backward (layers/dense_layer.hpp:67-100, Tensor::matmul tensor_cpu.hpp:754-814) and the
// batched products of tokenizer, projector and attention once channels and tokens are wide.
//
//   C[b] = alpha * op(A[b]) . op(B[b]) (+ bias, broadcast over rows) (+ C[b] when accumulate)
//
// Row-major operands with leading dimensions; op = identity or transpose.  One workgroup of four
// waves owns a 64x64 tile of C (each wave 32x32 = 2x2 MFMA tiles) and walks K in steps of 16 through
// LDS.  A long K with a small C (weight gradients: K = rows of the batch) is cut into `ksplit`
// slices whose partial tiles go to a workspace and are summed in a fixed order by a second kernel
// (no float atomics: the result does not depend on scheduling).
// Three arithmetic forms of the same kernel (template parameter PR): 0 = exact fp32 matrix instructions;
// 1 = operands rounded to bf16 (v_mfma_f32_16x16x16_bf16, fp32 accumulation); 2 = fp32-accurate products on
// the bf16 pipe: every operand is carried as hi = bf16(x), lo = bf16(x - hi) and a product is
// lo.hi + hi.lo + hi.hi (relative error about 2^-17 per product, fp32 accumulation), three matrix
// instructions of a quarter of the fp32 instruction's duration each in place of four fp32 ones.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "../../include/vtcnn_hip.h"
#include "vtcnn_common.h"

namespace vt {

constexpr int GBK = 16, GPAD = 4;

struct GemmP {
  const float *A, *B;
  float *C;
  const float *bias;
  int M, N, K, lda, ldb, ldc;
  long sA, sB, sC;
  float alpha;
  int accumulate, ksplit, kchunk, batch, vecA, vecB;
  float *part;  // [ksplit][batch][M*N] when ksplit > 1
  const float *bz_coef;  // kernels built with BZ: B holds delta records (load_tile_rec), ldb floats per row of records
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One operand tile (R = 64 or 128 rows of the M or N side x 16 of K) from global memory into R/16
// registers per thread.  ALONG_K: the K index is the contiguous one in memory (A not transposed / B
// transposed).  Vector form (16-byte loads) when the host found the operand aligned and its sizes
// multiples of 4.
template <bool ALONG_K, int R>
__device__ __forceinline__ void load_tile(const float *__restrict__ base, int ld, int r0, int rows, int k0, int kend,
                                          bool vec, int tid, float (&reg)[R / 16]) {
  if (vec) {
#pragma unroll
    for (int hb = 0; hb < R / 64; ++hb) {
      int r, k;
      if (ALONG_K) { r = (tid >> 2) + 64 * hb; k = (tid & 3) * 4; } else { k = tid >> 4; r = (tid & 15) * 4 + 64 * hb; }
      const int gr = r0 + r, gk = k0 + k;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < rows && gk < kend)
        t = *reinterpret_cast<const float4 *>(base + (ALONG_K ? (long)gr * ld + gk : (long)gk * ld + gr));
      reg[4 * hb + 0] = t.x; reg[4 * hb + 1] = t.y; reg[4 * hb + 2] = t.z; reg[4 * hb + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < R / 16; ++e) {
      const int idx = tid + e * 256;
      int r, k;
      if (ALONG_K) { k = idx & (GBK - 1); r = idx >> 4; } else { r = idx & (R - 1); k = idx / R; }
      const int gr = r0 + r, gk = k0 + k;
      reg[e] = (gr < rows && gk < kend) ? base[ALONG_K ? (long)gr * ld + gk : (long)gk * ld + gr] : 0.f;
    }
  }
}
// registers -> LDS tile [k][row]
template <bool ALONG_K, int R>
__device__ __forceinline__ void store_tile(float (*T)[R + GPAD], bool vec, int tid, const float (&reg)[R / 16]) {
  if (vec) {
#pragma unroll
    for (int hb = 0; hb < R / 64; ++hb) {
      if (ALONG_K) {
        const int r = (tid >> 2) + 64 * hb, k = (tid & 3) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) T[k + e][r] = reg[4 * hb + e];
      } else {
        const int k = tid >> 4, r = (tid & 15) * 4 + 64 * hb;
#pragma unroll
        for (int e = 0; e < 4; ++e) T[k][r + e] = reg[4 * hb + e];
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < R / 16; ++e) {
      const int idx = tid + e * 256;
      if (ALONG_K) T[idx & (GBK - 1)][idx >> 4] = reg[e]; else T[idx / R][idx & (R - 1)] = reg[e];
    }
  }
}

// bf16 mode: tiles are kept [row][k] in bf16 (round to nearest even), four consecutive k per lane feed
// v_mfma_f32_16x16x16_bf16 (fp32 accumulation): a quarter of the matrix instructions of the fp32 form
constexpr int HPAD = 4;
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned short to_bf16(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float from_bf16(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// (a, b) -> the packed bf16 pair of their high parts and of their low parts (a in the low half), with the packed
// hardware conversion (round to nearest even, as to_bf16) and a packed fp32 subtraction
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo) {
  const floatx2 f = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
  const floatx2 r = f - floatx2{__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}
// the same tile as hi / lo halves (form 2)
template <bool ALONG_K, int R>
__device__ __forceinline__ void store_tile_split(unsigned short (*Th)[GBK + HPAD], unsigned short (*Tl)[GBK + HPAD], bool vec,
                                                 int tid, const float (&reg)[R / 16]) {
  unsigned short h[R / 16], l[R / 16];
#pragma unroll
  for (int e = 0; e < R / 16; e += 2) {
    unsigned hp, lp;
    split2(reg[e], reg[e + 1], hp, lp);
    h[e] = (unsigned short)hp; h[e + 1] = (unsigned short)(hp >> 16);
    l[e] = (unsigned short)lp; l[e + 1] = (unsigned short)(lp >> 16);
  }
  if (vec) {
#pragma unroll
    for (int hb = 0; hb < R / 64; ++hb) {
      if (ALONG_K) {
        const int r = (tid >> 2) + 64 * hb, k = (tid & 3) * 4;
        *reinterpret_cast<uint2 *>(&Th[r][k]) = make_uint2(h[4 * hb] | ((unsigned)h[4 * hb + 1] << 16), h[4 * hb + 2] | ((unsigned)h[4 * hb + 3] << 16));
        *reinterpret_cast<uint2 *>(&Tl[r][k]) = make_uint2(l[4 * hb] | ((unsigned)l[4 * hb + 1] << 16), l[4 * hb + 2] | ((unsigned)l[4 * hb + 3] << 16));
      } else {
        const int k = tid >> 4, r = (tid & 15) * 4 + 64 * hb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Th[r + e][k] = h[4 * hb + e];
          Tl[r + e][k] = l[4 * hb + e];
        }
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < R / 16; ++e) {
      const int idx = tid + e * 256;
      if (ALONG_K) {
        Th[idx >> 4][idx & (GBK - 1)] = h[e];
        Tl[idx >> 4][idx & (GBK - 1)] = l[e];
      } else {
        Th[idx & (R - 1)][idx / R] = h[e];
        Tl[idx & (R - 1)][idx / R] = l[e];
      }
    }
  }
}
template <bool ALONG_K, int R>
__device__ __forceinline__ void store_tile_bf16(unsigned short (*T)[GBK + HPAD], bool vec, int tid, const float (&reg)[R / 16]) {
  if (vec) {
#pragma unroll
    for (int hb = 0; hb < R / 64; ++hb) {
      if (ALONG_K) {
        const int r = (tid >> 2) + 64 * hb, k = (tid & 3) * 4;
        const unsigned lo = to_bf16(reg[4 * hb]) | ((unsigned)to_bf16(reg[4 * hb + 1]) << 16);
        const unsigned hi = to_bf16(reg[4 * hb + 2]) | ((unsigned)to_bf16(reg[4 * hb + 3]) << 16);
        *reinterpret_cast<uint2 *>(&T[r][k]) = make_uint2(lo, hi);
      } else {
        const int k = tid >> 4, r = (tid & 15) * 4 + 64 * hb;
#pragma unroll
        for (int e = 0; e < 4; ++e) T[r + e][k] = to_bf16(reg[4 * hb + e]);
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < R / 16; ++e) {
      const int idx = tid + e * 256;
      if (ALONG_K) T[idx >> 4][idx & (GBK - 1)] = to_bf16(reg[e]); else T[idx & (R - 1)][idx / R] = to_bf16(reg[e]);
    }
  }
}

// Operand whose contiguous index is the row (A transposed, B as stored) on its way into the bf16 layouts [row][k]:
// a thread takes four consecutive k of ONE row (four 4-byte loads, each coalesced over the 64 rows of its wave), so
// that its LDS store is one 8-byte word per half instead of four scattered 2-byte stores.
template <int R>
__device__ __forceinline__ void load_tile_kq(const float *__restrict__ base, int ld, int r0, int rows, int k0, int kend,
                                             int tid, float (&reg)[R / 16]) {
#pragma unroll
  for (int hb = 0; hb < R / 64; ++hb) {
    const int gr = r0 + (tid & 63) + 64 * hb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gk = k0 + 4 * (tid >> 6) + j;
      reg[4 * hb + j] = (gr < rows && gk < kend) ? base[(long)gk * ld + gr] : 0.f;
    }
  }
}
template <int PR, int R>
__device__ __forceinline__ void store_tile_kq(unsigned short (*Th)[GBK + HPAD], unsigned short (*Tl)[GBK + HPAD], int tid,
                                              const float (&reg)[R / 16]) {
#pragma unroll
  for (int hb = 0; hb < R / 64; ++hb) {
    const int r = (tid & 63) + 64 * hb, k = 4 * (tid >> 6);
    unsigned h01, l01, h23, l23;
    split2(reg[4 * hb], reg[4 * hb + 1], h01, l01);
    split2(reg[4 * hb + 2], reg[4 * hb + 3], h23, l23);
    *reinterpret_cast<uint2 *>(&Th[r][k]) = make_uint2(h01, h23);
    if constexpr (PR == 2) *reinterpret_cast<uint2 *>(&Tl[r][k]) = make_uint2(l01, l23);
  }
}
// B given as delta records instead of a dense matrix (GemmP::bz_coef): per row k and group of 16 adjacent columns
// one pair (value, column inside the group), every other element of the group zero, and
// B[k][n] = coef[N + n] + coef[n] * delta[k][n].  A K step's records (16 rows x R / 16 groups: a few hundred bytes)
// are fetched by the first threads, one pair each, and passed through LDS; every thread then expands the elements
// the mapping of load_tile_kq gives it, so that the tile is staged by the same code as a dense one.
template <int R>
__device__ __forceinline__ float2 load_rec(const float *__restrict__ rec, int ld, int r0, int rows, int k0, int kend, int tid) {
  constexpr int NG = R / 16;
  const int k = tid / NG, grp = tid - k * NG, gk = k0 + k, n = r0 + 16 * grp;
  if (tid < GBK * NG && gk < kend && n < rows) return *reinterpret_cast<const float2 *>(rec + (long)gk * ld + 2 * (n >> 4));
  return make_float2(0.f, __int_as_float(-1));
}
template <int R>
__device__ __forceinline__ void expand_rec(const float2 *__restrict__ buf, const float (&ca)[R / 64], const float (&cb)[R / 64],
                                           int k0, int kend, int tid, float (&reg)[R / 16]) {
  constexpr int NG = R / 16;
#pragma unroll
  for (int hb = 0; hb < R / 64; ++hb) {
    const int col = (tid & 63) + 64 * hb;  // column inside the tile
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kq = 4 * (tid >> 6) + j;
      const float2 rc = buf[kq * NG + (col >> 4)];
      const float d = __float_as_int(rc.y) == (col & 15) ? rc.x : 0.f;
      reg[4 * hb + j] = k0 + kq < kend ? fmaf(d, ca[hb], cb[hb]) : 0.f;
    }
  }
}
// memory -> registers in the mapping the kernel's form stages from
template <bool ALONG_K, int PR, int R>
__device__ __forceinline__ void fetch_tile(const float *__restrict__ base, int ld, int r0, int rows, int k0, int kend, bool vec,
                                           int tid, float (&reg)[R / 16]) {
  if constexpr (!ALONG_K && PR != 0) load_tile_kq<R>(base, ld, r0, rows, k0, kend, tid, reg);
  else load_tile<ALONG_K, R>(base, ld, r0, rows, k0, kend, vec, tid, reg);
}

// LDS tiles of one workgroup (only the arrays of the kernel's arithmetic form have a size)
template <int PR, int BM, int BN>
struct GemmLds {
  float As[PR != 0 ? 1 : GBK][BM + GPAD];
  float Bs[PR != 0 ? 1 : GBK][BN + GPAD];
  __attribute__((aligned(8))) unsigned short Ah[PR != 0 ? BM : 1][GBK + HPAD];
  __attribute__((aligned(8))) unsigned short Bh[PR != 0 ? BN : 1][GBK + HPAD];
  __attribute__((aligned(8))) unsigned short Al[PR == 2 ? BM : 1][GBK + HPAD];
  __attribute__((aligned(8))) unsigned short Bl[PR == 2 ? BN : 1][GBK + HPAD];
};

// registers -> LDS in the kernel's arithmetic form
template <bool ALONG_K, int PR, int R>
__device__ __forceinline__ void stage_tile(float (*Tf)[R + GPAD], unsigned short (*Th)[GBK + HPAD],
                                           unsigned short (*Tl)[GBK + HPAD], bool vec, int tid, const float (&reg)[R / 16]) {
  if constexpr (!ALONG_K && PR != 0) store_tile_kq<PR, R>(Th, Tl, tid, reg);
  else if constexpr (PR == 2) store_tile_split<ALONG_K, R>(Th, Tl, vec, tid, reg);
  else if constexpr (PR == 1) store_tile_bf16<ALONG_K, R>(Th, vec, tid, reg);
  else store_tile<ALONG_K, R>(Tf, vec, tid, reg);
}

// acc += (the BM x 16 tile of A in LDS) . (the 16 x BN tile of B in LDS): each of the four waves owns
// (BM/2) x (BN/2) as 16x16 matrix-instruction tiles (waves laid out 2 x 2)
template <int PR, int BM, int BN>
__device__ __forceinline__ void gemm_step(const GemmLds<PR, BM, BN> &L, f32x4 (&acc)[BM / 32][BN / 32]) {
  constexpr int TM = BM / 32, TN = BN / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  if constexpr (PR == 2) {
    const int kq = 4 * (lane >> 4), c = lane & 15;
    s16x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = *reinterpret_cast<const s16x4 *>(&L.Ah[wm + 16 * i + c][kq]);
      al[i] = *reinterpret_cast<const s16x4 *>(&L.Al[wm + 16 * i + c][kq]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = *reinterpret_cast<const s16x4 *>(&L.Bh[wn + 16 * j + c][kq]);
      bl[j] = *reinterpret_cast<const s16x4 *>(&L.Bl[wn + 16 * j + c][kq]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {  // the two small terms first
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  } else if constexpr (PR == 1) {
    const int kq = 4 * (lane >> 4), c = lane & 15;
    s16x4 av[TM], bv[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const s16x4 *>(&L.Ah[wm + 16 * i + c][kq]);
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const s16x4 *>(&L.Bh[wn + 16 * j + c][kq]);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av[i], bv[j], acc[i][j], 0, 0, 0);
  } else {
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 4) {
      const int kr = kk + (lane >> 4), c = lane & 15;
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = L.As[kr][wm + 16 * i + c];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = L.Bs[kr][wn + 16 * j + c];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
}

// the tile's accumulators to memory: r = acc * alpha (+ bias[n]) (+ C); finish = false stores the bare products
template <int BM, int BN>
__device__ __forceinline__ void gemm_store(const f32x4 (&acc)[BM / 32][BN / 32], float *__restrict__ C, int ldc, int M, int N,
                                           int m0, int n0, float alpha, const float *__restrict__ bias, bool accumulate,
                                           bool finish) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
#pragma unroll
  for (int i = 0; i < BM / 32; ++i)
#pragma unroll
    for (int j = 0; j < BN / 32; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int gm = m0 + wm + i * 16 + 4 * (lane >> 4) + v, gn = n0 + wn + j * 16 + (lane & 15);
        if (gm < M && gn < N) {
          float r = acc[i][j][v] * alpha;
          if (finish) {
            if (bias) r += bias[gn];
            if (accumulate) r += C[(long)gm * ldc + gn];
          }
          C[(long)gm * ldc + gn] = r;
        }
      }
}

// Which tile a workgroup takes.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, and with it
// an L2), so the dispatch index is first regrouped by XCD (index b -> (b % 8) * total / 8 + b / 8), and the regrouped
// index walks the N tiles fastest, then the M tiles, then batch members / K slices: the workgroups resident on one XCD
// at a time then cover every N tile of a run of M tiles, and an operand tile that several of them need comes out of
// that XCD's L2 instead of out of memory once per consumer.  A speed hint only: nothing depends on the placement.
__device__ __forceinline__ void tile_of_block(unsigned &bx, unsigned &by, unsigned &bz) {
  const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
  const unsigned bid = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const unsigned lid = (total & 7u) == 0 ? (bid & 7u) * (total >> 3) + (bid >> 3) : bid;
  by = lid % gy;
  const unsigned t = lid / gy;
  bx = t % gx;
  bz = t / gx;
}

// BM x BN: the workgroup's tile of C (64x64 for narrow or small products, 64x128 when only N is
// wide, 128x128 for the wide shapes: each wave then owns 64x64 = 4x4 MFMA tiles and an LDS operand
// read feeds four matrix instructions instead of two).
template <bool TA, bool TB, int PR, int BM, int BN, bool BZ = false>
__global__ __launch_bounds__(256) void k_gemm(GemmP p) {
  __shared__ GemmLds<PR, BM, BN> L;
  unsigned bx, by, bz;
  tile_of_block(bx, by, bz);
  const int b = bz / p.ksplit, ks = bz - b * p.ksplit;
  const int m0 = bx * BM, n0 = by * BN;  // rows on x: the batch's pixel rows run into the millions
  const int kbeg = ks * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  f32x4 acc[BM / 32][BN / 32];
#pragma unroll
  for (int i = 0; i < BM / 32; ++i)
#pragma unroll
    for (int j = 0; j < BN / 32; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float *__restrict__ A = p.A + (long)b * p.sA;
  const float *__restrict__ B = p.B + (long)b * p.sB;
  const int tid = threadIdx.x;
  const bool vecA = p.vecA != 0, vecB = p.vecB != 0;
  // the next K step's tiles travel from memory while the matrix cores work on the current one
  float ra[BM / 16], rb[BN / 16];
  fetch_tile<!TA, PR, BM>(A, p.lda, m0, p.M, kbeg, kend, vecA, tid, ra);
  if constexpr (BZ) {
    // records: step s's pairs sit in recs[s % 2] (written while step s - 1 was staged), step s + 1's are in flight
    __shared__ float2 recs[2][GBK * (BN / 16)];
    float ca[BN / 64], cb[BN / 64];
#pragma unroll
    for (int hb = 0; hb < BN / 64; ++hb) {
      const int gn = n0 + (tid & 63) + 64 * hb;
      ca[hb] = gn < p.N ? p.bz_coef[gn] : 0.f;
      cb[hb] = gn < p.N ? p.bz_coef[p.N + gn] : 0.f;
    }
    float2 rnext = load_rec<BN>(B, p.ldb, n0, p.N, kbeg, kend, tid);
    if (tid < GBK * (BN / 16)) recs[0][tid] = rnext;
    rnext = load_rec<BN>(B, p.ldb, n0, p.N, kbeg + GBK, kend, tid);
    __syncthreads();
    int par = 0;
    for (int k0 = kbeg; k0 < kend; k0 += GBK, par ^= 1) {
      expand_rec<BN>(recs[par], ca, cb, k0, kend, tid, rb);
      __syncthreads();  // the previous step's tile reads are over
      stage_tile<!TA, PR, BM>(L.As, L.Ah, L.Al, vecA, tid, ra);
      stage_tile<TB, PR, BN>(L.Bs, L.Bh, L.Bl, vecB, tid, rb);
      if (tid < GBK * (BN / 16)) recs[par ^ 1][tid] = rnext;
      __syncthreads();
      if (k0 + GBK < kend) {
        fetch_tile<!TA, PR, BM>(A, p.lda, m0, p.M, k0 + GBK, kend, vecA, tid, ra);
        rnext = load_rec<BN>(B, p.ldb, n0, p.N, k0 + 2 * GBK, kend, tid);
      }
      gemm_step<PR, BM, BN>(L, acc);
    }
  } else {
    fetch_tile<TB, PR, BN>(B, p.ldb, n0, p.N, kbeg, kend, vecB, tid, rb);
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
      __syncthreads();  // the previous step's tile reads are over
      stage_tile<!TA, PR, BM>(L.As, L.Ah, L.Al, vecA, tid, ra);
      stage_tile<TB, PR, BN>(L.Bs, L.Bh, L.Bl, vecB, tid, rb);
      __syncthreads();
      if (k0 + GBK < kend) {
        fetch_tile<!TA, PR, BM>(A, p.lda, m0, p.M, k0 + GBK, kend, vecA, tid, ra);
        fetch_tile<TB, PR, BN>(B, p.ldb, n0, p.N, k0 + GBK, kend, vecB, tid, rb);
      }
      gemm_step<PR, BM, BN>(L, acc);
    }
  }
  const bool direct = p.ksplit == 1;
  float *__restrict__ C = direct ? p.C + (long)b * p.sC : p.part + ((long)ks * p.batch + b) * (long)p.M * p.N;
  gemm_store<BM, BN>(acc, C, direct ? p.ldc : p.N, p.M, p.N, m0, n0, p.alpha, p.bias, p.accumulate != 0, direct);
}

// A sum of up to four products into one tile of C without the partial results travelling through memory:
//   C[b] = alpha * sum_t A_t[b] . op(B_t[b]) (+ bias) (+ C[b])
// (the projector output x.Wd + attention.values; the input delta that four layers contribute to).  A is never
// transposed here; B is per term.
struct GemmTerm {
  const float *A, *B;
  long sA, sB;
  int K, lda, ldb, tb, vecA, vecB;
};
struct GemmSumP {
  GemmTerm t[4];
  int nterms;
  float *C;
  const float *bias;
  int M, N, ldc;
  long sC;
  float alpha;
  int accumulate;
  // row-panel kernel only: 1 = softmax over the row (Tensor::softmax over the last axis), 2 = its
  // backward r <- a * (r - sum_row a * r) with a = epi_a (laid out like C); 0 = none
  int epi;
  const float *epi_a;
  // epi 3: C is stored as it is, and per 16-row tile and column the softmax-over-rows record of the tile goes to
  // tile_rec [ceil(M / 16) per batch member][2][N]: the maximum of the tile's rows and the sum of exp(value - maximum)
  float *tile_rec;
  // epi 4: C is stored as it is, and every workgroup leaves the BatchNorm partial record of its rows (k_colstats' format:
  // sum, sum of squared distances from the workgroup's own mean) in col_part [batch][workgroups][2][N]; the second
  // moments are accumulated around col_pivot [N] (may be null: zero)
  float *col_part;
  long col_part_floats;
  const float *col_pivot;
  int c_first_row_only;  // epi 4: of C only row 0 of batch member 0 is stored (C is recomputed where it is needed)
  // epi 5: C is not stored; it is normalised (BatchNorm apply with pl_stat = mean | 1/sigma, pl_gamma, pl_beta), passed
  // through the ReLU and max-pooled in the layout the reference's reshape gives it: batch member b's M x N values, flat,
  // are N planes of pl_H x pl_H positions, pooled by pl_pk x pl_pk windows to 14 x 14 (vit_cnn.hpp:179,
  // layers/pooling_layer.hpp:18-66; the first maximum of a window in row-major order wins).  Per window: the pooled value,
  // the position of its maximum inside the plane, and the value of C there.
  const float *pl_stat, *pl_gamma, *pl_beta;
  float *pl_pooled, *pl_cmax;
  int *pl_idx;
  int pl_H, pl_pk;
  // row-panel kernels built with DZ: term 0's A holds delta records (one pair per row and group of 16 adjacent k:
  // value, k inside the group; every other element of the group zero) and A[m][k] = dz_coef[K + k] + dz_coef[k] * delta
  const float *dz_coef;
  // AX 4: term 0's A rows are the im2col rows of a 3x3 / stride 1 / pad 1 convolution over im_X [images][im_Ci][im_H][im_H]
  // (row = image * H * H + h * H + w, column = (ci * 3 + kh) * 3 + kw), gathered on load and written to ax_out
  const float *im_X;
  int im_Ci, im_H;
  // row-panel kernels built with AX == 2: term 0's A is a BatchNorm's input, normalised on load,
  //   A'[m][k] = (A[m][k] - ax_stat[k]) * ax_stat[K + k] * ax_gamma[k] + ax_beta[k]
  // (BatchNorm2DLayer::forward's apply, layers/batch_normalization_layer.hpp:292-297), and A' is also written to
  // ax_out (laid out like A): the first reader of a BatchNorm output produces it instead of a pass of its own
  const float *ax_stat, *ax_gamma, *ax_beta;
  float *ax_out;
};
template <int PR, int BM, int BN>
__global__ __launch_bounds__(256) void k_gemm_sum(GemmSumP p) {
  __shared__ GemmLds<PR, BM, BN> L;
  unsigned bx, by, bz;
  tile_of_block(bx, by, bz);
  const int b = bz;
  const int m0 = bx * BM, n0 = by * BN;
  f32x4 acc[BM / 32][BN / 32];
#pragma unroll
  for (int i = 0; i < BM / 32; ++i)
#pragma unroll
    for (int j = 0; j < BN / 32; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // one K walk over all terms (a term's last step is padded with zeros to 16), so that the loads of a term's
  // first tile travel under the previous term's last matrix instructions.  `cur` describes the tile held in
  // registers, `nxt` the one to request; a term is picked by comparisons, not by indexing (an indexed read of
  // the by-value argument would move all four terms into vector registers).
  const int tid = threadIdx.x;
  auto term = [&](int ti) { return ti == 0 ? p.t[0] : ti == 1 ? p.t[1] : ti == 2 ? p.t[2] : p.t[3]; };
  auto fetch = [&](const GemmTerm &t, int k0, float (&ra)[BM / 16], float (&rb)[BN / 16]) {
    fetch_tile<true, PR, BM>(t.A + (long)b * t.sA, t.lda, m0, p.M, k0, t.K, t.vecA != 0, tid, ra);
    if (t.tb) fetch_tile<true, PR, BN>(t.B + (long)b * t.sB, t.ldb, n0, p.N, k0, t.K, t.vecB != 0, tid, rb);
    else fetch_tile<false, PR, BN>(t.B + (long)b * t.sB, t.ldb, n0, p.N, k0, t.K, t.vecB != 0, tid, rb);
  };
  float ra[BM / 16], rb[BN / 16];
  int ti = 0, k0 = 0;
  GemmTerm cur = term(0);
  fetch(cur, 0, ra, rb);
  while (true) {
    __syncthreads();  // the previous step's tile reads are over
    stage_tile<true, PR, BM>(L.As, L.Ah, L.Al, cur.vecA != 0, tid, ra);
    if (cur.tb) stage_tile<true, PR, BN>(L.Bs, L.Bh, L.Bl, cur.vecB != 0, tid, rb);
    else stage_tile<false, PR, BN>(L.Bs, L.Bh, L.Bl, cur.vecB != 0, tid, rb);
    __syncthreads();
    k0 += GBK;
    if (k0 >= cur.K) {
      k0 = 0;
      ++ti;
      if (ti < p.nterms) cur = term(ti);
    }
    const bool more = ti < p.nterms;
    if (more) fetch(cur, k0, ra, rb);
    gemm_step<PR, BM, BN>(L, acc);
    if (!more) break;
  }
  gemm_store<BM, BN>(acc, p.C + (long)b * p.sC, p.ldc, p.M, p.N, m0, n0, p.alpha, p.bias, p.accumulate != 0, true);
}

// ---- tall and narrow results (M in the millions of pixel rows, N <= 208, total depth a few hundred): the
// row-panel form of the same sum of products.  Such a product is a stream: every row of A is read once, every
// row of C written once, and op(B) (all terms, a few tens of KB) is the only thing reused.  So B lives in LDS
// for the whole workgroup (as bf16 hi/lo pairs [n][k] in the split form, fp32 [k][n] in the exact one), and a
// wave takes 16-row tiles of A straight from memory into matrix-instruction operand layout: lane (row c, k
// group g) loads the four consecutive k of its row as one 16-byte load per 16-deep step, several steps ahead of
// the matrix instructions that consume them, with no LDS staging and no workgroup barrier in the loop.
// (The 16x16x4 fp32 instruction wants lane group g to hold k = step, not four consecutive k; a sum over k does
// not care about the order, so instruction j of a step takes k = 4g + j from both operands.)
// x of the lane N places further on in its row of 16 lanes (rotation within the row)
template <int N>
__device__ __forceinline__ float row_ror(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + N, 0xf, 0xf, false));
}
struct PanelStep {
  const float *A;  // the term's A at this batch member
  int lda, K, k0, kofs;  // kofs: the step's first k row / column inside the LDS copy of op(B)
  int comp;              // 1: A holds delta records (kernels built with DZ), lda floats per row of records
};
constexpr int PANEL_STEPS = 48, PANEL_HEAD = 1536;  // bytes reserved in front of B for the step table
constexpr int PANEL_T = 512, PANEL_W = PANEL_T / 64;  // eight waves share one LDS copy of B
constexpr int PANEL_NMAX = 208, PANEL_LDS_MAX = 160 * 1024;  // result columns (thirteen 16-column tiles), LDS per workgroup

// AX: what term 0's A is: 0 a plain matrix, 1 delta records (GemmSumP::dz_coef), 2 a BatchNorm input normalised on
// load and written back (GemmSumP::ax_*), 3 delta records whose product never enters the matrix cores: a row's
// expansion is shift[k] everywhere plus one value per group of 16, so its product with op(B) is a constant row
// (shift . op(B), formed once per workgroup) plus one scaled row of op(B) per record, added to the tile in fp32 before
// the epilogue (K0 / 16 rows of op(B) per result row instead of K0), 4 the im2col rows of a 3x3 convolution gathered from
// the images on load and written out for later readers (GemmSumP::im_*; H a multiple of 16, so a 16-row tile is 16
// neighbouring pixels of one image row; at most two 16-deep steps)
// EPI: built with the epilogue GemmSumP::epi: 1 row softmax, 2 row softmax backward, 3 per-tile records of a softmax
// over the ROW axis, 4 BatchNorm partial records of the result's columns (operands swapped: a lane owns a column),
// 5 BatchNorm apply + ReLU + 16 x 16 max-pool instead of storing; WV: waves per workgroup (sixteen where a copy of op(B)
// leaves room for one workgroup per compute unit and the kernel fits four waves on a SIMD)
// GS: 16-deep steps per group (one group is in the matrix cores while the next group's rows of A are in flight; with
// nsteps <= GS the whole next tile is in flight during a tile)
template <int PR, int NT, int AX = 0, int EPI = 0, int WV = PANEL_W, int GS = 4>
__global__ __launch_bounds__(64 * WV, (AX == 3 || EPI == 4 || WV > 8 || (NT > 8 && EPI != 2)) ? 4 : 1) void k_gemm_panel(GemmSumP p, int rows_per_wg, int Kp, int nsteps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  PanelStep *steps = reinterpret_cast<PanelStep *>(smem);
  constexpr int NP16 = NT * 16;
  // rows of Kp + 8 halves: Kp is a multiple of 16, so a row is an odd multiple of four banks and the 8-byte operand
  // reads of the 32 lanes of a half wave (16 rows x 2 k groups) fall on 64 different banks
  const int ldk = Kp + 8, ldn = NP16 + 4;
  unsigned short *__restrict__ Bh = reinterpret_cast<unsigned short *>(smem + PANEL_HEAD);
  unsigned short *__restrict__ Bl = Bh + NP16 * ldk;
  float *__restrict__ Bs = reinterpret_cast<float *>(smem + PANEL_HEAD);
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  // step table and the LDS copy of op(B), term by term (constant term indices: the by-value argument stays in
  // scalar registers)
  int kofs = 0, sbase = 0;
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    if (ti < p.nterms && !(AX == 3 && ti == 0)) {
      const GemmTerm t = p.t[ti];
      const int ns = (t.K + 15) >> 4, K16 = ns * 16;
      if (tid >= sbase && tid < sbase + ns) {
        const int sidx = tid - sbase;
        steps[tid] = PanelStep{t.A + (long)b * t.sA, t.lda, t.K, sidx * 16, kofs + sidx * 16, (AX != 0 && ti == 0) ? 1 : 0};
      }
      const float *__restrict__ B = t.B + (long)b * t.sB;
      auto put = [&](int k, int n, float v) {
        if constexpr (PR == 2) {
          const unsigned short h = to_bf16(v);
          Bh[n * ldk + kofs + k] = h;
          Bl[n * ldk + kofs + k] = to_bf16(v - from_bf16(h));
        } else {
          Bs[(kofs + k) * ldn + n] = v;
        }
      };
      if (t.vecB) {  // 16-byte loads, eight per thread in flight before the first LDS store
        const int qrow = t.tb ? K16 / 4 : NP16 / 4, Q = K16 * NP16 / 4;
        for (int e0 = tid; e0 < Q; e0 += (64 * WV) * 8) {
          float4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * (64 * WV), major = e / qrow, minor = 4 * (e - major * qrow);
            const int k = t.tb ? minor : major, n = t.tb ? major : minor;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < Q && k < t.K && n < p.N)
              v[u] = *reinterpret_cast<const float4 *>(B + (t.tb ? (long)n * t.ldb + k : (long)k * t.ldb + n));
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * (64 * WV), major = e / qrow, minor = 4 * (e - major * qrow);
            if (e >= Q) continue;
            const float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            if (t.tb) {  // four consecutive k of row n
              if constexpr (PR == 2) {
                unsigned short h[4], l[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  h[q] = to_bf16(x[q]);
                  l[q] = to_bf16(x[q] - from_bf16(h[q]));
                }
                *reinterpret_cast<uint2 *>(&Bh[major * ldk + kofs + minor]) =
                    make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
                *reinterpret_cast<uint2 *>(&Bl[major * ldk + kofs + minor]) =
                    make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) Bs[(kofs + minor + q) * ldn + major] = x[q];
              }
            } else {  // four consecutive n of row k
              if constexpr (PR == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) put(major, minor + q, x[q]);
              } else {
                *reinterpret_cast<float4 *>(&Bs[(kofs + major) * ldn + minor]) = v[u];
              }
            }
          }
        }
      } else {
        for (int e0 = tid; e0 < K16 * NP16; e0 += (64 * WV) * 4) {
          float v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * (64 * WV);
            const int k = t.tb ? e % K16 : e / NP16, n = t.tb ? e / K16 : e % NP16;
            v[u] = (e < K16 * NP16 && k < t.K && n < p.N) ? B[t.tb ? (long)n * t.ldb + k : (long)k * t.ldb + n] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * (64 * WV);
            if (e < K16 * NP16) put(t.tb ? e % K16 : e / NP16, t.tb ? e / K16 : e % NP16, v[u]);
          }
        }
      }
      kofs += K16;
      sbase += ns;
    }
  }
  // DZ: the expansion coefficients of term 0 (scale [K], shift [K]) behind the copy of op(B)
  float *__restrict__ zc = reinterpret_cast<float *>(smem + PANEL_HEAD) +
                           (PR == 2 ? NP16 * ldk : Kp * ldn);  // floats: the hi / lo halves together are NP16 * ldk floats
  if constexpr (AX == 1) {
    const int K0 = p.t[0].K;
    for (int i = tid; i < 2 * K0; i += (64 * WV)) zc[i] = p.dz_coef[i];
  }
  if constexpr (AX == 3) {  // scale [K0], shift . op(B) [NP16], op(B) of term 0 in fp32 [K0][NP16 + 4]
    const GemmTerm t = p.t[0];
    if (p.nterms == 1) {  // no other term: one step of zeros keeps the tile loop's shape (Kp = 16)
      if (tid == 0) steps[0] = PanelStep{t.A, 0, 0, 0, 0, 0};
      for (int i = tid; i < 2 * NP16 * ldk; i += (64 * WV)) Bh[i] = 0;
    }
    const float *__restrict__ B = t.B + (long)b * t.sB;
    float *__restrict__ sB = zc + t.K + NP16;
    for (int i = tid; i < t.K; i += (64 * WV)) zc[i] = p.dz_coef[i];
    for (int e = tid; e < t.K * NP16; e += (64 * WV)) {
      const int k = t.tb ? e % t.K : e / NP16, n = t.tb ? e / t.K : e % NP16;
      sB[k * (NP16 + 4) + n] = n < p.N ? B[t.tb ? (long)n * t.ldb + k : (long)k * t.ldb + n] : 0.f;
    }
    __syncthreads();
    if (tid < NP16) {
      float a = 0.f;
      for (int k = 0; k < t.K; ++k) a = fmaf(p.dz_coef[t.K + k], sB[k * (NP16 + 4) + tid], a);
      zc[t.K + tid] = a;
    }
  }
  if constexpr (AX == 2) {  // mean [K0p], 1/sigma [K0p], gamma [K0p], beta [K0p] (K0p = K padded to 16: zeros)
    const int K0 = p.t[0].K, K0p = ((K0 + 15) >> 4) * 16;
    for (int i = tid; i < 4 * K0p; i += (64 * WV)) {
      const int w = i / K0p, k = i - w * K0p;
      zc[i] = k >= K0 ? 0.f : w == 0 ? p.ax_stat[k] : w == 1 ? p.ax_stat[K0 + k] : w == 2 ? p.ax_gamma[k] : p.ax_beta[k];
    }
  }
  // EPI 4: the pivot and the bias [NP16 each] behind op(B)
  float *__restrict__ cst = reinterpret_cast<float *>(smem + PANEL_HEAD) + (PR == 2 ? NP16 * ldk : Kp * ldn);
  // EPI 5: mean, 1/sigma * gamma, beta [NP16 each], then the segment records of one chunk of 16 * WV rows:
  // best value, its column inside the segment, the value of C there [16 * WV][NT] each
  float *__restrict__ pl_c = cst;
  float *__restrict__ seg_y = cst + 3 * NP16;
  int *__restrict__ seg_k = reinterpret_cast<int *>(seg_y + 16 * WV * NT);
  float *__restrict__ seg_z = seg_y + 2 * 16 * WV * NT;
  if constexpr (EPI == 5) {
    for (int i = tid; i < NP16; i += (64 * WV)) {
      const bool on = i < p.N;
      pl_c[i] = on ? p.pl_stat[i] : 0.f;
      pl_c[NP16 + i] = on ? p.pl_stat[p.N + i] * p.pl_gamma[i] : 0.f;
      pl_c[2 * NP16 + i] = on ? p.pl_beta[i] : 0.f;
    }
  }
  if constexpr (EPI == 4) {  // pivot [NP16], bias [NP16]; the waves' sums take op(B)'s place once the rows are done
    for (int i = tid; i < NP16; i += (64 * WV)) {
      cst[i] = (p.col_pivot && i < p.N) ? p.col_pivot[i] : 0.f;
      cst[NP16 + i] = (p.bias && i < p.N) ? p.bias[i] : 0.f;
    }
  }
  __syncthreads();

  const int G = (nsteps + GS - 1) / GS;  // groups of GS steps: one group is in the matrix cores, the next in flight
  const long wg_row0 = (long)blockIdx.x * rows_per_wg;
  const int ntiles = rows_per_wg / (16 * WV);  // 16-row tiles per wave: tile i of this wave is tile wave + WV i of the workgroup
  auto tile_row0 = [&](int i) { return wg_row0 + (long)(wave + WV * i) * 16; };
  // AX 4: the lane's eight taps (two steps x four consecutive k): offset from the pixel inside its image, and the
  // tap's bit in the pixel's mask of the nine window positions that lie inside the image (bit 9, never set, for the
  // zero columns past 9 Ci); the workgroup's first row as (image, pixel inside it)
  int tap_off[AX == 4 ? 2 : 1][4], tap_bit[AX == 4 ? 2 : 1][4];
  int im_img0 = 0, im_rem0 = 0;
  unsigned im_hmagic = 0;
  if constexpr (AX == 4) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 16 * e + 4 * g + q, ci = k / 9, r9 = k - 9 * ci, kh = r9 / 3, kw = r9 - 3 * kh;
        tap_off[e][q] = ci * p.im_H * p.im_H + (kh - 1) * p.im_H + (kw - 1);
        tap_bit[e][q] = k < 9 * p.im_Ci ? r9 : 9;
      }
    const long w0 = (long)blockIdx.x * rows_per_wg;
    im_img0 = (int)(w0 / (p.im_H * p.im_H));
    im_rem0 = (int)(w0 - (long)im_img0 * p.im_H * p.im_H);
    im_hmagic = (unsigned)((1ull << 32) / (unsigned)p.im_H) + 1u;  // n / H = umulhi(n, magic) for n < H * H, H <= 1024
  }
  auto fetch = [&](long row0, int gi, float4 (&dst)[GS]) {
    const long row = row0 + c;
    if constexpr (AX == 4) {
      // the tile's 16 rows are 16 neighbouring pixels of one image row (H % 16 == 0): image and image row are the
      // wave's, computed once per tile from the workgroup's first row without a division
      const int H = p.im_H, P = H * H;
      int rem = __builtin_amdgcn_readfirstlane(im_rem0 + (int)(row0 - wg_row0)), img = im_img0;
      while (rem >= P) {
        rem -= P;
        ++img;
      }
      const int h = (int)__umulhi((unsigned)rem, im_hmagic), w = rem - h * H + c;
      const float *__restrict__ ximg = p.im_X + (long)img * p.im_Ci * P;
      const int pix = h * H + w;
      // bit 3 kh + kw: the window position lies inside the image (rows past M: none)
      const unsigned wm = (w >= 1 ? 1u : 0u) | 2u | (w + 1 < H ? 4u : 0u);
      const unsigned m9 = row < p.M ? ((h >= 1 ? wm : 0u) | (wm << 3) | (h + 1 < H ? wm << 6 : 0u)) : 0u;
#pragma unroll
      for (int e = 0; e < GS; ++e) {
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (e < 2 && e < nsteps) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool ok = (m9 >> tap_bit[e < 2 ? e : 0][q]) & 1u;
            x[q] = ok ? ximg[pix + tap_off[e < 2 ? e : 0][q]] : 0.f;
          }
        }
        dst[e] = make_float4(x[0], x[1], x[2], x[3]);
      }
      return;
    }
#pragma unroll
    for (int e = 0; e < GS; ++e) {
      const int si = gi * GS + e;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (si < nsteps) {
        const PanelStep st = steps[si];
        const int k = st.k0 + 4 * g;
        if (AX == 1 && st.comp) {  // the record of (row, group st.k0 / 16): the four lanes of a row read the same pair
          v.y = __int_as_float(-1);
          if (row < p.M) {
            const float2 rc = *reinterpret_cast<const float2 *>(st.A + row * st.lda + 2 * (st.k0 >> 4));
            v.x = rc.x;
            v.y = rc.y;
          }
        } else if (row < p.M && k < st.K) v = *reinterpret_cast<const float4 *>(st.A + row * st.lda + k);
      }
      dst[e] = v;
    }
  };
  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // EPI 6: the lane's running sums of (value - pivot) and of its square, one column per tile
  float cs1[EPI == 4 ? NT : 1], cs2[EPI == 4 ? NT : 1];
#pragma unroll
  for (int j = 0; j < (EPI == 4 ? NT : 1); ++j) cs1[j] = cs2[j] = 0.f;
  // AX 3: the records of the lane's row (pairs of value and position, two to a 16-byte load; at most eight)
  float4 rq[AX == 3 ? 4 : 1], rq_n[AX == 3 ? 4 : 1];
  auto fetch_records = [&](long row0, float4 (&dst)[AX == 3 ? 4 : 1]) {
    if constexpr (AX == 3) {
      const long row = row0 + c;
      const float *__restrict__ R = p.t[0].A + (long)b * p.t[0].sA + row * p.t[0].lda;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        dst[u] = make_float4(0.f, __int_as_float(-1), 0.f, __int_as_float(-1));
        if (row < p.M && 32 * u < p.t[0].K) dst[u] = *reinterpret_cast<const float4 *>(R + 4 * u);
      }
    }
  };
  fetch_records(tile_row0(0), rq);
  float4 cur[GS], nxt[GS];
  fetch(tile_row0(0), 0, cur);
  float *__restrict__ C = p.C + (long)b * p.sC;
  // kernels of more than eight tiles with an epilogue are only launched on 16-byte rows (C, epi_a) with N % 4 == 0:
  // no element-wise paths in them
  constexpr bool QUAD = NT > 8 && EPI != 0;
  const bool vecC = QUAD || ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && p.ldc % 4 == 0 && p.sC % 4 == 0);
  for (int i = 0; i < ntiles; ++i) {
    const long row0 = tile_row0(i);
    for (int gi = 0; gi < G; ++gi) {
      const bool last = gi + 1 == G;
      if (!last) fetch(row0, gi + 1, nxt);
      else if (i + 1 < ntiles) {
        fetch_records(tile_row0(i + 1), rq_n);
        fetch(tile_row0(i + 1), 0, nxt);
      }
#pragma unroll
      for (int e = 0; e < GS; ++e) {
        const int si = gi * GS + e;
        if (si < nsteps) {
          const int kb = steps[si].kofs + 4 * g;
          float av[4] = {cur[e].x, cur[e].y, cur[e].z, cur[e].w};
          if constexpr (AX == 2) {
            if (steps[si].comp) {
              const PanelStep st = steps[si];
              const int k = st.k0 + 4 * g, K0p = ((st.K + 15) >> 4) * 16;
              const float4 mu = *reinterpret_cast<const float4 *>(&zc[k]);
              const float4 iv = *reinterpret_cast<const float4 *>(&zc[K0p + k]);
              const float4 ga = *reinterpret_cast<const float4 *>(&zc[2 * K0p + k]);
              const float4 be = *reinterpret_cast<const float4 *>(&zc[3 * K0p + k]);
              av[0] = (av[0] - mu.x) * iv.x * ga.x + be.x;
              av[1] = (av[1] - mu.y) * iv.y * ga.y + be.y;
              av[2] = (av[2] - mu.z) * iv.z * ga.z + be.z;
              av[3] = (av[3] - mu.w) * iv.w * ga.w + be.w;
              const long row = row0 + c;
              if (row < p.M && k < st.K)  // K % 4 == 0 (the operand is 16-byte loadable)
                *reinterpret_cast<float4 *>(p.ax_out + (st.A - p.t[0].A) + row * st.lda + k) = make_float4(av[0], av[1], av[2], av[3]);
            }
          }
          if constexpr (AX == 4) {  // the im2col rows for later readers (the convolution's weight gradient)
            const long row = row0 + c;
            const int k = steps[si].k0 + 4 * g;
            if (row < p.M && k < steps[si].K)
              *reinterpret_cast<float4 *>(p.ax_out + row * steps[si].lda + k) = make_float4(av[0], av[1], av[2], av[3]);
          }
          if constexpr (AX == 1) {
            if (steps[si].comp) {
              const int k = steps[si].k0 + 4 * g, pos = __float_as_int(cur[e].y) - 4 * g;
              const float4 ca = *reinterpret_cast<const float4 *>(&zc[k]);
              const float4 cb = *reinterpret_cast<const float4 *>(&zc[steps[si].K + k]);
              const float val = cur[e].x;
              av[0] = fmaf(pos == 0 ? val : 0.f, ca.x, cb.x);
              av[1] = fmaf(pos == 1 ? val : 0.f, ca.y, cb.y);
              av[2] = fmaf(pos == 2 ? val : 0.f, ca.z, cb.z);
              av[3] = fmaf(pos == 3 ? val : 0.f, ca.w, cb.w);
            }
          }
          if constexpr (PR == 2) {
            unsigned h01, l01, h23, l23;
            split2(av[0], av[1], h01, l01);
            split2(av[2], av[3], h23, l23);
            const s16x4 ah = __builtin_bit_cast(s16x4, u32x2{h01, h23}), al = __builtin_bit_cast(s16x4, u32x2{l01, l23});
            // op(B)^T . A^T: the result tile comes out transposed, a lane holding four adjacent columns of
            // one row of C (one 16-byte store).  All of op(B)'s fragments first, then the three partial products
            // tile by tile: consecutive matrix instructions write different accumulators, none waits for the one
            // before it
            constexpr int JG = (GS > 4 || (EPI == 4 && NT == 8)) ? 4 : NT <= 8 ? NT : (EPI == 1 ? 4 : 7);  // tiles whose fragments are held at once
#pragma unroll
            for (int j0 = 0; j0 < NT; j0 += JG) {
              s16x4 bh[JG], bl[JG];
#pragma unroll
              for (int j = 0; j < JG; ++j)
                if (j0 + j < NT) {
                  bh[j] = *reinterpret_cast<const s16x4 *>(&Bh[((j0 + j) * 16 + c) * ldk + kb]);
                  bl[j] = *reinterpret_cast<const s16x4 *>(&Bl[((j0 + j) * 16 + c) * ldk + kb]);
                }
#pragma unroll
              for (int j = 0; j < JG; ++j)
                if (j0 + j < NT)
                  acc[j0 + j] = EPI == 4 ? __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh[j], acc[j0 + j], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bh[j], al, acc[j0 + j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < JG; ++j)
                if (j0 + j < NT)
                  acc[j0 + j] = EPI == 4 ? __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl[j], acc[j0 + j], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bl[j], ah, acc[j0 + j], 0, 0, 0);
#pragma unroll
              for (int j = 0; j < JG; ++j)
                if (j0 + j < NT)
                  acc[j0 + j] = EPI == 4 ? __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh[j], acc[j0 + j], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bh[j], ah, acc[j0 + j], 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int j = 0; j < NT; ++j)
                acc[j] = EPI == 4 ? __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], Bs[(kb + q) * ldn + j * 16 + c], acc[j], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_16x16x4f32(Bs[(kb + q) * ldn + j * 16 + c], av[q], acc[j], 0, 0, 0);
          }
        }
      }
      if (last) {
        const long gm = row0 + c;  // lane (c, g) holds C[row c][16 j + 4 g .. + 3]
        const bool row_on = gm < p.M;
        if constexpr (AX == 3) {
          const int K0 = p.t[0].K;
          const float *__restrict__ sB = zc + K0 + NP16;
          f32x4 sp[NT];
#pragma unroll
          for (int j = 0; j < NT; ++j) sp[j] = *reinterpret_cast<const f32x4 *>(zc + K0 + j * 16 + 4 * g);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float rv[4] = {rq[u].x, rq[u].y, rq[u].z, rq[u].w};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              if (16 * (2 * u + h) < K0) {
                const int pos = __float_as_int(rv[2 * h + 1]);
                const bool on = (unsigned)pos < 16u;
                const int k = 16 * (2 * u + h) + (on ? pos : 0);
                const float w = on ? rv[2 * h] * zc[k] : 0.f;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                  const f32x4 bq = *reinterpret_cast<const f32x4 *>(sB + k * (NP16 + 4) + j * 16 + 4 * g);
#pragma unroll
                  for (int v = 0; v < 4; ++v) sp[j][v] = fmaf(w, bq[v], sp[j][v]);
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);  // two records' rows of op(B) in registers at a time, not all eight's
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] += sp[j];
#pragma unroll
          for (int u = 0; u < 4; ++u) rq[u] = rq_n[u];
        }
        if constexpr (EPI == 4) {
          // operands the other way round: lane (c, g) holds C[rows 4 g .. + 3][column 16 j + c], so a column's sums over
          // the rows of all of the wave's tiles gather in registers and cross lanes once, after the last tile
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int gn = j * 16 + c;
            const float pv = cst[gn], bj = cst[NP16 + gn];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const long gr = row0 + 4 * g + v;
              const float x = acc[j][v] * p.alpha + bj;
              const float d = gr < p.M ? x - pv : 0.f;
              cs1[j] += d;
              cs2[j] += d * d;
              const bool st = p.c_first_row_only ? (b == 0 && gr == 0) : gr < p.M;
              if (st && gn < p.N) C[gr * p.ldc + gn] = x;
            }
            acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        } else if constexpr (EPI != 0) {
          // row epilogues: a row's values sit in the four lanes c, c+16, c+32, c+48, NT * 4 in each
          float r[NT][4];
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int gn = j * 16 + 4 * g;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              r[j][v] = acc[j][v] * p.alpha;
              if (p.bias && (QUAD ? gn : gn + v) < p.N) r[j][v] += p.bias[gn + v];
            }
            acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
          if constexpr (EPI == 1) {
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                if (j * 16 + 4 * g + (QUAD ? 0 : v) >= p.N) r[j][v] = -INFINITY;  // columns past N: exp(-inf - mx) = 0 below
                mx = fmaxf(mx, r[j][v]);
              }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                r[j][v] = NT > 8 ? __expf(r[j][v] - mx) : expf(r[j][v] - mx);
                sum += r[j][v];
              }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
              for (int v = 0; v < 4; ++v) r[j][v] /= sum;
          } else if constexpr (EPI == 5) {
            // y = relu((value - mean) * scale + beta); per row and 16-column segment the first maximum of y: the lane's
            // four columns first, then the four lanes of the row (columns ascend with the lane group)
            const int rl = (wave) * 16 + c;  // row inside the chunk
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const int gn = j * 16 + 4 * g;
              const float4 mu = *reinterpret_cast<const float4 *>(pl_c + gn);
              const float4 sc = *reinterpret_cast<const float4 *>(pl_c + NP16 + gn);
              const float4 be = *reinterpret_cast<const float4 *>(pl_c + 2 * NP16 + gn);
              const float mq[4] = {mu.x, mu.y, mu.z, mu.w}, sq[4] = {sc.x, sc.y, sc.z, sc.w}, bq[4] = {be.x, be.y, be.z, be.w};
              float best = -INFINITY, bz = 0.f;
              int bk = 0;
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                const float y = fmaxf(fmaf(r[j][v] - mq[v], sq[v], bq[v]), 0.f);
                if (y > best) {
                  best = y;
                  bk = 4 * g + v;
                  bz = r[j][v];
                }
              }
#pragma unroll
              for (int o = 16; o < 64; o <<= 1) {
                const float oy = __shfl_xor(best, o), oz = __shfl_xor(bz, o);
                const int ok = __shfl_xor(bk, o);
                if (oy > best || (oy == best && ok < bk)) {
                  best = oy;
                  bk = ok;
                  bz = oz;
                }
              }
              if (g == 0) {
                seg_y[rl * NT + j] = best;
                seg_k[rl * NT + j] = bk;
                seg_z[rl * NT + j] = bz;
              }
            }
          } else if constexpr (EPI == 3) {
            // column records of this tile: the 16 rows of a column sit in the 16 lanes c of one lane group g
            float *__restrict__ rec = p.tile_rec + ((long)b * ((p.M + 15) >> 4) + (row0 >> 4)) * 2 * p.N;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const int gn = j * 16 + 4 * g;
              float mx[4], se[4];
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                const float x = row_on ? r[j][v] : -INFINITY;
                // all-reduce over the 16 lanes of a row by rotations (data-parallel-primitive operands: no LDS trip)
                float m = x;
                m = fmaxf(m, row_ror<8>(m));
                m = fmaxf(m, row_ror<4>(m));
                m = fmaxf(m, row_ror<2>(m));
                m = fmaxf(m, row_ror<1>(m));
                float e = row_on ? __expf(x - m) : 0.f;  // row 0 of a tile inside the matrix exists: m is finite there
                e += row_ror<8>(e);
                e += row_ror<4>(e);
                e += row_ror<2>(e);
                e += row_ror<1>(e);
                mx[v] = m;
                se[v] = e;
              }
              if (c == 0 && gn < p.N && row0 < p.M) {  // (a workgroup's last tiles may lie past the matrix)
                if (QUAD || gn + 3 < p.N) {
                  *reinterpret_cast<float4 *>(rec + gn) = make_float4(mx[0], mx[1], mx[2], mx[3]);
                  *reinterpret_cast<float4 *>(rec + p.N + gn) = make_float4(se[0], se[1], se[2], se[3]);
                } else {
#pragma unroll
                  for (int v = 0; v < 4; ++v)
                    if (gn + v < p.N) {
                      rec[gn + v] = mx[v];
                      rec[p.N + gn + v] = se[v];
                    }
                }
              }
            }
          } else {
            const float *__restrict__ arow = p.epi_a + (long)b * p.sC + gm * p.ldc;
            const bool vecE = QUAD || (vecC && (reinterpret_cast<uintptr_t>(p.epi_a) & 15) == 0);
            float a[NT][4], dot = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const int gn = j * 16 + 4 * g;
              if constexpr (QUAD) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row_on && gn < p.N) t = *reinterpret_cast<const float4 *>(arow + gn);
                a[j][0] = t.x; a[j][1] = t.y; a[j][2] = t.z; a[j][3] = t.w;
              } else if (vecE && row_on && gn + 3 < p.N) {
                const float4 t = *reinterpret_cast<const float4 *>(arow + gn);
                a[j][0] = t.x; a[j][1] = t.y; a[j][2] = t.z; a[j][3] = t.w;
              } else {
#pragma unroll
                for (int v = 0; v < 4; ++v) a[j][v] = (row_on && gn + v < p.N) ? arow[gn + v] : 0.f;
              }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
              for (int v = 0; v < 4; ++v) dot += a[j][v] * r[j][v];
            dot += __shfl_xor(dot, 16);
            dot += __shfl_xor(dot, 32);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
              for (int v = 0; v < 4; ++v) r[j][v] = a[j][v] * (r[j][v] - dot);
          }
          const bool store_row = EPI == 5 ? false : row_on;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int gn = j * 16 + 4 * g;
            if (store_row && gn < p.N) {
              float *__restrict__ dst = C + gm * p.ldc + gn;
              if (QUAD || (vecC && gn + 3 < p.N)) {
                *reinterpret_cast<float4 *>(dst) = make_float4(r[j][0], r[j][1], r[j][2], r[j][3]);
              } else {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                  if (gn + v < p.N) dst[v] = r[j][v];
              }
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int gn = j * 16 + 4 * g;
            float r[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              r[v] = acc[j][v] * p.alpha;
              if (p.bias && gn + v < p.N) r[v] += p.bias[gn + v];
            }
            if (row_on && gn < p.N) {
              float *__restrict__ dst = C + gm * p.ldc + gn;
              if (vecC && gn + 3 < p.N) {
                if (p.accumulate) {
                  const float4 o = *reinterpret_cast<const float4 *>(dst);
                  r[0] += o.x; r[1] += o.y; r[2] += o.z; r[3] += o.w;
                }
                *reinterpret_cast<float4 *>(dst) = make_float4(r[0], r[1], r[2], r[3]);
              } else {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                  if (gn + v < p.N) dst[v] = p.accumulate ? r[v] + dst[v] : r[v];
              }
            }
            acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        }
      }
#pragma unroll
      for (int e = 0; e < GS; ++e) cur[e] = nxt[e];
    }
    if constexpr (EPI == 5) {
      // the chunk's 16 * WV rows are a whole number of bands (pk image rows of one plane): one thread per window walks
      // the window's pk rows, one segment record each, in row order
      __syncthreads();
      const int W = p.pl_H, pk = p.pl_pk, band = pk * W, P = W * W, wpb = W / pk;  // wpb = 14 windows per band
      const long F0 = (wg_row0 + (long)i * 16 * WV) * p.N;                            // the chunk's flat start inside the image
      const int nwin = (16 * WV * p.N / band) * wpb;
      for (int t = tid; t < nwin; t += 64 * WV) {
        const int bnd = t / wpb, wi = t - bnd * wpb;
        const long Fb = F0 + (long)bnd * band;
        const int plane = (int)(Fb / P), ii = (int)((Fb - (long)plane * P) / band);
        float best = -INFINITY, bz = 0.f;
        int bki = 0, bkj = 0;
        for (int ki = 0; ki < pk; ++ki) {
          const int f = bnd * band + ki * W + wi * pk;  // flat offset inside the chunk: a multiple of 16
          const int rl = f / p.N, sj = (f - rl * p.N) >> 4;
          const float y = seg_y[rl * NT + sj];
          if (y > best) {
            best = y;
            bki = ki;
            bkj = seg_k[rl * NT + sj];
            bz = seg_z[rl * NT + sj];
          }
        }
        const long win = ((long)b * p.N + plane) * (wpb * wpb) + ii * wpb + wi;
        p.pl_pooled[win] = best;
        p.pl_idx[win] = (ii * pk + bki) * W + wi * pk + bkj;
        p.pl_cmax[win] = bz;
      }
      __syncthreads();
    }
  }
  if constexpr (EPI == 4) {
    // the lane groups' sums of a column, then the waves' in wave order (their slots take op(B)'s place), brought from
    // the pivot to (sum, squared distances from the workgroup's own mean)
    __syncthreads();
    float *__restrict__ slot = reinterpret_cast<float *>(smem + PANEL_HEAD);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float a = cs1[j], q = cs2[j];
      a += __shfl_xor(a, 16); q += __shfl_xor(q, 16);
      a += __shfl_xor(a, 32); q += __shfl_xor(q, 32);
      if (g == 0) {
        slot[wave * 2 * NP16 + j * 16 + c] = a;
        slot[wave * 2 * NP16 + NP16 + j * 16 + c] = q;
      }
    }
    const float pvt = tid < NP16 ? cst[tid] : 0.f;
    __syncthreads();
    if (tid < p.N) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < WV; ++w) {
        s1 += slot[w * 2 * NP16 + tid];
        s2 += slot[w * 2 * NP16 + NP16 + tid];
      }
      const long left = p.M - wg_row0;
      const double n = (double)(left < rows_per_wg ? (left > 0 ? left : 0) : rows_per_wg);
      const double pv = (double)pvt;
      float *__restrict__ o = p.col_part + ((long)b * gridDim.x + blockIdx.x) * 2 * p.N;
      o[tid] = (float)((double)s1 + n * pv);
      o[p.N + tid] = n > 0.0 ? (float)((double)s2 - (double)s1 * (double)s1 / n) : 0.f;
    }
  }
}

// C[b][m][n] = sum over slices of part[s][b][m][n] (+ bias) (+ C), slices in ascending order.
__global__ __launch_bounds__(256) void k_gemm_splitk_reduce(GemmP p, int batch) {
  const long per = (long)p.M * p.N, total = per * batch;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int b = (int)(i / per);
    const long r = i - (long)b * per;
    const int m = (int)(r / p.N), n = (int)(r - (long)m * p.N);
    float s = 0.f;
    for (int k = 0; k < p.ksplit; ++k) s += p.part[((long)k * batch + b) * per + r];
    if (p.bias) s += p.bias[n];
    float *c = p.C + (long)b * p.sC + (long)m * p.ldc + n;
    if (p.accumulate) s += *c;
    *c = s;
  }
}

}  // namespace vt

using namespace vt;

// One instantiation of the row-panel kernel.  More than 64 KB of LDS per workgroup (a single workgroup may hold all of a
// compute unit's 160 KB) is asked for once per device and instantiation.
template <int PR, int NT, int AX, int EPI = 0, int WV = PANEL_W, int GS = 4>
static hipError_t panel_go(dim3 grid, size_t lds, hipStream_t hs, const GemmSumP &p, int rows, int Kp, int nsteps) {
  if (lds > 64 * 1024) {
    static std::atomic<unsigned long long> asked{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 64 || !((asked.load(std::memory_order_relaxed) >> dev) & 1)) {
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_panel<PR, NT, AX, EPI, WV, GS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PANEL_LDS_MAX);
      if (e != hipSuccess) return e;
      if (dev < 64) asked.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
  }
  hipLaunchKernelGGL((k_gemm_panel<PR, NT, AX, EPI, WV, GS>), grid, dim3(64 * WV), lds, hs, p, rows, Kp, nsteps);
  return hipGetLastError();
}

// Row-panel launch of a sum of products when it is tall and narrow and every A is 16-byte loadable; returns
// 1 when it took the job, 0 when the tiled kernels should, < 0 on a launch error.
static int panel_launch(int form, const GemmSumP &p, int batch, hipStream_t hs, bool dz = false, bool ax = false,
                        int *rows_out = nullptr) {
  static const bool off = [] {
    const char *e = getenv("VT_GEMM_PANEL");
    return e && *e == '0';
  }();
  static const int rows_forced = [] {
    const char *e = getenv("VT_PANEL_ROWS");
    const int v = e ? atoi(e) : 0;
    return (v >= 128 && v <= 8192 && (v & (v - 1)) == 0) ? v : 0;
  }();
  static const bool wide16 = [] {
    const char *e = getenv("VT_PANEL_WAVES16");
    return !(e && *e == '0');
  }();
  if (off || (form != 0 && form != 2) || p.N > PANEL_NMAX || p.M < 2048 || batch > 65535) return 0;
  if (p.epi != 0 && p.accumulate) return 0;
  const int NT = p.N <= 16 ? 1 : p.N <= 32 ? 2 : p.N <= 64 ? 4 : p.N <= 128 ? 8 : 13;
  static const bool sparse_on = [] {
    const char *e = getenv("VT_PANEL_RECORD_ROWS");
    return !(e && *e == '0');
  }();
  // records applied row by row (16-byte loads of a row's records: at most eight, in pairs)
  const bool sparse = dz && sparse_on && form == 2 && (NT == 1 || NT == 4) && p.t[0].K % 32 == 0 && p.t[0].K <= 128 &&
                      (reinterpret_cast<uintptr_t>(p.t[0].A) & 15) == 0 && p.t[0].sA % 4 == 0 && p.t[0].lda % 4 == 0;
  int Kp = 0, nsteps = 0;
  if (sparse && p.nterms == 1) Kp = 16, nsteps = 1;  // a step of zeros
  for (int i = sparse ? 1 : 0; i < p.nterms; ++i) {
    if (!p.t[i].vecA) return 0;
    const int ns = (p.t[i].K + 15) / 16;
    nsteps += ns;
    Kp += ns * 16;
  }
  if (nsteps > PANEL_STEPS) return 0;
  size_t lds = PANEL_HEAD + (form == 2 ? (size_t)2 * NT * 16 * (Kp + 8) * 2 : (size_t)Kp * (NT * 16 + 4) * 4);
  if (dz) {
    if (form != 2 || p.t[0].K % 16 != 0 || !p.dz_coef || (NT != 1 && NT != 4 && NT != 13)) return 0;
    if (sparse) lds += ((size_t)p.t[0].K + NT * 16 + (size_t)p.t[0].K * (NT * 16 + 4)) * sizeof(float);
    else lds += (size_t)2 * p.t[0].K * sizeof(float);
  }
  if (ax) {  // built for the exact form with one or thirteen result tiles (the tokenizer's scores)
    if (form != 0 || (NT != 1 && NT != 13) || !p.ax_stat || !p.ax_gamma || !p.ax_beta || !p.ax_out || p.t[0].K % 4 != 0) return 0;
    lds += (size_t)4 * ((p.t[0].K + 15) / 16) * 16 * sizeof(float);
  }
  if (lds > (size_t)PANEL_LDS_MAX) return 0;
  // too few rows to fill the device with 128-row workgroups that each pay for a large copy of op(B): the tiled kernels
  // (plain sums only: the record and epilogue forms have no tiled counterpart of one launch)
  if (lds > 64 * 1024 && !dz && p.epi == 0 && (long)((p.M + 127) / 128) * batch < 512) return 0;
  // rows per workgroup: as many as leave a few thousand workgroups (the LDS copy of B is paid once per workgroup);
  // a copy of more than half a compute unit's LDS leaves one workgroup per unit, which then takes at least 512 rows
  int rows = 1024;
  const int rows_min = lds > 80 * 1024 ? 512 : 128;
  while (rows > rows_min && (long)((p.M + rows - 1) / rows) * batch < 2048) rows >>= 1;
  if (rows_forced) rows = rows_forced;
  const dim3 grid((unsigned)((p.M + rows - 1) / rows), (unsigned)batch);
  hipError_t e = hipSuccess;
#define VT_PANEL(PR_, NT_, AX_) e = panel_go<PR_, NT_, AX_>(grid, lds, hs, p, rows, Kp, nsteps)
#define VT_PANEL_NT(PR_)                    \
  do {                                      \
    if (NT == 1) VT_PANEL(PR_, 1, 0);       \
    else if (NT == 2) VT_PANEL(PR_, 2, 0);  \
    else if (NT == 4) VT_PANEL(PR_, 4, 0);  \
    else if (NT == 8) VT_PANEL(PR_, 8, 0);  \
    else VT_PANEL(PR_, 13, 0);              \
  } while (0)
  if (p.epi == 5) {  // normalise, ReLU, max-pool instead of storing: seven-wave workgroups, chunks of 112 rows
    const int H = p.pl_H, pk = p.pl_pk;
    if (NT != 8 || form != 2 || ax || dz || p.accumulate || !p.pl_stat || !p.pl_gamma || !p.pl_beta || !p.pl_pooled ||
        !p.pl_idx || !p.pl_cmax)
      return 0;
    if (pk != 16 || H % 16 != 0 || H / pk != 14 || (long)H * H != p.M || p.N % 16 != 0 || p.M % 112 != 0 ||
        (112 * p.N) % (pk * H) != 0)
      return 0;
    const int chunks = p.M / 112;
    int k = 16;  // chunks per workgroup: op(B) is copied once per workgroup
    while (k > 1 && (chunks % k != 0 || (long)(chunks / k) * batch < 1024)) --k;
    lds += (size_t)(3 * NT * 16 + 3 * 16 * 7 * NT) * sizeof(float);
    if (lds > (size_t)PANEL_LDS_MAX) return 0;
    if (nsteps == 5) e = panel_go<2, 8, 0, 5, 7, 5>(dim3((unsigned)(chunks / k), (unsigned)batch), lds, hs, p, 112 * k, Kp, nsteps);
    else e = panel_go<2, 8, 0, 5, 7>(dim3((unsigned)(chunks / k), (unsigned)batch), lds, hs, p, 112 * k, Kp, nsteps);
  } else if (p.epi == 4) {  // BatchNorm partial records of the result's columns: built for the split form with eight tiles and
                     // the exact form with four
    if (!((NT == 8 && form == 2) || (NT == 4 && form == 0)) || ax || dz || p.accumulate || !p.col_part) return 0;
    if (batch > 1 && p.M % rows != 0) return 0;  // every workgroup but the launch's last must hold `rows` rows
    if ((long)batch * grid.x * 2 * p.N > p.col_part_floats) return 0;
    const bool sixteen = lds > 80 * 1024 && wide16;  // decided as for the plain launch
    lds += (size_t)2 * NT * 16 * sizeof(float);  // pivot and bias; the waves' sums reuse op(B)'s space
    if (lds > (size_t)PANEL_LDS_MAX) return 0;
    if (p.im_X) {  // im2col rows gathered from the images
      if (NT != 4 || form != 0 || p.nterms != 1 || nsteps > 2 || batch != 1 || p.im_H < 16 || p.im_H % 16 != 0 || p.im_H > 1024 ||
          p.M % (p.im_H * p.im_H) != 0 || p.im_Ci * 9 > p.t[0].K || !p.ax_out || p.t[0].lda % 4 != 0 ||
          (reinterpret_cast<uintptr_t>(p.ax_out) & 15) != 0)
        return 0;
      e = panel_go<0, 4, 4, 4>(grid, lds, hs, p, rows, Kp, nsteps);
    } else if (NT == 4) e = panel_go<0, 4, 0, 4>(grid, lds, hs, p, rows, Kp, nsteps);
    else if (sixteen) e = panel_go<2, 8, 0, 4, 16>(grid, lds, hs, p, rows, Kp, nsteps);
    else e = panel_go<2, 8, 0, 4>(grid, lds, hs, p, rows, Kp, nsteps);
    if (e == hipSuccess && rows_out) *rows_out = rows;
  } else if (p.epi == 3) {  // tile records of the softmax over rows: built for the normalise-on-load form
    if ((NT != 1 && NT != 13) || !ax || dz || !p.tile_rec || p.N % 4 != 0 || p.ldc % 4 != 0 || p.sC % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(p.C) & 15) != 0 || (reinterpret_cast<uintptr_t>(p.tile_rec) & 15) != 0)
      return 0;
    if (NT == 1) e = panel_go<0, 1, 2, 3>(grid, lds, hs, p, rows, Kp, nsteps);
    else e = panel_go<0, 13, 2, 3>(grid, lds, hs, p, rows, Kp, nsteps);
  } else if (p.epi != 0) {  // the row epilogues are built for one and for thirteen result tiles, without the on-load forms but records
    if ((NT != 1 && NT != 13) || ax || (dz && p.epi != 2)) return 0;
    if (NT == 13 && (p.N % 4 != 0 || p.ldc % 4 != 0 || p.sC % 4 != 0 || (reinterpret_cast<uintptr_t>(p.C) & 15) != 0 ||
                     (p.epi == 2 && (reinterpret_cast<uintptr_t>(p.epi_a) & 15) != 0)))
      return 0;  // the wide-row epilogues move whole 16-byte groups
#define VT_PANEL_EPI(PR_, NT_, AX_)                                                   \
  do {                                                                                \
    if (p.epi == 1) e = panel_go<PR_, NT_, AX_, 1>(grid, lds, hs, p, rows, Kp, nsteps); \
    else e = panel_go<PR_, NT_, AX_, 2>(grid, lds, hs, p, rows, Kp, nsteps);            \
  } while (0)
    if (sparse && NT == 1) e = panel_go<2, 1, 3, 2>(grid, lds, hs, p, rows, Kp, nsteps);
    else if (dz && NT == 1) e = panel_go<2, 1, 1, 2>(grid, lds, hs, p, rows, Kp, nsteps);
    else if (dz) e = panel_go<2, 13, 1, 2>(grid, lds, hs, p, rows, Kp, nsteps);
    else if (form == 2 && NT == 1) VT_PANEL_EPI(2, 1, 0);
    else if (form == 2) VT_PANEL_EPI(2, 13, 0);
    else if (NT == 1) VT_PANEL_EPI(0, 1, 0);
    else VT_PANEL_EPI(0, 13, 0);
#undef VT_PANEL_EPI
  } else if (ax && NT == 1) VT_PANEL(0, 1, 2);
  else if (ax) VT_PANEL(0, 13, 2);
  else if (sparse && NT == 1) VT_PANEL(2, 1, 3);
  else if (sparse) VT_PANEL(2, 4, 3);
  else if (dz && NT == 1) VT_PANEL(2, 1, 1);
  else if (dz && NT == 4) VT_PANEL(2, 4, 1);
  else if (dz) VT_PANEL(2, 13, 1);
  else if (form == 2 && NT == 8 && lds > 80 * 1024 && wide16) e = panel_go<2, 8, 0, false, 16>(grid, lds, hs, p, rows, Kp, nsteps);
  else if (form == 2) VT_PANEL_NT(2);
  else VT_PANEL_NT(0);
#undef VT_PANEL_NT
#undef VT_PANEL
  return e == hipSuccess ? 1 : -(int)e;
}

static int gemm_launch(int form, int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda,
                       long strideA, const float *B, int ldb, long strideB, int accumulate, float *C, int ldc, long strideC,
                       const float *bias, int batch, float *workspace, long workspace_floats, vt_stream s) {
  if (M < 1 || N < 1 || K < 1 || batch < 1 || !A || !B || !C) return VT_EINVAL;
  if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return VT_EINVAL;
  GemmP p{A, B, C, bias, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, alpha, accumulate ? 1 : 0, 1, K, batch, 0, 0, nullptr, nullptr};
  // 16-byte loads need the contiguous extent and the leading dimension in multiples of 4 floats, the
  // batch strides too, and a 16-byte aligned base; K slices are multiples of 16, so with K % 4 == 0 no
  // vector straddles a slice end
  auto aligned = [](const float *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  p.vecA = aligned(A) && lda % 4 == 0 && strideA % 4 == 0 && (transA ? M % 4 == 0 : K % 4 == 0);
  p.vecB = aligned(B) && ldb % 4 == 0 && strideB % 4 == 0 && (transB ? K % 4 == 0 : N % 4 == 0);
  // tile shape: 64x128 when N is wide (A is then read once per 128 columns); 128x128 when the product
  // is also deep enough to be bound by the matrix pipe rather than by streaming its operands
  int BM = 64, BN = 64;
  if (N > 64) {
    BN = 128;
    if (K >= 128 && M >= 128 && (long)((M + 127) / 128) * ((N + 127) / 128) * batch >= 512) BM = 128;
  }
  const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * batch;
  // a contraction that is long for the few output tiles it feeds: slice K so that several hundred
  // workgroups exist (a lone workgroup walking 1024 rows costs 50 us of pure latency)
  // (a very long one, e.g. a per-image contraction over 50176 pixels, is sliced up to twice as many tiles: 256
  // workgroups walking 3136 K steps each leave three quarters of the chip's wave slots empty)
  const long target = K >= 4096 ? 2048 : 1024;
  if (workspace && K >= 256 && tiles * 4 <= target) {
    long want = (target + tiles - 1) / tiles;
    const long most = K >= 4096 ? (K + 511) / 512 : K / 64;
    want = want < most ? want : most;
    const long fit = workspace_floats / ((long)M * N * batch);
    if (want > fit) want = fit;
    if (want > 1) {
      p.ksplit = (int)want;
      p.kchunk = (int)(((K + want - 1) / want + GBK - 1) / GBK * GBK);
      p.ksplit = (K + p.kchunk - 1) / p.kchunk;
      p.part = workspace;
    }
  }
  if (!transA && p.ksplit == 1) {  // tall and narrow: the row-panel kernel
    GemmSumP q{};
    q.t[0] = GemmTerm{A, B, strideA, strideB, K, lda, ldb, transB ? 1 : 0, p.vecA, p.vecB};
    q.nterms = 1; q.C = C; q.bias = bias; q.M = M; q.N = N; q.ldc = ldc; q.sC = strideC; q.alpha = alpha;
    q.accumulate = p.accumulate;
    if (const int took = panel_launch(form, q, batch, (hipStream_t)s)) return took > 0 ? 0 : -took;
  }
  const long gz = (long)batch * p.ksplit;
  if (gz > 65535) return VT_EUNSUPPORTED;
  if ((N + BN - 1) / BN > 65535) return VT_EUNSUPPORTED;
  const dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, (unsigned)gz);
  if ((long)grid.x * grid.y * grid.z >= (1L << 32)) return VT_EUNSUPPORTED;  // tile_of_block counts workgroups in 32 bits
  hipStream_t hs = (hipStream_t)s;
#define VT_GEMM_LAUNCH(TA_, TB_, BF_)                                                                                  \
  do {                                                                                                                 \
    if (BM == 128) hipLaunchKernelGGL((k_gemm<TA_, TB_, BF_, 128, 128>), grid, dim3(256), 0, hs, p);                 \
    else if (BN == 128) hipLaunchKernelGGL((k_gemm<TA_, TB_, BF_, 64, 128>), grid, dim3(256), 0, hs, p);              \
    else hipLaunchKernelGGL((k_gemm<TA_, TB_, BF_, 64, 64>), grid, dim3(256), 0, hs, p);                             \
  } while (0)
#define VT_GEMM_FORM(PR_)                                      \
  do {                                                         \
    if (transA && transB) VT_GEMM_LAUNCH(true, true, PR_);     \
    else if (transA) VT_GEMM_LAUNCH(true, false, PR_);         \
    else if (transB) VT_GEMM_LAUNCH(false, true, PR_);         \
    else VT_GEMM_LAUNCH(false, false, PR_);                    \
  } while (0)
  if (form == 2) VT_GEMM_FORM(2);
  else if (form == 1) VT_GEMM_FORM(1);
  else VT_GEMM_FORM(0);
#undef VT_GEMM_FORM
#undef VT_GEMM_LAUNCH
  VT_CHECK_LAUNCH();
  if (p.ksplit > 1 && !bias && !accumulate && ldc == N && (batch == 1 || strideC == (long)M * N) &&
      (long)M * N * batch <= (1L << 19)) {
    // plain case (the weight gradients): the partial tiles are rows of a [slices, batch*M*N] matrix
    // (results of more than half a million elements go to the element-wise pass below: the slab reduction is laid out
    // for many rows of few columns and reads a few wide rows at a fifth of that pass's rate)
    return vt_reduce_rows(workspace, p.ksplit, (int)((long)M * N * batch), (int)((long)M * N * batch), C, s);
  }
  if (p.ksplit > 1) {
    const long total = (long)M * N * batch;
    const int nb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3(nb), dim3(256), 0, hs, p, batch);
    VT_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int vt_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, long strideA,
                           const float *B, int ldb, long strideB, int accumulate, float *C, int ldc, long strideC,
                           const float *bias, int batch, float *workspace, long workspace_floats, vt_stream s) {
  return gemm_launch(0, transA, transB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, accumulate, C, ldc, strideC, bias,
                     batch, workspace, workspace_floats, s);
}

extern "C" int vt_gemm_bf16_mfma(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda,
                                 long strideA, const float *B, int ldb, long strideB, int accumulate, float *C, int ldc,
                                 long strideC, const float *bias, int batch, float *workspace, long workspace_floats,
                                 vt_stream s) {
  return gemm_launch(1, transA, transB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, accumulate, C, ldc, strideC, bias,
                     batch, workspace, workspace_floats, s);
}

extern "C" int vt_gemm_f32x3(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, long strideA,
                             const float *B, int ldb, long strideB, int accumulate, float *C, int ldc, long strideC,
                             const float *bias, int batch, float *workspace, long workspace_floats, vt_stream s) {
  return gemm_launch(2, transA, transB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, accumulate, C, ldc, strideC, bias,
                     batch, workspace, workspace_floats, s);
}

// col_part != nullptr: the launch also leaves the BatchNorm partial records of C's columns (row-panel kernel only)
struct PoolArgs {
  const float *stat, *gamma, *beta;
  float *pooled, *cmax;
  int *idx;
  int H, pk;
};
static int gemm_sum_launch(int form, int nterms, const vt_gemm_term *terms, int M, int N, float alpha, int accumulate,
                           float *C, int ldc, long strideC, const float *bias, int batch, const float *col_pivot, float *col_part,
                           long col_part_floats, int *rows_per_block, vt_stream s, const PoolArgs *pool = nullptr) {
  const bool first_row_only = col_part && ldc == 0;  // statistics without the matrix: only row 0 of member 0 is stored
  if (form < 0 || form > 2 || nterms < 1 || nterms > 4 || !terms || M < 1 || N < 1 || batch < 1 || (!C && !pool) ||
      (ldc < N && !first_row_only && !pool))
    return VT_EINVAL;
  GemmSumP p{};
  auto aligned = [](const float *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  long ktot = 0;
  for (int i = 0; i < nterms; ++i) {
    const vt_gemm_term &t = terms[i];
    if (!t.A || !t.B || t.K < 1 || t.lda < t.K || t.ldb < (t.transB ? t.K : N)) return VT_EINVAL;
    GemmTerm &d = p.t[i];
    d.A = t.A; d.B = t.B; d.sA = t.strideA; d.sB = t.strideB; d.K = t.K; d.lda = t.lda; d.ldb = t.ldb; d.tb = t.transB ? 1 : 0;
    d.vecA = aligned(t.A) && t.lda % 4 == 0 && t.strideA % 4 == 0 && t.K % 4 == 0;
    d.vecB = aligned(t.B) && t.ldb % 4 == 0 && t.strideB % 4 == 0 && (t.transB ? t.K % 4 == 0 : N % 4 == 0);
    ktot += t.K;
  }
  p.nterms = nterms; p.C = C; p.bias = bias; p.M = M; p.N = N; p.ldc = ldc; p.sC = strideC; p.alpha = alpha;
  p.accumulate = accumulate ? 1 : 0;
  if (pool) {
    p.epi = 5;
    p.pl_stat = pool->stat; p.pl_gamma = pool->gamma; p.pl_beta = pool->beta;
    p.pl_pooled = pool->pooled; p.pl_cmax = pool->cmax; p.pl_idx = pool->idx; p.pl_H = pool->H; p.pl_pk = pool->pk;
    const int took = panel_launch(form, p, batch, (hipStream_t)s);
    return took > 0 ? 0 : took < 0 ? -took : VT_EUNSUPPORTED;
  }
  if (col_part) {
    p.epi = 4;
    p.c_first_row_only = first_row_only ? 1 : 0;
    if (first_row_only) p.ldc = N;
    p.col_part = col_part;
    p.col_part_floats = col_part_floats;
    p.col_pivot = col_pivot;
    const int took = panel_launch(form, p, batch, (hipStream_t)s, false, false, rows_per_block);
    return took > 0 ? 0 : took < 0 ? -took : VT_EUNSUPPORTED;
  }
  if (const int took = panel_launch(form, p, batch, (hipStream_t)s)) return took > 0 ? 0 : -took;
  int BM = 64, BN = 64;
  if (N > 64) {
    BN = 128;
    if (ktot >= 128 && M >= 128 && (long)((M + 127) / 128) * ((N + 127) / 128) * batch >= 512) BM = 128;
  } else if (N > 32 && ktot >= 256 && (long)((M + 127) / 128) * batch >= 2048) {
    BM = 128;  // a deep sum into a narrow result: 128 rows per workgroup halve the barriers and the reads of op(B) per row
  }
  if (batch > 65535 || (N + BN - 1) / BN > 65535) return VT_EUNSUPPORTED;
  const dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, (unsigned)batch);
  if ((long)grid.x * grid.y * grid.z >= (1L << 32)) return VT_EUNSUPPORTED;
  hipStream_t hs = (hipStream_t)s;
#define VT_SUM_LAUNCH(PR_)                                                                               \
  do {                                                                                                   \
    if (BM == 128 && BN == 64) hipLaunchKernelGGL((k_gemm_sum<PR_, 128, 64>), grid, dim3(256), 0, hs, p); \
    else if (BM == 128) hipLaunchKernelGGL((k_gemm_sum<PR_, 128, 128>), grid, dim3(256), 0, hs, p);      \
    else if (BN == 128) hipLaunchKernelGGL((k_gemm_sum<PR_, 64, 128>), grid, dim3(256), 0, hs, p);       \
    else hipLaunchKernelGGL((k_gemm_sum<PR_, 64, 64>), grid, dim3(256), 0, hs, p);                       \
  } while (0)
  if (form == 2) VT_SUM_LAUNCH(2);
  else if (form == 1) VT_SUM_LAUNCH(1);
  else VT_SUM_LAUNCH(0);
#undef VT_SUM_LAUNCH
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_gemm_sum(int form, int nterms, const vt_gemm_term *terms, int M, int N, float alpha, int accumulate,
                           float *C, int ldc, long strideC, const float *bias, int batch, vt_stream s) {
  return gemm_sum_launch(form, nterms, terms, M, N, alpha, accumulate, C, ldc, strideC, bias, batch, nullptr, nullptr, 0, nullptr, s);
}

extern "C" int vt_gemm_sum_colstats(int form, int nterms, const vt_gemm_term *terms, int M, int N, float alpha, float *C,
                                    int ldc, long strideC, const float *bias, int batch, const float *col_pivot,
                                    float *col_part, long col_part_floats, int *rows_per_block, vt_stream s) {
  if (!col_part || !rows_per_block) return VT_EINVAL;
  return gemm_sum_launch(form, nterms, terms, M, N, alpha, 0, C, ldc, strideC, bias, batch, col_pivot, col_part,
                         col_part_floats, rows_per_block, s);
}

extern "C" int vt_conv3x3_rows_colstats(const float *X, int images, int Ci, int H, const float *Wp, int ld, int Co,
                                       const float *bias, float *col, float *C, const float *col_pivot, float *col_part,
                                       long col_part_floats, int *rows_per_block, vt_stream s) {
  if (!X || !Wp || !col || !C || !col_part || !rows_per_block || images < 1 || Ci < 1 || H < 1 || Co < 1 || ld < 9 * Ci)
    return VT_EINVAL;
  const long M = (long)images * H * H;
  if (M >= (1L << 31) || ld % 4 != 0 || (reinterpret_cast<uintptr_t>(Wp) & 15) != 0) return VT_EUNSUPPORTED;
  GemmSumP p{};
  GemmTerm &d = p.t[0];
  d.A = col; d.B = Wp; d.sA = 0; d.sB = 0; d.K = ld; d.lda = ld; d.ldb = ld; d.tb = 1; d.vecA = 1; d.vecB = 1;
  p.nterms = 1; p.C = C; p.bias = bias; p.M = (int)M; p.N = Co; p.ldc = Co; p.sC = 0; p.alpha = 1.f;
  p.epi = 4; p.col_part = col_part; p.col_part_floats = col_part_floats; p.col_pivot = col_pivot;
  p.im_X = X; p.im_Ci = Ci; p.im_H = H; p.ax_out = col;
  const int took = panel_launch(0, p, 1, (hipStream_t)s, false, false, rows_per_block);
  return took > 0 ? 0 : took < 0 ? -took : VT_EUNSUPPORTED;
}

extern "C" int vt_gemm_sum_bn_relu_pool(int form, int nterms, const vt_gemm_term *terms, int H, int N, const float *bias,
                                       int batch, const float *stat, const float *gamma, const float *beta, int pk,
                                       float *pooled, int *pool_idx, float *cmax, vt_stream s) {
  if (H < 1 || (long)H * H > (1L << 30) || !stat || !gamma || !beta || !pooled || !pool_idx || !cmax) return VT_EINVAL;
  const PoolArgs pa{stat, gamma, beta, pooled, cmax, pool_idx, H, pk};
  return gemm_sum_launch(form, nterms, terms, H * H, N, 1.f, 0, nullptr, N, 0, bias, batch, nullptr, nullptr, 0, nullptr, s, &pa);
}

static int bn_apply_launch(int M, int N, int K, const float *A, int lda, const float *stat, const float *gamma,
                           const float *beta, float *A_out, const float *B, int ldb, int transB, const float *bias, float *C,
                           int ldc, float *tile_rec, vt_stream s) {
  if (M < 1 || N < 1 || K < 1 || !A || !stat || !gamma || !beta || !A_out || !B || !C || lda < K || ldc < N ||
      ldb < (transB ? K : N))
    return VT_EINVAL;
  if (N > 16 && (N <= 128 || N > PANEL_NMAX)) return VT_EUNSUPPORTED;  // kernels are built for one and for thirteen tiles
  auto aligned = [](const float *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  GemmSumP p{};
  p.t[0] = GemmTerm{A, B, 0, 0, K, lda, ldb, transB ? 1 : 0, aligned(A) && aligned(A_out) && lda % 4 == 0 && K % 4 == 0,
                    aligned(B) && ldb % 4 == 0 && (transB ? K % 4 == 0 : N % 4 == 0)};
  p.nterms = 1; p.C = C; p.bias = bias; p.M = M; p.N = N; p.ldc = ldc; p.sC = 0; p.alpha = 1.f;
  p.ax_stat = stat; p.ax_gamma = gamma; p.ax_beta = beta; p.ax_out = A_out;
  if (tile_rec) {
    p.epi = 3;
    p.tile_rec = tile_rec;
  }
  const int took = panel_launch(0, p, 1, (hipStream_t)s, false, true);
  if (took < 0) return -took;
  return took > 0 ? 0 : VT_EUNSUPPORTED;
}

extern "C" int vt_gemm_bn_apply(int M, int N, int K, const float *A, int lda, const float *stat, const float *gamma,
                                const float *beta, float *A_out, const float *B, int ldb, int transB, const float *bias, float *C,
                                int ldc, vt_stream s) {
  return bn_apply_launch(M, N, K, A, lda, stat, gamma, beta, A_out, B, ldb, transB, bias, C, ldc, nullptr, s);
}

extern "C" int vt_gemm_bn_apply_tile_records(int M, int N, int K, const float *A, int lda, const float *stat, const float *gamma,
                                             const float *beta, float *A_out, const float *B, int ldb, int transB,
                                             const float *bias, float *C, int ldc, float *tile_rec, vt_stream s) {
  if (!tile_rec) return VT_EINVAL;
  return bn_apply_launch(M, N, K, A, lda, stat, gamma, beta, A_out, B, ldb, transB, bias, C, ldc, tile_rec, s);
}

// ---- transposed-A product with at most 16 result rows and B given as delta records: C[b] = A[b]^T . expand(rec[b]),
// A [K, M <= 16].  With so few result rows a 64-row tile wastes three quarters of its matrix instructions and every K
// step of 16 pays two barriers; here a wave takes 16 rows of K at a time straight from memory into operand layout
// (A: lane (m, row group) loads its four rows' element m; B: the lane's four rows' records, expanded), keeps the whole
// [16, N] result in registers over its share of K, and the waves' results are added in a fixed order: through LDS
// inside the workgroup, through part[slice][batch][M * N] and k_gemm_splitk_reduce between the K slices.  No LDS or
// barrier in the loop.  NT = N / 16 result tiles.
template <int NT>
__global__ __launch_bounds__(256) void k_tn16_records(GemmP p, int tiles_per_slice) {
  __shared__ float red[3][16 * NT * 16];
  const int b = blockIdx.y, slice = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
  const float *__restrict__ A = p.A + (long)b * p.sA;
  const float *__restrict__ R = p.B + (long)b * p.sB;
  float ca[NT], cb[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    ca[j] = p.bz_coef[16 * j + c];
    cb[j] = p.bz_coef[p.N + 16 * j + c];
  }
  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int t0 = slice * tiles_per_slice, t1 = min(t0 + tiles_per_slice, (p.K + 15) >> 4);
  for (int t = t0 + wave; t < t1; t += 4) {
    const int r0 = t * 16 + 4 * g;  // this lane's four rows of K
    float a[4];
    float2 rc[NT][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = (r0 + q < p.K && c < p.M) ? A[(long)(r0 + q) * p.lda + c] : 0.f;
    // a row's records are adjacent: two per 16-byte load (the launcher checked the alignment; NT is even)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < NT; j += 2) {
        const float4 v = r0 + q < p.K ? *reinterpret_cast<const float4 *>(R + (long)(r0 + q) * p.ldb + 2 * j)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
        rc[j][q] = make_float2(v.x, v.y);
        rc[j + 1][q] = make_float2(v.z, v.w);
      }
    unsigned h01, l01, h23, l23;
    split2(a[0], a[1], h01, l01);
    split2(a[2], a[3], h23, l23);
    const s16x4 ah = __builtin_bit_cast(s16x4, u32x2{h01, h23}), al = __builtin_bit_cast(s16x4, u32x2{l01, l23});
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float d = __float_as_int(rc[j][q].y) == c ? rc[j][q].x : 0.f;
        bv[q] = r0 + q < p.K ? fmaf(d, ca[j], cb[j]) : 0.f;  // rows past the end contribute nothing (not the shift)
      }
      unsigned bh01, bl01, bh23, bl23;
      split2(bv[0], bv[1], bh01, bl01);
      split2(bv[2], bv[3], bh23, bl23);
      const s16x4 bh = __builtin_bit_cast(s16x4, u32x2{bh01, bh23}), bl = __builtin_bit_cast(s16x4, u32x2{bl01, bl23});
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc[j], 0, 0, 0);
    }
  }
  // lane (n = c, g) holds D[4 g + v][16 j + c]; waves 1..3 hand theirs to wave 0 through LDS
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave - 1][(4 * g + v) * (NT * 16) + 16 * j + c] = acc[j][v];
  }
  __syncthreads();
  if (wave == 0) {
    float *__restrict__ dst = p.part + ((long)slice * p.batch + b) * (long)p.M * p.N;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int m = 4 * g + v, idx = m * (NT * 16) + 16 * j + c;
        const float r = ((acc[j][v] + red[0][idx]) + red[1][idx]) + red[2][idx];
        if (m < p.M) dst[(long)m * p.N + 16 * j + c] = r;
      }
  }
}

// ---- products with the projector's output delta given as records (vt_pool_delta_records) ----------------------
extern "C" int vt_gemm_tn_records(int M, int N, int K, const float *A, int lda, long strideA, const float *rec, long strideRec,
                                  const float *coef, float *C, int ldc, long strideC, int batch, float *workspace,
                                  long workspace_floats, vt_stream s) {
  if (M < 1 || N < 16 || N % 16 != 0 || K < 1 || batch < 1 || !A || !rec || !coef || !C || lda < M || ldc < N) return VT_EINVAL;
  if (N <= 64) return VT_EUNSUPPORTED;  // built for the 64 x 128 tile only
  const int ldr = N / 8;                // floats per row of records
  GemmP p{A, rec, C, nullptr, M, N, K, lda, ldr, ldc, strideA, strideRec, strideC, 1.f, 0, 1, K, batch, 0, 0, nullptr, coef};
  auto aligned = [](const float *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  p.vecA = aligned(A) && lda % 4 == 0 && strideA % 4 == 0 && M % 4 == 0;
  if ((reinterpret_cast<uintptr_t>(rec) & 7) != 0 || strideRec % 2 != 0) return VT_EINVAL;
  static const bool skinny_off = [] {
    const char *e = getenv("VT_GEMM_TN16");
    return e && *e == '0';
  }();
  if (!skinny_off && M <= 16 && N == 128 && workspace && K >= 1024 && aligned(rec) && strideRec % 4 == 0) {
    // at most 16 result rows: the streaming kernel, K cut into slices that leave several thousand workgroups and fit the
    // workspace
    const long tiles = (K + 15) / 16, per = (long)M * N * batch;
    long slices = (8192 + batch - 1) / batch;  // (a workgroup is one wave per SIMD: several rounds of resident workgroups, so that the last round's share is small)
    if (slices > tiles / 8) slices = tiles / 8 > 0 ? tiles / 8 : 1;
    if (slices > workspace_floats / per) slices = workspace_floats / per;
    if (slices >= 1 && slices <= 65535 && batch <= 65535) {
      const int tps = (int)((tiles + slices - 1) / slices);
      p.ksplit = (int)((tiles + tps - 1) / tps);
      p.part = workspace;
      hipLaunchKernelGGL(k_tn16_records<8>, dim3(p.ksplit, batch), dim3(256), 0, (hipStream_t)s, p, tps);
      VT_CHECK_LAUNCH();
      const long total = per;
      const int nb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
      hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3(nb), dim3(256), 0, (hipStream_t)s, p, batch);
      VT_CHECK_LAUNCH();
      return 0;
    }
  }
  const int BM = 64, BN = 128;
  const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * batch;
  const long target = K >= 4096 ? 2048 : 1024;  // K slices as in gemm_launch
  if (workspace && K >= 256 && tiles * 4 <= target) {
    long want = (target + tiles - 1) / tiles;
    const long most = K >= 4096 ? (K + 511) / 512 : K / 64;
    want = want < most ? want : most;
    const long fit = workspace_floats / ((long)M * N * batch);
    if (want > fit) want = fit;
    if (want > 1) {
      p.ksplit = (int)want;
      p.kchunk = (int)(((K + want - 1) / want + GBK - 1) / GBK * GBK);
      p.ksplit = (K + p.kchunk - 1) / p.kchunk;
      p.part = workspace;
    }
  }
  const long gz = (long)batch * p.ksplit;
  if (gz > 65535) return VT_EUNSUPPORTED;
  const dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, (unsigned)gz);
  hipLaunchKernelGGL((k_gemm<true, false, 2, 64, 128, true>), grid, dim3(256), 0, (hipStream_t)s, p);
  VT_CHECK_LAUNCH();
  if (p.ksplit > 1 && ldc == N && (batch == 1 || strideC == (long)M * N) && (long)M * N * batch < (1L << 30))
    return vt_reduce_rows(workspace, p.ksplit, (int)((long)M * N * batch), (int)((long)M * N * batch), C, s);
  if (p.ksplit > 1) {
    const long total = (long)M * N * batch;
    const int nb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3(nb), dim3(256), 0, (hipStream_t)s, p, batch);
    VT_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int vt_gemm_sum_records(int nterms, const vt_gemm_term *terms, const float *coef, int epi, const float *epi_a, int M,
                                   int N, float *C, int ldc, long strideC, const float *bias, int batch, vt_stream s) {
  if (nterms < 1 || nterms > 4 || !terms || !coef || M < 1 || N < 1 || batch < 1 || !C || ldc < N) return VT_EINVAL;
  if ((epi != 0 && epi != 2) || (epi == 2 && (!epi_a || ldc != N))) return VT_EINVAL;
  const int K0 = terms[0].K;
  if (K0 % 16 != 0 || terms[0].lda != K0 / 8 || terms[0].strideA % 2 != 0 ||
      (reinterpret_cast<uintptr_t>(terms[0].A) & 7) != 0)
    return VT_EINVAL;
  GemmSumP p{};
  vt_gemm_term t0 = terms[0];
  t0.lda = K0;  // passes the dense-operand checks; the kernel reads records with the stride set below
  vt_gemm_term tt[4];
  for (int i = 0; i < nterms; ++i) tt[i] = i == 0 ? t0 : terms[i];
  auto aligned = [](const float *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  for (int i = 0; i < nterms; ++i) {
    const vt_gemm_term &t = tt[i];
    if (!t.A || !t.B || t.K < 1 || t.lda < t.K || t.ldb < (t.transB ? t.K : N)) return VT_EINVAL;
    GemmTerm &d = p.t[i];
    d.A = t.A; d.B = t.B; d.sA = t.strideA; d.sB = t.strideB; d.K = t.K; d.lda = t.lda; d.ldb = t.ldb; d.tb = t.transB ? 1 : 0;
    d.vecA = i == 0 ? 1 : (aligned(t.A) && t.lda % 4 == 0 && t.strideA % 4 == 0 && t.K % 4 == 0);
    d.vecB = aligned(t.B) && t.ldb % 4 == 0 && t.strideB % 4 == 0 && (t.transB ? t.K % 4 == 0 : N % 4 == 0);
  }
  p.t[0].lda = K0 / 8;
  p.nterms = nterms; p.C = C; p.bias = bias; p.M = M; p.N = N; p.ldc = ldc; p.sC = strideC; p.alpha = 1.f;
  p.epi = epi; p.epi_a = epi_a; p.dz_coef = coef;
  const int took = panel_launch(2, p, batch, (hipStream_t)s, true);
  if (took < 0) return -took;
  return took > 0 ? 0 : VT_EUNSUPPORTED;
}

extern "C" int vt_gemm_softmax_rows(int form, int epi, const float *a, int transB, int M, int N, int K, float alpha,
                                    const float *A, int lda, long strideA, const float *B, int ldb, long strideB, float *C,
                                    long strideC, int batch, vt_stream s) {
  if ((form != 0 && form != 2) || (epi != 1 && epi != 2) || M < 1 || N < 1 || K < 1 || batch < 1 || !A || !B || !C ||
      (epi == 2 && !a))
    return VT_EINVAL;
  if (lda < K || ldb < (transB ? K : N)) return VT_EINVAL;
  auto aligned = [](const float *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  GemmSumP q{};
  q.t[0] = GemmTerm{A, B, strideA, strideB, K, lda, ldb, transB ? 1 : 0,
                    aligned(A) && lda % 4 == 0 && strideA % 4 == 0 && K % 4 == 0,
                    aligned(B) && ldb % 4 == 0 && strideB % 4 == 0 && (transB ? K % 4 == 0 : N % 4 == 0)};
  q.nterms = 1; q.C = C; q.M = M; q.N = N; q.ldc = N; q.sC = strideC; q.alpha = alpha; q.epi = epi; q.epi_a = a;
  if (const int took = panel_launch(form, q, batch, (hipStream_t)s)) return took > 0 ? 0 : -took;
  // shapes the row-panel kernel does not take with an epilogue: the product, then the softmax kernel over its rows
  if (batch > 1 && strideC != (long)M * N) return VT_EUNSUPPORTED;
  const int rc = gemm_launch(form, 0, transB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, 0, C, N, strideC, nullptr, batch,
                             nullptr, 0, s);
  if (rc != 0) return rc;
  return epi == 1 ? vt_softmax_rows(C, (long)M * batch, N, s) : vt_softmax_bwd_rows(a, C, (long)M * batch, N, s);
}
