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
#include <hip/hip_runtime.h>

#include "../../include/vtcnn_hip.h"
#include "vtcnn_common.h"

namespace vt {

constexpr int GBM = 64, GBN = 64, GBK = 16, GPAD = 4;

struct GemmP {
  const float *A, *B;
  float *C;
  const float *bias;
  int M, N, K, lda, ldb, ldc;
  long sA, sB, sC;
  float alpha;
  int accumulate, ksplit, kchunk, batch, vecA, vecB;
  float *part;  // [ksplit][batch][M*N] when ksplit > 1
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One operand tile (64 rows of the M or N side x 16 of K) from global memory into four registers per
// thread.  `along_k`: the K index is the contiguous one in memory (A not transposed / B transposed).
// Vector form (one 16-byte load) when the host found the operand aligned and its sizes multiples of 4.
template <bool ALONG_K>
__device__ __forceinline__ void load_tile(const float *__restrict__ base, int ld, int r0, int rows, int k0, int kend,
                                          bool vec, int tid, float (&reg)[4]) {
  if (vec) {
    int r, k;
    if (ALONG_K) { r = tid >> 2; k = (tid & 3) * 4; } else { k = tid >> 4; r = (tid & 15) * 4; }
    const int gr = r0 + r, gk = k0 + k;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gr < rows && gk < kend)
      t = *reinterpret_cast<const float4 *>(base + (ALONG_K ? (long)gr * ld + gk : (long)gk * ld + gr));
    reg[0] = t.x; reg[1] = t.y; reg[2] = t.z; reg[3] = t.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      int r, k;
      if (ALONG_K) { k = idx & (GBK - 1); r = idx >> 4; } else { r = idx & 63; k = idx >> 6; }
      const int gr = r0 + r, gk = k0 + k;
      reg[e] = (gr < rows && gk < kend) ? base[ALONG_K ? (long)gr * ld + gk : (long)gk * ld + gr] : 0.f;
    }
  }
}
// registers -> LDS tile [k][row]
template <bool ALONG_K>
__device__ __forceinline__ void store_tile(float (*T)[GBM + GPAD], bool vec, int tid, const float (&reg)[4]) {
  if (vec) {
    if (ALONG_K) {
      const int r = tid >> 2, k = (tid & 3) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) T[k + e][r] = reg[e];
    } else {
      const int k = tid >> 4, r = (tid & 15) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) T[k][r + e] = reg[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      if (ALONG_K) T[idx & (GBK - 1)][idx >> 4] = reg[e]; else T[idx >> 6][idx & 63] = reg[e];
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
template <bool ALONG_K>
__device__ __forceinline__ void store_tile_bf16(unsigned short (*T)[GBK + HPAD], bool vec, int tid, const float (&reg)[4]) {
  if (vec) {
    if (ALONG_K) {
      const int r = tid >> 2, k = (tid & 3) * 4;
      const unsigned lo = to_bf16(reg[0]) | ((unsigned)to_bf16(reg[1]) << 16), hi = to_bf16(reg[2]) | ((unsigned)to_bf16(reg[3]) << 16);
      *reinterpret_cast<uint2 *>(&T[r][k]) = make_uint2(lo, hi);
    } else {
      const int k = tid >> 4, r = (tid & 15) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) T[r + e][k] = to_bf16(reg[e]);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      if (ALONG_K) T[idx >> 4][idx & (GBK - 1)] = to_bf16(reg[e]); else T[idx & 63][idx >> 6] = to_bf16(reg[e]);
    }
  }
}

template <bool TA, bool TB, bool BF>
__global__ __launch_bounds__(256) void k_gemm(GemmP p) {
  __shared__ float As[BF ? 1 : GBK][GBM + GPAD];
  __shared__ float Bs[BF ? 1 : GBK][GBN + GPAD];
  __shared__ __attribute__((aligned(8))) unsigned short Ah[BF ? GBM : 1][GBK + HPAD];
  __shared__ __attribute__((aligned(8))) unsigned short Bh[BF ? GBN : 1][GBK + HPAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z / p.ksplit, ks = blockIdx.z - b * p.ksplit;
  const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN;  // rows on x: the batch's pixel rows run into the millions
  const int kbeg = ks * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  const float *__restrict__ A = p.A + (long)b * p.sA;
  const float *__restrict__ B = p.B + (long)b * p.sB;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const bool vecA = p.vecA != 0, vecB = p.vecB != 0;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the next K step's tiles travel from memory while the matrix cores work on the current one
  float ra[4], rb[4];
  load_tile<!TA>(A, p.lda, m0, p.M, kbeg, kend, vecA, tid, ra);
  load_tile<TB>(B, p.ldb, n0, p.N, kbeg, kend, vecB, tid, rb);
  for (int k0 = kbeg; k0 < kend; k0 += GBK) {
    __syncthreads();  // the previous step's tile reads are over
    if constexpr (BF) {
      store_tile_bf16<!TA>(Ah, vecA, tid, ra);
      store_tile_bf16<TB>(Bh, vecB, tid, rb);
    } else {
      store_tile<!TA>(As, vecA, tid, ra);
      store_tile<TB>(Bs, vecB, tid, rb);
    }
    __syncthreads();
    if (k0 + GBK < kend) {
      load_tile<!TA>(A, p.lda, m0, p.M, k0 + GBK, kend, vecA, tid, ra);
      load_tile<TB>(B, p.ldb, n0, p.N, k0 + GBK, kend, vecB, tid, rb);
    }
    if constexpr (BF) {
      const int kq = 4 * (lane >> 4), c = lane & 15;
      const s16x4 a0 = *reinterpret_cast<const s16x4 *>(&Ah[wm + c][kq]), a1 = *reinterpret_cast<const s16x4 *>(&Ah[wm + 16 + c][kq]);
      const s16x4 b0 = *reinterpret_cast<const s16x4 *>(&Bh[wn + c][kq]), b1 = *reinterpret_cast<const s16x4 *>(&Bh[wn + 16 + c][kq]);
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1, b1, acc[1][1], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < GBK; kk += 4) {
        const int kr = kk + (lane >> 4), c = lane & 15;
        const float a0 = As[kr][wm + c], a1 = As[kr][wm + 16 + c];
        const float b0 = Bs[kr][wn + c], b1 = Bs[kr][wn + 16 + c];
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
  }
  const bool direct = p.ksplit == 1;
  float *__restrict__ C = direct ? p.C + (long)b * p.sC : p.part + ((long)ks * p.batch + b) * (long)p.M * p.N;
  const int ldc = direct ? p.ldc : p.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int gm = m0 + wm + i * 16 + 4 * (lane >> 4) + v, gn = n0 + wn + j * 16 + (lane & 15);
        if (gm < p.M && gn < p.N) {
          float r = acc[i][j][v] * p.alpha;
          if (direct) {
            if (p.bias) r += p.bias[gn];
            if (p.accumulate) r += C[(long)gm * ldc + gn];
          }
          C[(long)gm * ldc + gn] = r;
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

static int gemm_launch(bool bf16, int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda,
                       long strideA, const float *B, int ldb, long strideB, int accumulate, float *C, int ldc, long strideC,
                       const float *bias, int batch, float *workspace, long workspace_floats, vt_stream s) {
  if (M < 1 || N < 1 || K < 1 || batch < 1 || !A || !B || !C) return VT_EINVAL;
  if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return VT_EINVAL;
  GemmP p{A, B, C, bias, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, alpha, accumulate ? 1 : 0, 1, K, batch, 0, 0, nullptr};
  // 16-byte loads need the contiguous extent and the leading dimension in multiples of 4 floats, the
  // batch strides too, and a 16-byte aligned base; K slices are multiples of 16, so with K % 4 == 0 no
  // vector straddles a slice end
  auto aligned = [](const float *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  p.vecA = aligned(A) && lda % 4 == 0 && strideA % 4 == 0 && (transA ? M % 4 == 0 : K % 4 == 0);
  p.vecB = aligned(B) && ldb % 4 == 0 && strideB % 4 == 0 && (transB ? K % 4 == 0 : N % 4 == 0);
  const long tiles = (long)((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN) * batch;
  // a contraction that is long for the few output tiles it feeds: slice K so that several hundred
  // workgroups exist (a lone workgroup walking 1024 rows costs 50 us of pure latency)
  if (workspace && K >= 256 && tiles < 256) {
    long want = (1024 + tiles - 1) / tiles;
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
  if ((N + GBN - 1) / GBN > 65535) return VT_EUNSUPPORTED;
  const dim3 grid((M + GBM - 1) / GBM, (N + GBN - 1) / GBN, (unsigned)gz);
  hipStream_t hs = (hipStream_t)s;
  if (bf16) {
    if (transA && transB) hipLaunchKernelGGL((k_gemm<true, true, true>), grid, dim3(256), 0, hs, p);
    else if (transA) hipLaunchKernelGGL((k_gemm<true, false, true>), grid, dim3(256), 0, hs, p);
    else if (transB) hipLaunchKernelGGL((k_gemm<false, true, true>), grid, dim3(256), 0, hs, p);
    else hipLaunchKernelGGL((k_gemm<false, false, true>), grid, dim3(256), 0, hs, p);
  } else {
    if (transA && transB) hipLaunchKernelGGL((k_gemm<true, true, false>), grid, dim3(256), 0, hs, p);
    else if (transA) hipLaunchKernelGGL((k_gemm<true, false, false>), grid, dim3(256), 0, hs, p);
    else if (transB) hipLaunchKernelGGL((k_gemm<false, true, false>), grid, dim3(256), 0, hs, p);
    else hipLaunchKernelGGL((k_gemm<false, false, false>), grid, dim3(256), 0, hs, p);
  }
  VT_CHECK_LAUNCH();
  if (p.ksplit > 1 && !bias && !accumulate && ldc == N && (batch == 1 || strideC == (long)M * N) &&
      (long)M * N * batch < (1L << 30)) {
    // plain case (the weight gradients): the partial tiles are rows of a [slices, batch*M*N] matrix
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
  return gemm_launch(false, transA, transB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, accumulate, C, ldc, strideC, bias,
                     batch, workspace, workspace_floats, s);
}

extern "C" int vt_gemm_bf16_mfma(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda,
                                 long strideA, const float *B, int ldb, long strideB, int accumulate, float *C, int ldc,
                                 long strideC, const float *bias, int batch, float *workspace, long workspace_floats,
                                 vt_stream s) {
  return gemm_launch(true, transA, transB, M, N, K, alpha, A, lda, strideA, B, ldb, strideB, accumulate, C, ldc, strideC, bias,
                     batch, workspace, workspace_floats, s);
}
