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
  int accumulate, ksplit, kchunk, batch;
  float *part;  // [ksplit][batch][M*N] when ksplit > 1
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm(GemmP p) {
  __shared__ float As[GBK][GBM + GPAD];
  __shared__ float Bs[GBK][GBN + GPAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z / p.ksplit, ks = blockIdx.z - b * p.ksplit;
  const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN;  // rows on x: the batch's pixel rows run into the millions
  const int kbeg = ks * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  const float *__restrict__ A = p.A + (long)b * p.sA;
  const float *__restrict__ B = p.B + (long)b * p.sB;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k0 = kbeg; k0 < kend; k0 += GBK) {
    float ra[4], rb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      int m, k;
      if (TA) { m = idx & (GBM - 1); k = idx >> 6; } else { k = idx & (GBK - 1); m = idx >> 4; }
      const int gm = m0 + m, gk = k0 + k;
      const bool ok = gm < p.M && gk < kend;
      const long off = TA ? (long)gk * p.lda + gm : (long)gm * p.lda + gk;
      ra[e] = ok ? A[off] : 0.f;
      int n, kb;
      if (TB) { kb = idx & (GBK - 1); n = idx >> 4; } else { n = idx & (GBN - 1); kb = idx >> 6; }
      const int gn = n0 + n, gkb = k0 + kb;
      const bool okb = gn < p.N && gkb < kend;
      const long offb = TB ? (long)gn * p.ldb + gkb : (long)gkb * p.ldb + gn;
      rb[e] = okb ? B[offb] : 0.f;
    }
    __syncthreads();  // the previous step's tile reads are over
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      if (TA) As[idx >> 6][idx & (GBM - 1)] = ra[e]; else As[idx & (GBK - 1)][idx >> 4] = ra[e];
      if (TB) Bs[idx & (GBK - 1)][idx >> 4] = rb[e]; else Bs[idx >> 6][idx & (GBN - 1)] = rb[e];
    }
    __syncthreads();
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

extern "C" int vt_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, long strideA,
                           const float *B, int ldb, long strideB, int accumulate, float *C, int ldc, long strideC,
                           const float *bias, int batch, float *workspace, long workspace_floats, vt_stream s) {
  if (M < 1 || N < 1 || K < 1 || batch < 1 || !A || !B || !C) return VT_EINVAL;
  if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return VT_EINVAL;
  GemmP p{A, B, C, bias, M, N, K, lda, ldb, ldc, strideA, strideB, strideC, alpha, accumulate ? 1 : 0, 1, K, batch, nullptr};
  const long tiles = (long)((M + GBM - 1) / GBM) * ((N + GBN - 1) / GBN) * batch;
  // a long contraction over few output tiles: slice K so that about 1024 workgroups exist
  if (workspace && K >= 4096 && tiles < 256) {
    long want = (1024 + tiles - 1) / tiles;
    want = want < (K + 511) / 512 ? want : (K + 511) / 512;
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
  if (transA && transB) hipLaunchKernelGGL((k_gemm<true, true>), grid, dim3(256), 0, hs, p);
  else if (transA) hipLaunchKernelGGL((k_gemm<true, false>), grid, dim3(256), 0, hs, p);
  else if (transB) hipLaunchKernelGGL((k_gemm<false, true>), grid, dim3(256), 0, hs, p);
  else hipLaunchKernelGGL((k_gemm<false, false>), grid, dim3(256), 0, hs, p);
  VT_CHECK_LAUNCH();
  if (p.ksplit > 1) {
    const long total = (long)M * N * batch;
    const int nb = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_gemm_splitk_reduce, dim3(nb), dim3(256), 0, hs, p, batch);
    VT_CHECK_LAUNCH();
  }
  return 0;
}
