// Classifier head of VTCNN for gfx950: projector BatchNorm1D apply + ReLU, the [N,P,C2] ->
// [N,C2,H,W] reinterpretation, max-pool, fc, softmax, and their backward.
//
// The reinterpretation (reference model/vit_cnn.hpp:179) moves no data: the P*C2 floats of one
// image are simply read as C2 planes of H*W.  A pooling window therefore touches flat offsets
// base, base+1, ..., base+W, ... of the image's record block, and the BatchNorm channel of a flat
// offset F is F mod C2.  Both kernels work straight from global memory (no LDS staging), so many
// workgroups fit per CU and latency is hidden by occupancy.
#include "vtcnn_common.h"

namespace vt {

// BatchNorm1D apply of one element (x_norm * gamma + beta, batch_normalization_layer.hpp:155,168)
__device__ __forceinline__ float bn_apply(float v, const float *__restrict__ sbn, int c, float *xhat) {
  const float xh = (v - sbn[c]) * sbn[C2 + c];
  if (xhat) *xhat = xh;
  return xh * sbn[2 * C2 + c] + sbn[3 * C2 + c];
}

// fc weight row W[f][0..K) into registers with the widest loads the row alignment allows
// (K = 8: two 16-byte loads; K even: 8-byte loads; otherwise scalar).
template <int K>
__device__ __forceinline__ void load_wrow(const float *__restrict__ Wfc, int f, float (&w)[K]) {
  const float *src = Wfc + (size_t)f * K;
  if constexpr (K % 4 == 0) {
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
      const float4 t = reinterpret_cast<const float4 *>(src)[q];
      w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
    }
  } else if constexpr (K % 2 == 0) {
#pragma unroll
    for (int q = 0; q < K / 2; ++q) {
      const float2 t = reinterpret_cast<const float2 *>(src)[q];
      w[2 * q] = t.x; w[2 * q + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) w[k] = src[k];
  }
}

// ---------------------------------------------------------------------------------------------
// Forward: ReLU(BN(z)) (projector_layer.hpp:106-113) -> MaxPool2DLayer::forward, first strictly
// greater element wins (layers/pooling_layer.hpp:40-59) -> Flatten -> fc -> softmax
// (vit_cnn.hpp:191-203, tensor_cpu.hpp:894-944).  One workgroup per image, one thread per pooled
// output (12.25 per thread at 28x28), processed HB at a time: the window and weight-row loads of
// a batch are all issued before the first is used.  K is a template parameter so that the weight
// row is a few vector loads and the multiply-adds carry no branches.
// ---------------------------------------------------------------------------------------------
constexpr int HB = 4;  // pooled outputs in flight per thread
template <int K>
__global__ __launch_bounds__(TPB) void k_head_fwd(const float *__restrict__ z, const float *__restrict__ st,
                                                  const float *__restrict__ g, const float *__restrict__ b,
                                                  const float *__restrict__ Wfc, const float *__restrict__ bfc,
                                                  float *__restrict__ pooled, unsigned char *__restrict__ pool_idx,
                                                  float *__restrict__ logits, float *__restrict__ probs, int H,
                                                  int group) {
  __shared__ float sbn[4 * C2];
  __shared__ float red[NW * K];
  const int W = H, P = H * W, pk = H / 14, Q = C2 * 196;
  const int n = blockIdx.x, tid = threadIdx.x;
  if (group > 0) st += (size_t)(n / group) * 2 * C2;  // statistics group of this image
  if (tid < C2) {
    sbn[tid] = st[tid];
    sbn[C2 + tid] = st[C2 + tid];
    sbn[2 * C2 + tid] = g[tid];
    sbn[3 * C2 + tid] = b[tid];
  }
  __syncthreads();
  const float *__restrict__ zi = z + (size_t)n * P * C2;
  float acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0.f;
  if (pk == 2) {  // 28x28: two aligned 8-byte loads per window
    for (int f0 = tid; f0 < Q; f0 += HB * TPB) {
      float2 r0[HB], r1[HB];
      float w[HB][K];
      int base[HB];
#pragma unroll
      for (int u = 0; u < HB; ++u) {
        const int f = min(f0 + u * TPB, Q - 1);
        const int ch = f / 196, ij = f % 196, i = ij / 14, jj = ij % 14;
        base[u] = ch * P + (i * 2) * W + jj * 2;
        r0[u] = *reinterpret_cast<const float2 *>(zi + base[u]);
        r1[u] = *reinterpret_cast<const float2 *>(zi + base[u] + W);
        load_wrow<K>(Wfc, f, w[u]);
      }
#pragma unroll
      for (int u = 0; u < HB; ++u) {
        const int f = f0 + u * TPB;
        const float v[4] = {r0[u].x, r0[u].y, r1[u].x, r1[u].y};
        const int off[4] = {0, 1, W, W + 1};
        float mv = -INFINITY;
        int mi = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float y = fmaxf(bn_apply(v[q], sbn, (base[u] + off[q]) & (C2 - 1), nullptr), 0.f);
          if (y > mv) {
            mv = y;
            mi = q;
          }
        }
        if (f < Q) {
          pooled[(size_t)n * Q + f] = mv;
          pool_idx[(size_t)n * Q + f] = (unsigned char)mi;
#pragma unroll
          for (int k = 0; k < K; ++k) acc[k] += mv * w[u][k];
        }
      }
    }
  } else {
    for (int f = tid; f < Q; f += TPB) {
      const int ch = f / 196, ij = f % 196, i = ij / 14, jj = ij % 14;
      const int base = ch * P + (i * pk) * W + jj * pk;
      float mv = -INFINITY;
      int mi = 0;
      for (int ki = 0; ki < pk; ++ki)
        for (int kj = 0; kj < pk; ++kj) {
          const int F = base + ki * W + kj;
          const float y = fmaxf(bn_apply(zi[F], sbn, F & (C2 - 1), nullptr), 0.f);
          if (y > mv) {
            mv = y;
            mi = ki * pk + kj;
          }
        }
      pooled[(size_t)n * Q + f] = mv;
      pool_idx[(size_t)n * Q + f] = (unsigned char)mi;
      float w[K];
      load_wrow<K>(Wfc, f, w);
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] += mv * w[k];
    }
  }
  block_sum<K>(acc, red);
  if (tid == 0) {
    float lg[K], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      lg[k] = acc[k] + bfc[k];
      mx = fmaxf(mx, lg[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      logits[(size_t)n * K + k] = lg[k];
      lg[k] = expf(lg[k] - mx);
      sum += lg[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) probs[(size_t)n * K + k] = lg[k] / sum;
  }
}

// ---------------------------------------------------------------------------------------------
// Backward.  SoftmaxLayer::backward with targets: delta = softmax - onehot, no 1/N
// (reference layers/softmax_layer.hpp:29); fc input delta = delta . W^T (dense_layer.hpp:98-99);
// MaxPool2DLayer::backward routes it to the arg-max slot (pooling_layer.hpp:88-92); ReLU mask on
// the flat reinterpretation (relu_layer.hpp:20-33, projector_layer.hpp:121-131); partial sums of
// the projector BatchNorm1D backward (batch_normalization_layer.hpp:187-188).
// Phase 1 forms the fc delta of every pooling window (K MACs each) into a 12.5 KB LDS table.
// Phase 2 is a gather: one thread per pixel record (C2 consecutive flat offsets); each element
// looks up its window's arg-max slot and takes the window's delta if it is the winner and active.
// Every element of du is written exactly once; no scatter, no zero fill, and the per-channel sums
// fall out of the record layout.
// ---------------------------------------------------------------------------------------------
template <int PKC, int K>  // PKC = compile-time pool kernel (2 for 28x28 inputs), 0 = use the runtime value
__global__ __launch_bounds__(TPB) void k_head_bwd(const float *__restrict__ probs, const int *__restrict__ labels,
                                                  const float *__restrict__ Wfc, const unsigned char *__restrict__ pool_idx,
                                                  const float *__restrict__ z, const float *__restrict__ st,
                                                  const float *__restrict__ g, const float *__restrict__ b,
                                                  float *__restrict__ dlogits, float *__restrict__ du,
                                                  float *__restrict__ part, int H) {
  __shared__ float red[NW * 2 * C2];
  const int W = H, P = H * W, pk = PKC ? PKC : H / 14, Q = C2 * 196;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int label = labels[n];
  float dl[K];
#pragma unroll
  for (int k = 0; k < K; ++k) dl[k] = probs[(size_t)n * K + k] - (k == label ? 1.0f : 0.0f);
  if (tid < K) dlogits[(size_t)n * K + tid] = probs[(size_t)n * K + tid] - (tid == label ? 1.0f : 0.0f);
  const unsigned char *__restrict__ pix = pool_idx + (size_t)n * Q;
  // phase 1: fc input delta of every pooling window (one window per thread trip), kept in LDS;
  // the weight rows of HB windows are loaded together
  // the first pixel record of this thread is requested now: its trip to HBM runs under phase 1, which
  // only touches the (cached) fc weights
  float4 zpre[C2 / 4];
  {
    const float4 *src0 = reinterpret_cast<const float4 *>(z + ((size_t)n * P + min(tid, P - 1)) * C2);
#pragma unroll
    for (int q = 0; q < C2 / 4; ++q) zpre[q] = src0[q];
  }
  extern __shared__ float sdf[];  // Q floats
  for (int f0 = tid; f0 < Q; f0 += HB * TPB) {
    float w[HB][K];
#pragma unroll
    for (int u = 0; u < HB; ++u) load_wrow<K>(Wfc, min(f0 + u * TPB, Q - 1), w[u]);
#pragma unroll
    for (int u = 0; u < HB; ++u) {
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) d += dl[k] * w[u][k];
      if (f0 + u * TPB < Q) sdf[f0 + u * TPB] = d;
    }
  }
  __syncthreads();
  float s[2 * C2];
#pragma unroll
  for (int c = 0; c < 2 * C2; ++c) s[c] = 0.f;
  for (int p = tid; p < P; p += TPB) {
    const float4 *src = reinterpret_cast<const float4 *>(z + ((size_t)n * P + p) * C2);
    float4 *dst = reinterpret_cast<float4 *>(du + ((size_t)n * P + p) * C2);
    const int F0 = p * C2;
    int ch = F0 / P, rem = F0 - ch * P, h = rem / W, w = rem - h * W;
    if (PKC == 2 && (W & 3) == 0) {
      // 2x2 windows, row length a multiple of 4: each group of 4 consecutive elements lies in one
      // row and covers exactly two windows f0, f0+1 (f0 even).  Everything a group needs (two slot
      // bytes, two fc rows) is loaded unconditionally, so all loads of a pixel are independent.
#pragma unroll
      for (int q = 0; q < C2 / 4; ++q) {
        const float4 v4 = p == tid ? zpre[q] : src[q];
        const float zv[4] = {v4.x, v4.y, v4.z, v4.w};
        const int f0 = ch * 196 + (h >> 1) * 14 + (w >> 1);
        const unsigned short sl2 = *reinterpret_cast<const unsigned short *>(pix + f0);
        const int win_slot[2] = {sl2 & 0xff, sl2 >> 8};
        const float2 df2 = *reinterpret_cast<const float2 *>(sdf + f0);
        const float df[2] = {df2.x, df2.y};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * q + e;
          const float xh = (zv[e] - st[c]) * st[C2 + c];
          const float y = xh * g[c] + b[c];
          const int slot = (h & 1) * 2 + (e & 1);
          const float d = (y > 0.f && win_slot[e >> 1] == slot) ? df[e >> 1] : 0.f;
          o[e] = d;
          s[c] += d;
          s[C2 + c] += d * xh;
        }
        dst[q] = make_float4(o[0], o[1], o[2], o[3]);
        w += 4;
        if (w == W) {
          w = 0;
          if (++h == H) {
            h = 0;
            ++ch;
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int q = 0; q < C2 / 4; ++q) {
      const float4 v4 = src[q];
      const float zv[4] = {v4.x, v4.y, v4.z, v4.w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * q + e;
        const float xh = (zv[e] - st[c]) * st[C2 + c];
        const float y = xh * g[c] + b[c];
        const int f = ch * 196 + (h / pk) * 14 + (w / pk);
        const int slot = (h % pk) * pk + (w % pk);
        const float d = (y > 0.f && pix[f] == slot) ? sdf[f] : 0.f;
        o[e] = d;
        s[c] += d;
        s[C2 + c] += d * xh;
        if (++w == W) {
          w = 0;
          if (++h == H) {
            h = 0;
            ++ch;
          }
        }
      }
      dst[q] = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
  block_sum<2 * C2>(s, red);
  if (tid < 2 * C2) part[(size_t)n * 2 * C2 + tid] = s[tid];
}

}  // namespace vt

using namespace vt;

template <int K>
static void launch_head_fwd(const float *z, const float *pn_stat, const float *pn_gamma, const float *pn_beta,
                            const float *wfc, const float *bfc, float *pooled, unsigned char *pool_idx, float *logits,
                            float *probs, int N, int H, int group, hipStream_t s) {
  hipLaunchKernelGGL(k_head_fwd<K>, dim3(N), dim3(TPB), 0, s, z, pn_stat, pn_gamma, pn_beta, wfc, bfc, pooled, pool_idx,
                     logits, probs, H, group);
}
template <int K>
static void launch_head_bwd(const float *probs, const int *labels, const float *wfc, const unsigned char *pool_idx,
                            const float *z, const float *pn_stat, const float *pn_gamma, const float *pn_beta,
                            float *d_logits, float *du, float *part, int N, int H, hipStream_t s) {
  const size_t lds = (size_t)C2 * 196 * sizeof(float);
  if (H == 28)
    hipLaunchKernelGGL((k_head_bwd<2, K>), dim3(N), dim3(TPB), lds, s, probs, labels, wfc, pool_idx, z, pn_stat,
                       pn_gamma, pn_beta, d_logits, du, part, H);
  else
    hipLaunchKernelGGL((k_head_bwd<0, K>), dim3(N), dim3(TPB), lds, s, probs, labels, wfc, pool_idx, z, pn_stat,
                       pn_gamma, pn_beta, d_logits, du, part, H);
}
// one instantiation per class count 1..KMAX (the count fixes the fc row width and its alignment)
#define VT_FOR_K(CALL)                                                                                      \
  switch (K) {                                                                                              \
    case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break;       \
    case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break;       \
    case 9: CALL(9); break; case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break; \
    case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; case 16: CALL(16); break; \
    default: return VT_EUNSUPPORTED;                                                                        \
  }

extern "C" int vt_head_fwd_grouped(const float *z, const float *pn_stat, const float *pn_gamma, const float *pn_beta,
                                   const float *wfc, const float *bfc, float *pooled, unsigned char *pool_idx,
                                   float *logits, float *probs, int N, int H, int K, int group, vt_stream s) {
  if (N < 1 || H < 14 || H % 14 != 0 || K < 1 || K > KMAX || group < 0) return VT_EUNSUPPORTED;
#define CALL(KK) launch_head_fwd<KK>(z, pn_stat, pn_gamma, pn_beta, wfc, bfc, pooled, pool_idx, logits, probs, N, H, group, (hipStream_t)s)
  VT_FOR_K(CALL)
#undef CALL
  VT_CHECK_LAUNCH();
  return 0;
}
extern "C" int vt_head_fwd(const float *z, const float *pn_stat, const float *pn_gamma, const float *pn_beta,
                           const float *wfc, const float *bfc, float *pooled, unsigned char *pool_idx, float *logits,
                           float *probs, int N, int H, int K, vt_stream s) {
  return vt_head_fwd_grouped(z, pn_stat, pn_gamma, pn_beta, wfc, bfc, pooled, pool_idx, logits, probs, N, H, K, 0, s);
}

extern "C" int vt_head_bwd(const float *probs, const int *labels, const float *wfc, const unsigned char *pool_idx,
                           const float *z, const float *pn_stat, const float *pn_gamma, const float *pn_beta,
                           float *d_logits, float *du, float *part, int N, int H, int K, vt_stream s) {
  if (N < 1 || H < 14 || H % 14 != 0 || K < 1 || K > KMAX) return VT_EUNSUPPORTED;
#define CALL(KK) launch_head_bwd<KK>(probs, labels, wfc, pool_idx, z, pn_stat, pn_gamma, pn_beta, d_logits, du, part, N, H, (hipStream_t)s)
  VT_FOR_K(CALL)
#undef CALL
  VT_CHECK_LAUNCH();
  return 0;
}
