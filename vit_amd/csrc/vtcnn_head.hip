// Classifier head of VTCNN for gfx950: projector BatchNorm1D apply + ReLU, the [N,P,C2] ->
// [N,C2,H,W] reinterpretation, max-pool, fc, softmax, and their backward.
//
// The reinterpretation (reference model/vit_cnn.hpp:179) moves no data: the P*C2 floats of one
// image are simply read as C2 planes of H*W.  A pooling window therefore touches flat offsets
// base, base+1, ..., base+W, ... of the image's record block, and the BatchNorm channel of a flat
// offset F is F mod C2.  Both kernels work straight from global memory (no LDS staging), so many
// workgroups fit per CU and latency is hidden by occupancy.
#include <algorithm>

#include "vtcnn_common.h"
#include "vtcnn_tail.h"

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

// ---------------------------------------------------------------------------------------------
// The same backward for 2x2 pooling of 28x28 maps, without the dense delta: per pooling window only
// the arg-max element can receive a delta, and the ReLU gate is known from the window maximum.  The delta
// leaves as dpair [N, P*C2/2]: one float per horizontally adjacent element pair (flat offsets F, F+1, F even)
// of the [N,C2,H,W] view, i.e. per window two entries, top row and bottom row: the entry of the row that
// holds the arg-max carries the window's fc input delta with its lowest mantissa bit replaced by the arg-max
// column (0 = left, 1 = right), the other entry is 0.  That is half the dense [N,P,C2] tensor, in the
// order in which the projector backward walks pixel records, with nothing to look up.
// Nothing is read from z either: the window maximum
// mv kept by the forward pass is the post-ReLU value of the winning element, so that element was active
// iff mv > 0 and its normalised input is x_hat = (mv - beta) / gamma (z is consulted only for a channel
// whose gamma is exactly 0, where the BatchNorm output no longer determines x_hat).
// Window-major: a thread owns one window f for a chunk of HW_NB images, so its fc weight row is loaded
// once (one workgroup per image would stream the whole 100 KB matrix through every workgroup), the
// loads and stores of a wave are 256 contiguous bytes per image, and the BatchNorm channel of the
// window's four slots is fixed: the partial sums (sum d, sum d*x_hat) are four predicated accumulators
// per thread, brought to channel order through LDS in a fixed order at the end.
// part receives one [2][C2] record per workgroup (gridDim.x * gridDim.y of them).
// ---------------------------------------------------------------------------------------------
constexpr int HW_NB = 16;   // most images per workgroup
constexpr int HW_RS = 16;   // LDS row stride of the per-thread channel bins
constexpr int HW_B1 = TPB * HW_RS + 16;  // start of the second bin array: half the banks away from the first
template <int K, bool TAIL>  // TAIL: the last workgroup closes the projector BatchNorm1D backward reduction
__global__ __launch_bounds__(TPB) void k_head_bwd_win(const float *__restrict__ probs, const int *__restrict__ labels,
                                                      const float *__restrict__ Wfc,
                                                      const unsigned char *__restrict__ pool_idx,
                                                      const float *__restrict__ pooled, const float *__restrict__ z,
                                                      const float *__restrict__ st, const float *__restrict__ g,
                                                      const float *__restrict__ b, float *__restrict__ dlogits,
                                                      float *__restrict__ dpair, float *__restrict__ part, int N, int nb,
                                                      int sgroup, const BnBwdTail tail) {
  constexpr int W = 28, P = 784, Q = C2 * 196;
  __shared__ __attribute__((aligned(16))) float sbin[HW_B1 + TPB * HW_RS];  // [sum d | sum d*x_hat][thread][channel]
  __shared__ float sred[8][2 * C2];
  const int tid = threadIdx.x;
  const int f = blockIdx.x * TPB + tid, fc = min(f, Q - 1);
  const bool okf = f < Q;
  const int n0 = blockIdx.y * nb, nn = min(nb, N - n0);
  if (sgroup > 0) st += (size_t)(n0 / sgroup) * 2 * C2;  // a chunk never straddles two statistics groups
  // every load of the workgroup is requested before anything waits: the window maxima and slot bytes of all its
  // images (one trip to memory for the whole chunk) and the weight row
  float mv[HW_NB];
  int sl[HW_NB];
#pragma unroll
  for (int u = 0; u < HW_NB; ++u) {
    const size_t o = (size_t)(n0 + min(u, nn - 1)) * Q + fc;
    mv[u] = pooled[o];
    sl[u] = pool_idx[o];
  }
  float w[K];
  load_wrow<K>(Wfc, fc, w);
  if (blockIdx.x == 0)
    for (int idx = tid; idx < nn * K; idx += TPB) {
      const int n = n0 + idx / K, k = idx % K;
      dlogits[(size_t)n * K + k] = probs[(size_t)n * K + k] - (k == labels[n] ? 1.0f : 0.0f);
    }
  // the four elements of the window in the [N,C2,H,W] view and their BatchNorm channels (flat offset mod C2)
  const int ch = fc / 196, ij = fc - ch * 196, i = ij / 14, jj = ij - i * 14;
  const int F00 = ch * P + (2 * i) * W + 2 * jj;
  int cq[4];
  float gq4[4], bq4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    cq[q] = (F00 + (q >> 1) * W + (q & 1)) & (C2 - 1);
    gq4[q] = g[cq[q]];
    bq4[q] = b[cq[q]];
  }
  // this thread's channel bins start at zero
  {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(&sbin[a * HW_B1 + tid * HW_RS + 4 * q]) = z4;
  }
  // per slot q: sum of the deltas routed to it and of delta * window maximum; x_hat = (mv - beta) / gamma turns the
  // pair into sum d*x_hat after the loop
  float ad[4] = {0.f, 0.f, 0.f, 0.f}, am[4] = {0.f, 0.f, 0.f, 0.f};
  const int e_top = ch * (P / 2) + (2 * i) * (W / 2) + jj;  // pair index of the window's top row; bottom: + W/2
#pragma unroll
  for (int u = 0; u < HW_NB; ++u) {
    if (u < nn) {  // uniform
      const int n = n0 + u;
      const int label = labels[n];  // uniform: scalar loads
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) d += (probs[(size_t)n * K + k] - (k == label ? 1.0f : 0.0f)) * w[k];
      d = (okf && mv[u] > 0.f) ? d : 0.f;
      if (okf) {
        const float enc = __uint_as_float((__float_as_uint(d) & ~1u) | (unsigned)(sl[u] & 1));
        float *dst = dpair + (size_t)n * (P * C2 / 2) + e_top;
        dst[0] = (sl[u] >> 1) == 0 ? enc : 0.f;
        dst[W / 2] = (sl[u] >> 1) == 1 ? enc : 0.f;
      }
      const float dm = d * mv[u];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool hit = sl[u] == q;
        ad[q] += hit ? d : 0.f;
        am[q] += hit ? dm : 0.f;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int F = F00 + (q >> 1) * W + (q & 1), c = cq[q];
    const float gq = gq4[q], bq = bq4[q];
    float ax;
    // x_hat from the post-BatchNorm maximum is (mv - beta) / gamma: its rounding error is eps * |beta| / |gamma|, so a
    // gamma that is small against beta (or zero) takes x_hat from z instead
    if (fabsf(gq) >= 1e-3f * fmaxf(1.0f, fabsf(bq))) {
      ax = (am[q] - bq * ad[q]) / gq;
    } else {
      ax = 0.f;
      for (int u = 0; u < nn; ++u) {
        const size_t o = (size_t)(n0 + u) * Q + fc;
        if (okf && pool_idx[o] == q && pooled[o] > 0.f) {
          float d = 0.f;
          const int label = labels[n0 + u];
          for (int k = 0; k < K; ++k) d += (probs[(size_t)(n0 + u) * K + k] - (k == label ? 1.0f : 0.0f)) * w[k];
          ax += d * (z[(size_t)(n0 + u) * P * C2 + F] - st[c]) * st[C2 + c];
        }
      }
    }
    sbin[tid * HW_RS + c] = ad[q];  // the four channels of a window are distinct
    sbin[HW_B1 + tid * HW_RS + c] = ax;
  }
  __syncthreads();
  {
    // thread = (channel c, which sum, eighth of the workgroup): 32 bins in a fixed order
    const int c = tid & 15, which = (tid >> 4) & 1, e8 = tid >> 5;
    const float *col = &sbin[which * HW_B1 + (e8 * 32) * HW_RS + c];
    float sum = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) sum += col[r * HW_RS];
    sred[e8][which * C2 + c] = sum;
  }
  __syncthreads();
  if (tid < 2 * C2) {
    float tot = 0.f;
#pragma unroll
    for (int e8 = 0; e8 < 8; ++e8) tot += sred[e8][tid];
    float *dst = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C2 + tid;
    if constexpr (!TAIL) *dst = tot;
    else tail_store(dst, tot);
  }
  if constexpr (TAIL) bn_bwd_tail(tail, part, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, C2);
}

// dpair -> the dense [N,P,C2] delta (what k_head_bwd writes): for hosts and tests that want it
__global__ __launch_bounds__(TPB) void k_head_expand_win(const float *__restrict__ dpair, float *__restrict__ du, long total) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const unsigned bits = __float_as_uint(dpair[i >> 1]);
  du[i] = (bits & 1u) == (unsigned)(i & 1) ? __uint_as_float(bits & ~1u) : 0.f;
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

// Images per workgroup and workgroups of k_head_bwd_win for a mini-batch of N: about 64 chunks of images at
// most HW_NB long, times the 13 window blocks.
// With statistics groups of sgroup images a chunk must not straddle two groups: the chunk length divides sgroup.
static int head_win_chunk(int N, int sgroup = 0) {
  int nb = std::max(1, std::min(HW_NB, N / 64));
  if (sgroup > 0) {
    nb = std::min(nb, sgroup);
    while (sgroup % nb != 0) --nb;
  }
  return nb;
}
extern "C" int vt_head_bwd_windows_blocks(int N) {
  if (N < 1) return VT_EINVAL;
  const int nb = head_win_chunk(N);
  return ((C2 * 196 + TPB - 1) / TPB) * ((N + nb - 1) / nb);
}
// partial records of the grouped launch: total, and per full statistics group
int vt::head_bwd_windows_blocks_grouped(int N, int sgroup, int *per_group) {
  const int nb = head_win_chunk(N, sgroup), wb = (C2 * 196 + TPB - 1) / TPB;
  if (per_group) *per_group = sgroup > 0 ? wb * (sgroup / nb) : wb * ((N + nb - 1) / nb);
  return wb * ((N + nb - 1) / nb);
}
int vt::head_bwd_windows_tail(const float *probs, const int *labels, const float *wfc, const unsigned char *pool_idx,
                              const float *pooled, const float *z, const float *pn_stat, const float *pn_gamma,
                              const float *pn_beta, float *d_logits, float *dpair, float *part, int N, int H, int K,
                              const BnBwdTail &tail, hipStream_t s, int sgroup) {
  if (N < 1 || H != 28 || K < 1 || K > KMAX || sgroup < 0 || (sgroup > 0 && tail.ticket)) return VT_EUNSUPPORTED;
  const int nb = head_win_chunk(N, sgroup);
  const dim3 grid((C2 * 196 + TPB - 1) / TPB, (N + nb - 1) / nb);
#define CALL(KK)                                                                                                        \
  if (tail.ticket)                                                                                                      \
    hipLaunchKernelGGL((k_head_bwd_win<KK, true>), grid, dim3(TPB), 0, s, probs, labels, wfc, pool_idx, pooled, z, pn_stat, \
                       pn_gamma, pn_beta, d_logits, dpair, part, N, nb, 0, tail);                                       \
  else                                                                                                                  \
    hipLaunchKernelGGL((k_head_bwd_win<KK, false>), grid, dim3(TPB), 0, s, probs, labels, wfc, pool_idx, pooled, z, pn_stat, \
                       pn_gamma, pn_beta, d_logits, dpair, part, N, nb, sgroup, tail)
  VT_FOR_K(CALL)
#undef CALL
  VT_CHECK_LAUNCH();
  return 0;
}
extern "C" int vt_head_bwd_windows(const float *probs, const int *labels, const float *wfc, const unsigned char *pool_idx,
                                   const float *pooled, const float *z, const float *pn_stat, const float *pn_gamma,
                                   const float *pn_beta, float *d_logits, float *dpair, float *part, int N, int H, int K,
                                   vt_stream s) {
  return head_bwd_windows_tail(probs, labels, wfc, pool_idx, pooled, z, pn_stat, pn_gamma, pn_beta, d_logits, dpair, part, N, H, K,
                               BnBwdTail{}, (hipStream_t)s, 0);
}

extern "C" int vt_head_expand_windows(const float *dpair, float *du, int N, int H, vt_stream s) {
  if (N < 1 || H != 28) return VT_EUNSUPPORTED;
  const long total = (long)N * 784 * C2;
  hipLaunchKernelGGL(k_head_expand_win, dim3((unsigned)((total + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)s, dpair, du, total);
  VT_CHECK_LAUNCH();
  return 0;
}
