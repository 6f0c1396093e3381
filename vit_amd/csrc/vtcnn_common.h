// Shared device helpers for the gfx950 VTCNN kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>

#include "../../include/vtcnn_hip.h"

namespace vt {

// The one compiled instantiation: the reference's hyper-parameters (main.cpp:66-78).
constexpr int C1 = 8;    // conv / "in" channels = vt_channels / 2
constexpr int C2 = 16;   // vt channels
constexpr int L = 16;    // tokens
constexpr int D = 32;    // token channels = attention dim
constexpr int HEADS = 2;
constexpr int HD = D / HEADS;
constexpr int KMAX = 16; // classes supported by the fused head
constexpr int TPB = 256; // threads per workgroup (4 waves)
constexpr int NW = TPB / 64;
constexpr int PPT = 4;   // pixels per thread: up to 1024 pixels per sample
constexpr int RB = 64;   // token rows per workgroup in the row-matrix kernels
constexpr int PK_STRIDE = C1 * L + L * C2 + L * D; // per-sample projector token block (Mq, v, k)
constexpr float BN_EPS = 1e-5f;
constexpr float BN_MOM = 0.1f;

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
  return x;
}

// Sum of NV per-thread values over the whole workgroup; every thread receives the totals.
// red: LDS scratch of NW*NV floats.  Fixed summation order => bitwise reproducible.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * NV + i];
    v[i] = s;
  }
}

// The same for a workgroup of NWV waves (kernels that run more than TPB threads per image).
template <int NV, int NWV>
__device__ __forceinline__ void block_sum_w(float (&v)[NV], float *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) s += red[w * NV + i];
    v[i] = s;
  }
}

template <int NV>
__device__ __forceinline__ void block_max(float (&v)[NV], float *red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_max(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = red[i];
#pragma unroll
    for (int w = 1; w < NW; ++w) s = fmaxf(s, red[w * NV + i]);
    v[i] = s;
  }
}

// Transposing butterfly reduction inside one wave: every lane contributes NV values (NV = 64*m);
// afterwards lane l holds in v[0..m) the wave-wide sums of elements l*m .. l*m+m-1.
// Costs NV-m shuffles instead of 6*NV.
template <int NV>
__device__ __forceinline__ void wave_xreduce(float (&v)[NV]) {
  static_assert(NV % 64 == 0, "NV must be a multiple of the wave size");
  const int lane = threadIdx.x & 63;
  int n = NV / 2;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const bool up = (lane & off) != 0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const float keep = up ? v[n + i] : v[i];
      const float send = up ? v[i] : v[n + i];
      v[i] = keep + __shfl_xor(send, off);
    }
    n >>= 1;
  }
}

// Workgroup version: out[NV] (LDS) receives the totals of NV per-thread values.
// red: LDS scratch of NW*NV floats.  Ends with a barrier.
template <int NV>
__device__ __forceinline__ void block_xreduce(float (&v)[NV], float *red, float *out) {
  constexpr int m = NV / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wave_xreduce<NV>(v);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < m; ++k) red[wave * NV + lane * m + k] = v[k];
  __syncthreads();
  for (int i = threadIdx.x; i < NV; i += TPB) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * NV + i];
    out[i] = s;
  }
  __syncthreads();
}

// BatchNorm2D apply for one pixel record (x_norm * gamma + beta, batch_normalization_layer.hpp:279,296).
__device__ __forceinline__ void load_x(const float *__restrict__ conv, size_t rec, const float *__restrict__ st,
                                       const float *__restrict__ g, const float *__restrict__ b, float (&x)[C1]) {
  const float4 *src = reinterpret_cast<const float4 *>(conv + rec * C1);
  const float4 a = src[0], c = src[1];
  const float raw[C1] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
  for (int ch = 0; ch < C1; ++ch) x[ch] = ((raw[ch] - st[ch]) * st[C1 + ch]) * g[ch] + b[ch];
}


// Column reductions of partial-gradient slabs (out[c] = sum_r slab[r][c]).  A launcher that needs
// one calls reduce_rows(); while a ReduceBatch is open on the calling host thread the request is
// only queued, and reduce_batch_flush() then performs every queued reduction in ONE launch.  The
// model opens a batch around its backward pass: nine 5-microsecond launches become one.
struct ReduceJob {
  const float *slab;
  float *out;
  long rows;
  int cols, ld, first_block;
};
constexpr int MAX_REDUCE_JOBS = 16;
struct ReduceBatch {
  ReduceJob job[MAX_REDUCE_JOBS];
  int n = 0, blocks = 0;
};
int reduce_rows(const float *slab, long rows, int cols, int ld, float *out, hipStream_t s);
void reduce_batch_begin(ReduceBatch *b);         // b must outlive the matching flush / abandon
int reduce_batch_flush(hipStream_t s);           // launches the queued jobs and closes the batch
void reduce_batch_abandon();                     // closes the batch without launching
// flush whose reductions feed the SGD update of the whole trainable arena in the same launch (w, g: arena and
// gradient arena of n floats; gradients end cleared)
int reduce_batch_flush_sgd(float *w, float *g, int n, float lr, float divisor, hipStream_t s);
// vt_sgd_update (gradients cleared) that leaves floats [hole_first, hole_first + hole_count) untouched
int sgd_update_hole(float *w, float *g, int n, float lr, float divisor, int hole_first, int hole_count, hipStream_t s);

}  // namespace vt

// A launch that carries an event of its own completion: when the model has put an event into g_stop_event, the
// next launch through this macro is made with hipExtLaunchKernelGGL and signals that event itself when it
// finishes, so no separate event-record packet (which the stream's next kernel would wait behind) follows the
// kernel on its stream.  The launch takes the event and clears the slot.
namespace vt {
extern thread_local hipEvent_t g_stop_event;
}
#define VT_LAUNCH_SIGNALLING(kernel, grid, block, lds, stream, ...)                                          \
  do {                                                                                                       \
    if (vt::g_stop_event) {                                                                                  \
      hipEvent_t ev__ = vt::g_stop_event;                                                                    \
      vt::g_stop_event = nullptr;                                                                            \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, ev__, 0, __VA_ARGS__);                \
    } else {                                                                                                 \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                     \
    }                                                                                                        \
  } while (0)

#define VT_CHECK_LAUNCH()                      \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)
