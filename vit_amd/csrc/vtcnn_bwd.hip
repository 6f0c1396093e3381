// Backward kernels of the VTCNN hot path for gfx950 (MI355X).
//
// The backward pass is cut at the four BatchNorm backward reductions (sum delta, sum delta*xhat
// over the whole mini-batch) and at the weight-gradient reductions.  Every cross-workgroup sum
// goes through a partial slab + fixed-order column reduction (no float atomics: results are
// bitwise reproducible).  Each kernel names the reference code it restates, quirks included.
#include "vtcnn_common.h"

namespace vt {

// Column sums of a [rows][ld] slab: out[c] = sum_r slab[r][c], c < cols.  1024 threads =
// 64 columns x 16 row lanes; fixed order => reproducible.
__global__ __launch_bounds__(1024) void k_reduce_rows(const float *__restrict__ slab, long rows, int cols, int ld,
                                                      float *__restrict__ out) {
  __shared__ float sh[16][65];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < cols) {
    long r = rl;
    for (; r + 7 * 16 < rows; r += 8 * 16) {  // 8 independent loads in flight
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = slab[(r + 16 * k) * ld + c];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; r < rows; r += 16) s += slab[r * ld + c];
  }
  sh[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][cl];
    out[c] = t;
  }
}

// The same reduction for up to MAX_REDUCE_JOBS slabs in one launch: workgroup b serves the job
// whose [first_block, next first_block) range contains b.
__global__ __launch_bounds__(1024) void k_reduce_multi(const ReduceBatch batch) {
  __shared__ float sh[16][65];
  int ji = 0;
#pragma unroll 1
  for (int k = 1; k < batch.n; ++k)
    if ((int)blockIdx.x >= batch.job[k].first_block) ji = k;
  const float *__restrict__ slab = batch.job[ji].slab;
  const long rows = batch.job[ji].rows;
  const int cols = batch.job[ji].cols, ld = batch.job[ji].ld;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = ((int)blockIdx.x - batch.job[ji].first_block) * 64 + cl;
  float s = 0.f;
  if (c < cols) {
    long r = rl;
    for (; r + 7 * 16 < rows; r += 8 * 16) {  // 8 independent loads in flight
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = slab[(r + 16 * k) * ld + c];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; r < rows; r += 16) s += slab[r * ld + c];
  }
  sh[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][cl];
    batch.job[ji].out[c] = t;
  }
}

// k_reduce_multi whose results are consumed on the spot by the SGD update (k_sgd's arithmetic, gradients
// cleared): a column that lands in the gradient arena gr[0, n) updates its weight instead of being stored,
// and workgroups past the reduction jobs update the floats of the arena that no job produces (gradients
// written directly by other kernels).  One launch instead of two at the end of a training step.
__global__ __launch_bounds__(1024) void k_reduce_multi_sgd(const ReduceBatch batch, float *__restrict__ w,
                                                           float *__restrict__ gr, int n, float lr, float inv_div) {
  __shared__ float sh[16][65];
  if ((int)blockIdx.x >= batch.blocks) {
    const int i = ((int)blockIdx.x - batch.blocks) * 1024 + (int)threadIdx.x;
    if (i >= n) return;
    bool produced = false;
#pragma unroll 1
    for (int k = 0; k < batch.n; ++k) {
      const long off = (long)i - (long)(batch.job[k].out - gr);
      produced |= off >= 0 && off < batch.job[k].cols;
    }
    if (!produced) {
      const float gs = __fmul_rn(gr[i], inv_div);
      w[i] = __fsub_rn(w[i], __fmul_rn(gs, lr));
      gr[i] = 0.f;
    }
    return;
  }
  int ji = 0;
#pragma unroll 1
  for (int k = 1; k < batch.n; ++k)
    if ((int)blockIdx.x >= batch.job[k].first_block) ji = k;
  const float *__restrict__ slab = batch.job[ji].slab;
  const long rows = batch.job[ji].rows;
  const int cols = batch.job[ji].cols, ld = batch.job[ji].ld;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = ((int)blockIdx.x - batch.job[ji].first_block) * 64 + cl;
  const long off = (long)(batch.job[ji].out - gr) + c;
  const bool in_arena = off >= 0 && off < n;
  float wv = 0.f;
  if (rl == 0 && c < cols && in_arena) wv = w[off];  // its round trip runs under the reduction
  float s = 0.f;
  if (c < cols) {
    long r = rl;
    for (; r + 7 * 16 < rows; r += 8 * 16) {  // 8 independent loads in flight
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = slab[(r + 16 * k) * ld + c];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; r < rows; r += 16) s += slab[r * ld + c];
  }
  sh[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][cl];
    if (in_arena) {
      const float gs = __fmul_rn(t, inv_div);
      w[off] = __fsub_rn(wv, __fmul_rn(gs, lr));
      batch.job[ji].out[c] = 0.f;
    } else {
      batch.job[ji].out[c] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm backward reduction step (reference layers/batch_normalization_layer.hpp:187-199 and
// 316-326).  From S1 = sum delta, S2 = sum delta*xhat per channel:
//   grad_gamma = S2, grad_beta = S1
//   dvar  = sum (delta*gamma*(x-mean)) * -0.5 * inv^3 = -0.5 * gamma * inv^2 * S2
//   dmean = sum delta*gamma*(-inv)                    = -gamma * inv * S1
//           (the reference's extra term dvar*(-2/R)*sum(x-mean) is rounding noise around 0 and is
//            dropped)
//   delta_in = delta*gamma*inv + (dvar*2/R) * (x[row 0] - mean) + dmean/R
// The middle term uses row 0 of the input for every row: in the reference the product is written
// small[1,C] * big[R,C] and Tensor::operator* keeps the left operand's shape
// (tensor_cpu.hpp:1050-1084).  coef = (scale, shift) with delta_in = delta*scale + shift.
// ---------------------------------------------------------------------------------------------
// One workgroup per channel; a thread takes every FIN_T-th block with eight loads in flight.
// Grouped form (blockIdx.y = statistics group, several independent mini-batches side by side): group g owns blocks
// [g*nbpg, (g+1)*nbpg) and rows [g*rows_group, ...); it takes stat[g][2C], row 0 of ITS rows, writes coef[g][2C] and,
// instead of the gradient arena, gpart[g][2C] = (grad gamma | grad beta) of its own mini-batch; the caller sums those
// over the groups.  gpart == nullptr (one group): the gradients go to ggamma / gbeta.
constexpr int FIN_T = 256;
__global__ __launch_bounds__(FIN_T) void k_bn_bwd_finalize(const float *__restrict__ part, int nb_total, int nbpg, int C,
                                                           long rows_group, long rows_total,
                                                           const float *__restrict__ gamma, const float *__restrict__ stat,
                                                           const float *__restrict__ in_row0, float *__restrict__ ggamma,
                                                           float *__restrict__ gbeta, float *__restrict__ coef,
                                                           float *__restrict__ gpart) {
  __shared__ double sh[2][FIN_T / 64];
  const int tid = threadIdx.x, c = blockIdx.x, grp = blockIdx.y;
  part += (size_t)grp * nbpg * 2 * C;
  stat += (size_t)grp * 2 * C;
  coef += (size_t)grp * 2 * C;
  in_row0 += (size_t)grp * rows_group * C;
  const int nb = min(nbpg, nb_total - grp * nbpg);
  rows_total = min(rows_group, rows_total - (long)grp * rows_group);
  if (gpart) {
    ggamma = gpart + (size_t)grp * 2 * C;
    gbeta = ggamma + C;
  }
  // operands of the closing arithmetic, requested before the partial loads so that their round trip is hidden
  float gm = 0.f, mean = 0.f, inv = 0.f, x0 = 0.f;
  if (tid == 0) {
    gm = gamma[c];
    mean = stat[c];
    inv = stat[C + c];
    x0 = in_row0[c];
  }
  double s1 = 0.0, s2 = 0.0;
  for (int b0 = tid; b0 < nb; b0 += 8 * FIN_T) {
    float a[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = min(b0 + k * FIN_T, nb - 1);
      a[k] = part[(size_t)b * 2 * C + c];
      q[k] = part[(size_t)b * 2 * C + C + c];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (b0 + k * FIN_T < nb) {
        s1 += (double)a[k];
        s2 += (double)q[k];
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if ((tid & 63) == 0) {
    sh[0][tid >> 6] = s1;
    sh[1][tid >> 6] = s2;
  }
  __syncthreads();
  if (tid == 0) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < FIN_T / 64; ++k) {
      t1 += sh[0][k];
      t2 += sh[1][k];
    }
    const float S1 = (float)t1, S2 = (float)t2;
    const float R = (float)rows_total;
    ggamma[c] = S2;
    gbeta[c] = S1;
    const float dvar = -0.5f * gm * inv * inv * S2;
    const float dmean = -gm * inv * S1;
    const float k1 = ((dvar * 2.0f) * (1.0f / R)) * (x0 - mean);
    const float k2 = dmean * (1.0f / R);
    coef[c] = gm * inv;
    coef[C + c] = k1 + k2;
  }
}

// update_weights + SGD::update (reference e.g. layers/dense_layer.hpp:102-112,
// utils/optimizer.hpp:22-28):  g <- g * (1/divisor);  w <- w - g * lr.  Two roundings, as there.
// Floats [hole_first, hole_first + hole_count) are left untouched (weights and gradients): the model's
// deferred attention update owns that range while its gradients are still being computed.
__global__ __launch_bounds__(TPB) void k_sgd(float *__restrict__ w, float *__restrict__ gr, int n, float lr,
                                             float inv_div, int zero_grad, int hole_first, int hole_count) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n && (unsigned)(i - hole_first) >= (unsigned)hole_count) {
    const float gs = __fmul_rn(gr[i], inv_div);
    w[i] = __fsub_rn(w[i], __fmul_rn(gs, lr));
    if (zero_grad) gr[i] = 0.f;
  }
}

// Textbook Adam (Kingma & Ba).  Not part of the reference: extension, parity unpinned.
__global__ __launch_bounds__(TPB) void k_adam(float *__restrict__ w, float *__restrict__ gr, float *__restrict__ m1,
                                              float *__restrict__ m2, int n, float lr, float b1, float b2, float eps,
                                              float c1, float c2, float inv_div) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n) {
    const float gq = gr[i] * inv_div;
    const float a = b1 * m1[i] + (1.f - b1) * gq;
    const float v = b2 * m2[i] + (1.f - b2) * gq * gq;
    m1[i] = a;
    m2[i] = v;
    w[i] -= lr * (a / c1) / (sqrtf(v / c2) + eps);
    gr[i] = 0.f;
  }
}

// Generic DenseLayer::backward over a row matrix (reference layers/dense_layer.hpp:79-100) for
// in, out <= 32: per-workgroup partial gW[in*out] | gb[out], and dX = dY W^T.
__global__ __launch_bounds__(TPB) void k_dense_bwd_rows(const float *__restrict__ X, const float *__restrict__ Wm,
                                                        const float *__restrict__ dY, float *__restrict__ dX,
                                                        int accumulate, float *__restrict__ scratch, long rows, int in,
                                                        int out, int relu_x, int ld) {
  __shared__ float sx[RB][33], sdy[RB][33], sw[32 * 32];
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * RB;
  const int cnt = (int)min((long)RB, rows - r0);
  for (int i = tid; i < RB * in; i += TPB) {
    const int r = i / in, c = i % in;
    float v = (r < cnt && X) ? X[(r0 + r) * in + c] : 0.f;
    if (relu_x) v = v > 0.f ? v : 0.f;
    sx[r][c] = v;
  }
  for (int i = tid; i < RB * out; i += TPB) {
    const int r = i / out, c = i % out;
    sdy[r][c] = r < cnt ? dY[(r0 + r) * out + c] : 0.f;
  }
  for (int i = tid; i < in * out; i += TPB) sw[i] = Wm[i];
  __syncthreads();
  if (scratch) {
    float *row = scratch + (size_t)blockIdx.x * ld;
    for (int idx = tid; idx < in * out; idx += TPB) {
      const int i = idx / out, o = idx % out;
      float s = 0.f;
      for (int r = 0; r < RB; ++r) s += sx[r][i] * sdy[r][o];
      row[idx] = s;
    }
    if (tid < out) {
      float s = 0.f;
      for (int r = 0; r < RB; ++r) s += sdy[r][tid];
      row[in * out + tid] = s;
    }
  }
  if (dX) {
    for (int idx = tid; idx < RB * in; idx += TPB) {
      const int r = idx / in, i = idx % in;
      if (r < cnt) {
        float s = 0.f;
        for (int o = 0; o < out; ++o) s += sdy[r][o] * sw[i * out + o];
        if (accumulate) dX[(r0 + r) * in + i] += s;
        else dX[(r0 + r) * in + i] = s;
      }
    }
  }
}

// Bandwidth yardstick with the access pattern of the per-image kernels: `blocks` workgroups, each
// streaming its own contiguous region of `floats_per_block` floats (mode 0: one region per workgroup,
// what a one-workgroup-per-image kernel does; mode 1: the same bytes, but trip k of workgroup b moves
// the 4 KB chunk k*blocks + b, so that the chunks in flight at any moment are neighbours in memory;
// modes 2, 3: mode 0 with nontemporal loads and stores / nontemporal stores only).
__global__ __launch_bounds__(TPB) void k_probe_copy(const float4 *__restrict__ src, float4 *__restrict__ dst,
                                                    long vec_per_block, int mode) {
  const long nb = gridDim.x, b = blockIdx.x;
  if (mode == 0) {
    const float4 *s = src + b * vec_per_block;
    float4 *d = dst + b * vec_per_block;
    for (long i = threadIdx.x; i < vec_per_block; i += TPB) d[i] = s[i];
  } else if (mode >= 2) {
    typedef float fv4 __attribute__((ext_vector_type(4)));
    const fv4 *s = reinterpret_cast<const fv4 *>(src) + b * vec_per_block;
    fv4 *d = reinterpret_cast<fv4 *>(dst) + b * vec_per_block;
    for (long i = threadIdx.x; i < vec_per_block; i += TPB) {
      const fv4 v = mode == 2 ? __builtin_nontemporal_load(s + i) : s[i];
      __builtin_nontemporal_store(v, d + i);
    }
  } else {
    const long trips = (vec_per_block + TPB - 1) / TPB, total = nb * vec_per_block;
    for (long k = 0; k < trips; ++k) {
      const long i = (k * nb + b) * TPB + threadIdx.x;
      if (i < total) dst[i] = src[i];
    }
  }
}

}  // namespace vt

// ---------------------------------------------------------------------------------------------
// extern "C" launchers
// ---------------------------------------------------------------------------------------------
using namespace vt;

static thread_local ReduceBatch *g_reduce_batch = nullptr;

void vt::reduce_batch_begin(ReduceBatch *b) {
  b->n = 0;
  b->blocks = 0;
  g_reduce_batch = b;
}
void vt::reduce_batch_abandon() { g_reduce_batch = nullptr; }
int vt::reduce_batch_flush(hipStream_t s) {
  ReduceBatch *b = g_reduce_batch;
  g_reduce_batch = nullptr;
  if (!b || b->n == 0) return 0;
  hipLaunchKernelGGL(k_reduce_multi, dim3(b->blocks), dim3(1024), 0, s, *b);
  VT_CHECK_LAUNCH();
  return 0;
}
int vt::reduce_batch_flush_sgd(float *w, float *g, int n, float lr, float divisor, hipStream_t s) {
  ReduceBatch *b = g_reduce_batch;
  g_reduce_batch = nullptr;
  if (n < 1 || divisor == 0.f) return VT_EINVAL;
  static const ReduceBatch none{};
  const ReduceBatch &rb = b ? *b : none;
  hipLaunchKernelGGL(k_reduce_multi_sgd, dim3(rb.blocks + (n + 1023) / 1024), dim3(1024), 0, s, rb, w, g, n, lr, 1.0f / divisor);
  VT_CHECK_LAUNCH();
  return 0;
}
int vt::reduce_rows(const float *slab, long rows, int cols, int ld, float *out, hipStream_t s) {
  ReduceBatch *b = g_reduce_batch;
  if (b && b->n < MAX_REDUCE_JOBS) {
    b->job[b->n++] = ReduceJob{slab, out, rows, cols, ld, b->blocks};
    b->blocks += (cols + 63) / 64;
    return 0;
  }
  hipLaunchKernelGGL(k_reduce_rows, dim3((cols + 63) / 64), dim3(1024), 0, s, slab, rows, cols, ld, out);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_reduce_rows(const float *slab, long rows, int cols, int ld, float *out, vt_stream s) {
  if (rows < 1 || cols < 1 || ld < cols) return VT_EINVAL;
  return reduce_rows(slab, rows, cols, ld, out, (hipStream_t)s);
}

extern "C" int vt_bn_bwd_finalize(const float *part, int nblocks, int C, long rows_total, const float *gamma,
                                  const float *stat, const float *in_row0, float *ggamma, float *gbeta, float *coef,
                                  vt_stream s) {
  if (C < 1 || C > 1024 || nblocks < 1 || rows_total < 1) return VT_EUNSUPPORTED;
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(C, 1), dim3(FIN_T), 0, (hipStream_t)s, part, nblocks, nblocks, C, rows_total,
                     rows_total, gamma, stat, in_row0, ggamma, gbeta, coef, (float *)nullptr);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_bn_bwd_finalize_grouped(const float *part, int nblocks, int blocks_per_group, int C, long rows_per_group,
                                          long rows_total, const float *gamma, const float *stat, const float *in,
                                          float *gpart, float *coef, vt_stream s) {
  if (C < 1 || C > 1024 || nblocks < 1 || blocks_per_group < 1 || rows_per_group < 1 || rows_total < 1 || !gpart)
    return VT_EUNSUPPORTED;
  const int groups = (nblocks + blocks_per_group - 1) / blocks_per_group;
  if ((long)(groups - 1) * rows_per_group >= rows_total) return VT_EINVAL;
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(C, groups), dim3(FIN_T), 0, (hipStream_t)s, part, nblocks, blocks_per_group, C,
                     rows_per_group, rows_total, gamma, stat, in, (float *)nullptr, (float *)nullptr, coef, gpart);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_probe_copy(const float *src, float *dst, int blocks, long floats_per_block, int mode, vt_stream s) {
  if (!src || !dst || blocks < 1 || floats_per_block < 4 || (floats_per_block & 3) || mode < 0 || mode > 3) return VT_EINVAL;
  hipLaunchKernelGGL(k_probe_copy, dim3(blocks), dim3(TPB), 0, (hipStream_t)s, reinterpret_cast<const float4 *>(src),
                     reinterpret_cast<float4 *>(dst), floats_per_block / 4, mode);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_sgd_update(float *w, float *g, int n, float lr, float divisor, int zero_grad, vt_stream s) {
  if (n < 1 || divisor == 0.f) return VT_EINVAL;
  hipLaunchKernelGGL(k_sgd, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)s, w, g, n, lr, 1.0f / divisor,
                     zero_grad, 0, 0);
  VT_CHECK_LAUNCH();
  return 0;
}

int vt::sgd_update_hole(float *w, float *g, int n, float lr, float divisor, int hole_first, int hole_count, hipStream_t s) {
  if (n < 1 || divisor == 0.f || hole_first < 0 || hole_count < 0) return VT_EINVAL;
  hipLaunchKernelGGL(k_sgd, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, s, w, g, n, lr, 1.0f / divisor, 1, hole_first,
                     hole_count);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_adam_update(float *w, float *g, float *m1, float *m2, int n, float lr, float beta1, float beta2,
                              float eps, int step, float divisor, vt_stream s) {
  if (n < 1 || step < 1 || divisor == 0.f) return VT_EINVAL;
  const float c1 = 1.f - powf(beta1, (float)step), c2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(k_adam, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)s, w, g, m1, m2, n, lr, beta1, beta2,
                     eps, c1, c2, 1.0f / divisor);
  VT_CHECK_LAUNCH();
  return 0;
}

// scratch >= ceil(rows/64) * (in*out + out) floats when gW is requested.
extern "C" int vt_dense_bwd_rows(const float *X, const float *W, const float *dY, float *gW, float *gb, float *dX,
                                 int accumulate_dx, float *scratch, long rows, int in, int out, int relu_x,
                                 vt_stream s) {
  if (rows < 1 || in < 1 || in > 32 || out < 1 || out > 32) return VT_EUNSUPPORTED;
  if (gW && (!scratch || !X)) return VT_EINVAL;
  const int nb = (int)((rows + RB - 1) / RB), ld = in * out + out;
  hipLaunchKernelGGL(k_dense_bwd_rows, dim3(nb), dim3(TPB), 0, (hipStream_t)s, X, W, dY, dX, accumulate_dx,
                     gW ? scratch : nullptr, rows, in, out, relu_x, ld);
  VT_CHECK_LAUNCH();
  if (gW) {
    int rc = reduce_rows(scratch, nb, in * out, ld, gW, (hipStream_t)s);
    if (rc) return rc;
    if (gb) rc = reduce_rows(scratch + in * out, nb, out, ld, gb, (hipStream_t)s);
    if (rc) return rc;
  }
  return 0;
}
