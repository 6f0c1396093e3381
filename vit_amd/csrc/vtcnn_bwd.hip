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
constexpr int FIN_T = 1024;
__global__ __launch_bounds__(FIN_T) void k_bn_bwd_finalize(const float *__restrict__ part, int nb, int C, long rows_total,
                                                           const float *__restrict__ gamma, const float *__restrict__ stat,
                                                           const float *__restrict__ in_row0, float *__restrict__ ggamma,
                                                           float *__restrict__ gbeta, float *__restrict__ coef) {
  __shared__ double sh[FIN_T];
  const int tid = threadIdx.x;
  const int C2x = 2 * C;  // (S1, S2) pairs
  const int c = tid % C2x, j = tid / C2x, ns = FIN_T / C2x;
  double s = 0.0;
  int b = j;
  for (; b + 3 * ns < nb; b += 4 * ns) {  // 4 independent loads in flight
    const float v0 = part[(size_t)b * C2x + c], v1 = part[(size_t)(b + ns) * C2x + c];
    const float v2 = part[(size_t)(b + 2 * ns) * C2x + c], v3 = part[(size_t)(b + 3 * ns) * C2x + c];
    s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
  }
  for (; b < nb; b += ns) s += (double)part[(size_t)b * C2x + c];
  sh[tid] = s;
  __syncthreads();
  if (tid < C) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < ns; ++k) {
      s1 += sh[k * C2x + tid];
      s2 += sh[k * C2x + C + tid];
    }
    const float S1 = (float)s1, S2 = (float)s2;
    const float gm = gamma[tid], mean = stat[tid], inv = stat[C + tid];
    const float R = (float)rows_total;
    ggamma[tid] = S2;
    gbeta[tid] = S1;
    const float dvar = -0.5f * gm * inv * inv * S2;
    const float dmean = -gm * inv * S1;
    const float k1 = ((dvar * 2.0f) * (1.0f / R)) * (in_row0[tid] - mean);
    const float k2 = dmean * (1.0f / R);
    coef[tid] = gm * inv;
    coef[C + tid] = k1 + k2;
  }
}

// ---------------------------------------------------------------------------------------------
// ProjectorLayer::backward, token side (reference projector_layer.hpp:186-202): Dense backward of
// linear3 (values) and linear2 (keys) over the N*L token rows.
//   gWv = o^T dv, gbv = sum dv, gWk = o^T dk, token delta = dv.Wv^T + dk.Wk^T
// Per-workgroup partials: gWk[D*D] | gWv[D*C2] | gbv[C2]  (PT_STRIDE floats), summed afterwards.
// ---------------------------------------------------------------------------------------------
constexpr int PT_STRIDE = D * D + D * C2 + C2;  // 1552
__global__ __launch_bounds__(TPB) void k_projtok_bwd(const float *__restrict__ o, const float *__restrict__ dv,
                                                     const float *__restrict__ dk, const float *__restrict__ Wv,
                                                     const float *__restrict__ Wk, float *__restrict__ pdt,
                                                     float *__restrict__ scratch, long rows) {
  __shared__ float so[RB][D + 1], sdk[RB][D + 1], sdv[RB][C2 + 1];
  __shared__ float sWk[D * D], sWv[D * C2];
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * RB;
  const int cnt = (int)min((long)RB, rows - r0);
  for (int i = tid; i < RB * D; i += TPB) {
    const int r = i / D, d = i % D;
    const bool ok = r < cnt;
    so[r][d] = ok ? o[(r0 + r) * D + d] : 0.f;
    sdk[r][d] = ok ? dk[(r0 + r) * D + d] : 0.f;
  }
  for (int i = tid; i < RB * C2; i += TPB) {
    const int r = i / C2, c = i % C2;
    sdv[r][c] = r < cnt ? dv[(r0 + r) * C2 + c] : 0.f;
  }
  for (int i = tid; i < D * D; i += TPB) sWk[i] = Wk[i];
  for (int i = tid; i < D * C2; i += TPB) sWv[i] = Wv[i];
  __syncthreads();
  float *row = scratch + (size_t)blockIdx.x * PT_STRIDE;
  for (int idx = tid; idx < D * D; idx += TPB) {
    const int d = idx / D, e = idx % D;
    float s = 0.f;
    for (int r = 0; r < RB; ++r) s += so[r][d] * sdk[r][e];
    row[idx] = s;
  }
  for (int idx = tid; idx < D * C2; idx += TPB) {
    const int d = idx / C2, c = idx % C2;
    float s = 0.f;
    for (int r = 0; r < RB; ++r) s += so[r][d] * sdv[r][c];
    row[D * D + idx] = s;
  }
  if (tid < C2) {
    float s = 0.f;
    for (int r = 0; r < RB; ++r) s += sdv[r][tid];
    row[D * D + D * C2 + tid] = s;
  }
  for (int idx = tid; idx < RB * D; idx += TPB) {
    const int r = idx / D, d = idx % D;
    if (r < cnt) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < C2; ++c) s1 += sdv[r][c] * sWv[d * C2 + c];
#pragma unroll 8
      for (int e = 0; e < D; ++e) s2 += sdk[r][e] * sWk[d * D + e];
      pdt[(r0 + r) * D + d] = s1 + s2;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// TransformerLayer::backward, step A (reference layers/transformer_layer.hpp:142-161):
//   d0 = last_output - targets  with targets = projector token delta (visual_transformer.hpp:156-157)
//   partial sums of the second BatchNorm1D backward.
// Also the tokenizer's linear2 input delta dT0 = token_delta . W2tok^T (tokenizer.hpp:73-75), which
// reads the same rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_tr_bwd_a(const float *__restrict__ o, const float *__restrict__ pdt,
                                                  const float *__restrict__ s2, const float *__restrict__ st2,
                                                  const float *__restrict__ W2tok, float *__restrict__ part,
                                                  float *__restrict__ dT0, long rows) {
  __shared__ float sd[RB][D + 1], sxh[RB][D + 1], sp[RB][D + 1];
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * RB;
  const int cnt = (int)min((long)RB, rows - r0);
  for (int i = tid; i < RB * D; i += TPB) {
    const int r = i / D, d = i % D;
    const bool ok = r < cnt;
    const float pv = ok ? pdt[(r0 + r) * D + d] : 0.f;
    sp[r][d] = pv;
    sd[r][d] = ok ? o[(r0 + r) * D + d] - pv : 0.f;
    sxh[r][d] = ok ? (s2[(r0 + r) * D + d] - st2[d]) * st2[D + d] : 0.f;
  }
  __syncthreads();
  if (tid < 2 * D) {
    const int d = tid % D;
    float s = 0.f;
    if (tid < D) for (int r = 0; r < RB; ++r) s += sd[r][d];
    else for (int r = 0; r < RB; ++r) s += sd[r][d] * sxh[r][d];
    part[(size_t)blockIdx.x * 2 * D + tid] = s;
  }
  for (int idx = tid; idx < RB * C1; idx += TPB) {
    const int r = idx / C1, c = idx % C1;
    if (r < cnt) {
      float s = 0.f;
#pragma unroll 8
      for (int d = 0; d < D; ++d) s += sp[r][d] * W2tok[c * D + d];
      dT0[(r0 + r) * C1 + c] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// TransformerLayer::backward, step B (reference transformer_layer.hpp:161-189):
//   ds2 = BN2.backward(d0);  linear2 backward (its cached input is relu(hdn));
//   linear1 backward WITHOUT the ReLU derivative (:174-181);  da = dh.W1^T + ds2;
//   partial sums of the first BatchNorm1D backward.
// Per-workgroup partials: gW1[D*D] | gb1[D] | gW2[D*D] | gb2[D]  (FF_STRIDE, arena order).
// ---------------------------------------------------------------------------------------------
constexpr int FF_STRIDE = 2 * (D * D + D);  // 2112
__global__ __launch_bounds__(TPB) void k_tr_bwd_b(const float *__restrict__ o, const float *__restrict__ pdt,
                                                  const float *__restrict__ coef2, const float *__restrict__ hdn,
                                                  const float *__restrict__ a, const float *__restrict__ tok,
                                                  const float *__restrict__ st1, const float *__restrict__ W1,
                                                  const float *__restrict__ W2, float *__restrict__ ds2_out,
                                                  float *__restrict__ da_out, float *__restrict__ scratch,
                                                  float *__restrict__ part, long rows) {
  __shared__ float sds2[RB][D + 1], srl[RB][D + 1], sa[RB][D + 1], sdh[RB][D + 1];
  __shared__ float sW1[D * D], sW2[D * D];
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * RB;
  const int cnt = (int)min((long)RB, rows - r0);
  for (int i = tid; i < RB * D; i += TPB) {
    const int r = i / D, d = i % D;
    const bool ok = r < cnt;
    const float d0 = ok ? o[(r0 + r) * D + d] - pdt[(r0 + r) * D + d] : 0.f;
    const float v = ok ? d0 * coef2[d] + coef2[D + d] : 0.f;
    sds2[r][d] = v;
    if (ok) ds2_out[(r0 + r) * D + d] = v;
    const float hv = ok ? hdn[(r0 + r) * D + d] : 0.f;
    srl[r][d] = hv > 0.f ? hv : 0.f;
    sa[r][d] = ok ? a[(r0 + r) * D + d] : 0.f;
  }
  for (int i = tid; i < D * D; i += TPB) {
    sW1[i] = W1[i];
    sW2[i] = W2[i];
  }
  __syncthreads();
  float *row = scratch + (size_t)blockIdx.x * FF_STRIDE;
  // dh = ds2 . W2^T
  for (int idx = tid; idx < RB * D; idx += TPB) {
    const int r = idx / D, i = idx % D;
    float s = 0.f;
#pragma unroll 8
    for (int e = 0; e < D; ++e) s += sds2[r][e] * sW2[i * D + e];
    sdh[r][i] = s;
  }
  // gW2 = relu(hdn)^T ds2 ; gb2 = sum ds2
  for (int idx = tid; idx < D * D; idx += TPB) {
    const int i = idx / D, e = idx % D;
    float s = 0.f;
    for (int r = 0; r < RB; ++r) s += srl[r][i] * sds2[r][e];
    row[D * D + D + idx] = s;
  }
  if (tid < D) {
    float s = 0.f;
    for (int r = 0; r < RB; ++r) s += sds2[r][tid];
    row[2 * D * D + D + tid] = s;
  }
  __syncthreads();
  // gW1 = a^T dh ; gb1 = sum dh
  for (int idx = tid; idx < D * D; idx += TPB) {
    const int i = idx / D, e = idx % D;
    float s = 0.f;
    for (int r = 0; r < RB; ++r) s += sa[r][i] * sdh[r][e];
    row[idx] = s;
  }
  if (tid < D) {
    float s = 0.f;
    for (int r = 0; r < RB; ++r) s += sdh[r][tid];
    row[D * D + tid] = s;
  }
  // da = dh . W1^T + ds2   (reuse srl for da)
  __syncthreads();
  for (int idx = tid; idx < RB * D; idx += TPB) {
    const int r = idx / D, i = idx % D;
    float s = 0.f;
#pragma unroll 8
    for (int e = 0; e < D; ++e) s += sdh[r][e] * sW1[i * D + e];
    s += sds2[r][i];
    srl[r][i] = (r < cnt) ? s : 0.f;
    if (r < cnt) da_out[(r0 + r) * D + i] = s;
  }
  __syncthreads();
  if (tid < 2 * D) {
    const int d = tid % D;
    float s = 0.f;
    if (tid < D) {
      for (int r = 0; r < cnt; ++r) s += srl[r][d];
    } else {
      for (int r = 0; r < cnt; ++r) s += srl[r][d] * ((tok[(r0 + r) * D + d] - st1[d]) * st1[D + d]);
    }
    part[(size_t)blockIdx.x * 2 * D + tid] = s;
  }
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
                                                  float *__restrict__ dc, float *__restrict__ qkv, long rows) {
  __shared__ float sdr[RB][D + 1], sx[RB][D + 1];
  __shared__ float sWo[D * D], sW[3 * D * D];  // sW[m][i][h*HD+e] = W_{m,h}[i][e]
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * RB;
  const int cnt = (int)min((long)RB, rows - r0);
  for (int i = tid; i < RB * D; i += TPB) {
    const int r = i / D, d = i % D;
    const bool ok = r < cnt;
    const float v = ok ? da[(r0 + r) * D + d] * coef1[d] + coef1[D + d] : 0.f;
    sdr[r][d] = v;
    if (ok) dr_out[(r0 + r) * D + d] = v;
    sx[r][d] = ok ? tok[(r0 + r) * D + d] : 0.f;
  }
  for (int i = tid; i < D * D; i += TPB) sWo[i] = Wo[i];
  // arena order: q0,k0,v0,q1,k1,v1, each [D][HD]
  for (int i = tid; i < 3 * D * D; i += TPB) {
    const int m = i / (D * D), rem = i % (D * D), ii = rem / D, col = rem % D, h = col / HD, e = col % HD;
    sW[i] = Wqkv[(size_t)(h * 3 + m) * D * HD + ii * HD + e];
  }
  __syncthreads();
  for (int idx = tid; idx < RB * D; idx += TPB) {
    const int r = idx / D, jx = idx % D;
    if (r < cnt) {
      float s = 0.f;
#pragma unroll 8
      for (int i = 0; i < D; ++i) s += sdr[r][i] * sWo[jx * D + i];
      dc[(r0 + r) * D + jx] = s;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        float t = 0.f;
#pragma unroll 8
        for (int i = 0; i < D; ++i) t += sx[r][i] * sW[m * D * D + i * D + jx];
        qkv[(size_t)m * rows * D + (r0 + r) * D + jx] = t;
      }
    }
  }
}

// update_weights + SGD::update (reference e.g. layers/dense_layer.hpp:102-112,
// utils/optimizer.hpp:22-28):  g <- g * (1/divisor);  w <- w - g * lr.  Two roundings, as there.
__global__ __launch_bounds__(TPB) void k_sgd(float *__restrict__ w, float *__restrict__ gr, int n, float lr,
                                             float inv_div, int zero_grad) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n) {
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
  if (C < 1 || FIN_T % (2 * C) != 0 || nblocks < 1 || rows_total < 1) return VT_EUNSUPPORTED;
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(1), dim3(FIN_T), 0, (hipStream_t)s, part, nblocks, C, rows_total, gamma,
                     stat, in_row0, ggamma, gbeta, coef);
  VT_CHECK_LAUNCH();
  return 0;
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

// g_ffn: gradient block [gW1 | gb1 | gW2 | gb2] (2112 floats, arena order); scratch >= nb*2112.
extern "C" int vt_transformer_bwd_b(const float *o, const float *pdt, const float *n2_coef, const float *hdn,
                                    const float *a, const float *tok, const float *n1_stat, const float *w1,
                                    const float *w2, float *ds2, float *da, float *g_ffn, float *scratch, float *part,
                                    long rows, vt_stream s) {
  if (rows < 1) return VT_EINVAL;
  const int nb = (int)((rows + RB - 1) / RB);
  hipLaunchKernelGGL(k_tr_bwd_b, dim3(nb), dim3(TPB), 0, (hipStream_t)s, o, pdt, n2_coef, hdn, a, tok, n1_stat, w1, w2,
                     ds2, da, scratch, part, rows);
  VT_CHECK_LAUNCH();
  return reduce_rows(scratch, nb, FF_STRIDE, FF_STRIDE, g_ffn, (hipStream_t)s);
}

extern "C" int vt_transformer_bwd_c(const float *da, const float *n1_coef, const float *wo, const float *tok,
                                    const float *wqkv, float *dr, float *dc, float *qkv, long rows, vt_stream s) {
  if (rows < 1) return VT_EINVAL;
  const int nb = (int)((rows + RB - 1) / RB);
  hipLaunchKernelGGL(k_tr_bwd_c, dim3(nb), dim3(TPB), 0, (hipStream_t)s, da, n1_coef, wo, tok, wqkv, dr, dc, qkv, rows);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_sgd_update(float *w, float *g, int n, float lr, float divisor, int zero_grad, vt_stream s) {
  if (n < 1 || divisor == 0.f) return VT_EINVAL;
  hipLaunchKernelGGL(k_sgd, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)s, w, g, n, lr, 1.0f / divisor,
                     zero_grad);
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
