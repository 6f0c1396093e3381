// Forward kernels of the VTCNN hot path for gfx950 (MI355X).
//
// Structure: the forward pass is cut at the four mini-batch-wide BatchNorm statistics (after the
// conv, before/inside the transformer FFN, after the projector).  Between two cuts all work is
// per sample (one 256-thread workgroup per image: 784 pixels = 3.06 pixels per lane) or per
// token row; per-sample operands that every lane needs (folded projector matrices, weights) are
// read through wave-uniform addresses so they live in SGPRs, not in LDS.
// HBM layout: feature maps [N,P,C] channel-innermost (one 32-byte / 64-byte record per pixel,
// fully coalesced per wave), token matrices [N*L, D].
#include "vtcnn_common.h"

namespace vt {

// ---------------------------------------------------------------------------------------------
// Conv2DLayer::forward 3x3/s1/p1 (reference layers/conv2d_layer.hpp:35-87) + per-sample partial
// statistics for BatchNorm2D (reference layers/batch_normalization_layer.hpp:274-276).
// One workgroup per image; the zero-padded image tile is staged in LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_conv_fwd(const float *__restrict__ X, const float *__restrict__ Wc,
                                                  const float *__restrict__ bc, float *__restrict__ conv,
                                                  float *__restrict__ part, int Ci, int H, int W) {
  extern __shared__ float sm[];
  const int P = H * W, TH = H + 2, TW = W + 2;
  const int tile_n = Ci * TH * TW;
  float *tile = sm;
  float *red = sm + ((tile_n + 3) & ~3);
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < tile_n; i += TPB) {
    const int ci = i / (TH * TW), r = i % (TH * TW), hh = r / TW - 1, ww = r % TW - 1;
    float v = 0.f;
    if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = X[((size_t)(n * Ci + ci) * H + hh) * W + ww];
    tile[i] = v;
  }
  __syncthreads();
  float vals[PPT][C1];
  float s[C1];
#pragma unroll
  for (int co = 0; co < C1; ++co) s[co] = 0.f;
#pragma unroll
  for (int it = 0; it < PPT; ++it) {
    const int p = tid + it * TPB;
    const bool ok = p < P;
    const int h = ok ? p / W : 0, w = ok ? p % W : 0;
    float acc[C1];
#pragma unroll
    for (int co = 0; co < C1; ++co) acc[co] = bc[co];
    for (int ci = 0; ci < Ci; ++ci) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float v = tile[(ci * TH + h + kh) * TW + w + kw];
#pragma unroll
          for (int co = 0; co < C1; ++co) acc[co] += v * Wc[((co * Ci + ci) * 3 + kh) * 3 + kw];
        }
    }
#pragma unroll
    for (int co = 0; co < C1; ++co) {
      vals[it][co] = ok ? acc[co] : 0.f;
      s[co] += vals[it][co];
    }
    if (ok) {
      float4 *dst = reinterpret_cast<float4 *>(conv + ((size_t)n * P + p) * C1);
      dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
  }
  if (part == nullptr) return;
  block_sum<C1>(s, red);
  float m2[C1];
  const float invP = 1.0f / (float)P;
#pragma unroll
  for (int co = 0; co < C1; ++co) {
    const float mu = s[co] * invP;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < PPT; ++it) {
      const float dlt = vals[it][co] - mu;
      q += (tid + it * TPB < P) ? dlt * dlt : 0.f;
    }
    m2[co] = q;
  }
  block_sum<C1>(m2, red);
  if (tid < C1) {
    part[(size_t)n * 2 * C1 + tid] = s[tid];
    part[(size_t)n * 2 * C1 + C1 + tid] = m2[tid];
  }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm statistics finalisation (reference batch_normalization_layer.hpp:152-159, 274-283):
// merges per-block (sum, centred M2) partials with Chan's formula in double precision, writes
// stat = (mean, 1/sqrt(var+eps)) and updates running = 0.1*batch + 0.9*running (biased var).
// One workgroup.
// ---------------------------------------------------------------------------------------------
constexpr int FIN_T = 1024;  // threads of the single finalisation workgroup
__global__ __launch_bounds__(FIN_T) void k_bn_stats_finalize(const float *__restrict__ part, int nb, int C,
                                                             long rows_per_block, long rows_total,
                                                             float *__restrict__ run_mean, float *__restrict__ run_var,
                                                             float *__restrict__ stat) {
  // One pass: S = sum_b sum_b, Q = sum_b M2_b, U = sum_b sum_b^2/cnt_b; then
  //   mean = S/R,  M2 = Q + U - S^2/R   (Chan's merge summed over all blocks; doubles absorb the
  //   cancellation).  Thread = (channel, slice of blocks); independent loads are unrolled so the
  //   global-load latency is paid once per 4 blocks, not once per block.
  __shared__ double sh[3][FIN_T];
  const int tid = threadIdx.x;
  const int c = tid % C, j = tid / C, ns = FIN_T / C;
  const long last_cnt = rows_total - (long)(nb - 1) * rows_per_block;
  double S = 0.0, Q = 0.0, U = 0.0;
  int b = j;
  for (; b + 3 * ns < nb; b += 4 * ns) {
    float sv[4], qv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sv[k] = part[(size_t)(b + k * ns) * 2 * C + c];
      qv[k] = part[(size_t)(b + k * ns) * 2 * C + C + c];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double cnt = (double)((b + k * ns == nb - 1) ? last_cnt : rows_per_block);
      S += (double)sv[k];
      Q += (double)qv[k];
      U += (double)sv[k] * (double)sv[k] / cnt;
    }
  }
  for (; b < nb; b += ns) {
    const double sv = (double)part[(size_t)b * 2 * C + c], qv = (double)part[(size_t)b * 2 * C + C + c];
    const double cnt = (double)((b == nb - 1) ? last_cnt : rows_per_block);
    S += sv;
    Q += qv;
    U += sv * sv / cnt;
  }
  sh[0][tid] = S;
  sh[1][tid] = Q;
  sh[2][tid] = U;
  __syncthreads();
  if (tid < C) {
    double s = 0.0, q = 0.0, u = 0.0;
    for (int k = 0; k < ns; ++k) {
      s += sh[0][k * C + tid];
      q += sh[1][k * C + tid];
      u += sh[2][k * C + tid];
    }
    const double R = (double)rows_total;
    const double mu = s / R;
    double m2 = q + u - s * s / R;
    if (m2 < 0.0) m2 = 0.0;
    const float mean = (float)mu;
    const float var = (float)(m2 / R);
    stat[tid] = mean;
    stat[C + tid] = 1.0f / sqrtf(var + BN_EPS);
    run_mean[tid] = mean * BN_MOM + run_mean[tid] * (1 - BN_MOM);
    run_var[tid] = var * BN_MOM + run_var[tid] * (1 - BN_MOM);
  }
}

// Eval-mode statistics from the running buffers (batch_normalization_layer.hpp:287-291).
__global__ void k_bn_stats_eval(const float *__restrict__ run_mean, const float *__restrict__ run_var, int C,
                                float *__restrict__ stat) {
  const int c = threadIdx.x;
  if (c < C) {
    stat[c] = run_mean[c];
    stat[C + c] = 1.0f / sqrtf(run_var[c] + BN_EPS);
  }
}

// ---------------------------------------------------------------------------------------------
// TransformerLayer::forward (reference layers/transformer_layer.hpp:43-120) over the flat
// [N*L, D] token matrix (Tensor::permute does not move data, tensor_cpu.hpp:719-747).
// The attention branch is identically zero in the reference (Tensor::concat returns zeros,
// tensor_cpu.hpp:251; no bias on the output projection), so residual = tokens.
//   a = BN1(tok);  hdn = a.W1 + b1;  s2 = a + relu(hdn).W2 + b2   (+ partial stats of s2)
// RB rows per workgroup; thread = (row, group of 8 columns); weights in LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_tr_ffn_fwd(const float *__restrict__ tok, const float *__restrict__ st,
                                                    const float *__restrict__ g, const float *__restrict__ b,
                                                    const float *__restrict__ W1, const float *__restrict__ b1,
                                                    const float *__restrict__ W2, const float *__restrict__ b2,
                                                    float *__restrict__ a_out, float *__restrict__ hdn_out,
                                                    float *__restrict__ s2_out, float *__restrict__ part_n2,
                                                    long rows) {
  __shared__ float sW1[D * D], sW2[D * D];
  __shared__ float sa[RB][D + 1], sr[RB][D + 1];
  const int tid = threadIdx.x, r = tid >> 2, cg = (tid & 3) * 8;
  const long row = (long)blockIdx.x * RB + r;
  const bool ok = row < rows;
  for (int i = tid; i < D * D; i += TPB) {
    sW1[i] = W1[i];
    sW2[i] = W2[i];
  }
  float av[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg + j;
    const float tv = ok ? tok[row * D + c] : 0.f;
    av[j] = ((tv - st[c]) * st[D + c]) * g[c] + b[c];
    sa[r][c] = av[j];
    if (ok) a_out[row * D + c] = av[j];
  }
  __syncthreads();
  float h[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) h[j] = 0.f;
#pragma unroll 8
  for (int i = 0; i < D; ++i) {
    const float ai = sa[r][i];
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] += ai * sW1[i * D + cg + j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    h[j] += b1[cg + j];
    if (ok) hdn_out[row * D + cg + j] = h[j];
    sr[r][cg + j] = h[j] > 0.f ? h[j] : 0.f;
  }
  __syncthreads();
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll 8
  for (int i = 0; i < D; ++i) {
    const float ri = sr[r][i];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] += ri * sW2[i * D + cg + j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = av[j] + (o[j] + b2[cg + j]);
    if (ok) s2_out[row * D + cg + j] = v;
    sa[r][cg + j] = v;
  }
  __syncthreads();
  if (tid < D && part_n2 != nullptr) {
    const long left = rows - (long)blockIdx.x * RB;
    const int cnt = left < RB ? (int)left : RB;
    float s = 0.f;
    for (int rr = 0; rr < cnt; ++rr) s += sa[rr][tid];
    const float mu = s / (float)cnt;
    float q = 0.f;
    for (int rr = 0; rr < cnt; ++rr) {
      const float dlt = sa[rr][tid] - mu;
      q += dlt * dlt;
    }
    part_n2[(size_t)blockIdx.x * 2 * D + tid] = s;
    part_n2[(size_t)blockIdx.x * 2 * D + D + tid] = q;
  }
}

// ---------------------------------------------------------------------------------------------
// Transformer output o = BN2(s2) and the token side of ProjectorLayer::forward
// (reference layers/projector_layer.hpp:72-88): k = o.Wk, v = o.Wv + bv, plus the folded score
// matrix Mq[c][l] = sum_d Wq[c][d] k[l][d], so that the per-pixel score x.Wq.k^T becomes x.Mq
// (8 MACs per token instead of 8*32 + 32).  One workgroup per image, everything in LDS.
// pk record per image: Mq[C1][L] | v[L][C2] | k[L][D].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_proj_tokens_fwd(const float *__restrict__ s2, const float *__restrict__ st,
                                                         const float *__restrict__ g, const float *__restrict__ b,
                                                         const float *__restrict__ Wq, const float *__restrict__ Wk,
                                                         const float *__restrict__ Wv, const float *__restrict__ bv,
                                                         float *__restrict__ o_out, float *__restrict__ pk) {
  __shared__ float so[L][D + 1], sk[L][D + 1];
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < L * D; i += TPB) {
    const int l = i / D, d = i % D;
    const float v = ((s2[(size_t)n * L * D + i] - st[d]) * st[D + d]) * g[d] + b[d];
    so[l][d] = v;
    o_out[(size_t)n * L * D + i] = v;
  }
  __syncthreads();
  float *rec = pk + (size_t)n * PK_STRIDE;
  for (int i = tid; i < L * D; i += TPB) {
    const int l = i / D, e = i % D;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < D; ++d) s += so[l][d] * Wk[d * D + e];
    sk[l][e] = s;
    rec[C1 * L + L * C2 + i] = s;
  }
  for (int i = tid; i < L * C2; i += TPB) {
    const int l = i / C2, c = i % C2;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < D; ++d) s += so[l][d] * Wv[d * C2 + c];
    rec[C1 * L + i] = s + bv[c];
  }
  __syncthreads();
  for (int i = tid; i < C1 * L; i += TPB) {
    const int c = i / L, l = i % L;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < D; ++d) s += Wq[c * D + d] * sk[l][d];
    rec[i] = s;
  }
}

// Tensor::argmax_batch (first maximum, tensor_cpu.hpp:1294-1307) and the cross-entropy terms of
// Tensor::compute_loss (tensor_cpu.hpp:1155-1173).  One workgroup; fixed order => reproducible.
__global__ __launch_bounds__(TPB) void k_metrics(const float *__restrict__ probs, const int *__restrict__ labels, int N,
                                                 int K, float *__restrict__ loss_sum, int *__restrict__ argmax) {
  __shared__ float red[NW];
  float part[1] = {0.f};
  for (int n = threadIdx.x; n < N; n += TPB) {
    int best = 0;
    float bv = probs[(size_t)n * K];
    for (int k = 1; k < K; ++k) {
      const float v = probs[(size_t)n * K + k];
      if (v > bv) {
        bv = v;
        best = k;
      }
    }
    if (argmax) argmax[n] = best;
    if (labels) part[0] += -logf(probs[(size_t)n * K + labels[n]] + 1e-8f);
  }
  block_sum<1>(part, red);
  if (threadIdx.x == 0 && loss_sum) loss_sum[0] = part[0];
}

}  // namespace vt

// ---------------------------------------------------------------------------------------------
// extern "C" launchers
// ---------------------------------------------------------------------------------------------
using namespace vt;

extern "C" int vt_conv3x3_fwd(const float *x, const float *w, const float *b, float *conv_out, float *part, int N, int Ci,
                              int H, int W, vt_stream s) {
  if (N < 1 || Ci < 1 || H * W > PPT * TPB || H < 1 || W < 1) return VT_EUNSUPPORTED;
  const size_t lds = (size_t)(((Ci * (H + 2) * (W + 2) + 3) & ~3) + NW * C1) * sizeof(float);
  if (lds > 64 * 1024) return VT_EUNSUPPORTED;
  hipLaunchKernelGGL(k_conv_fwd, dim3(N), dim3(TPB), lds, (hipStream_t)s, x, w, b, conv_out, part, Ci, H, W);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_bn_stats_finalize(const float *part, int nblocks, int C, long rows_per_block, long rows_total,
                                    float *running_mean, float *running_var, float *stat, vt_stream s) {
  if (C < 1 || C > 64 || FIN_T % C != 0 || nblocks < 1) return VT_EUNSUPPORTED;
  if ((long)(nblocks - 1) * rows_per_block >= rows_total || (long)nblocks * rows_per_block < rows_total) return VT_EINVAL;
  hipLaunchKernelGGL(k_bn_stats_finalize, dim3(1), dim3(FIN_T), 0, (hipStream_t)s, part, nblocks, C, rows_per_block,
                     rows_total, running_mean, running_var, stat);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_bn_stats_eval(const float *running_mean, const float *running_var, int C, float *stat, vt_stream s) {
  if (C < 1 || C > 64) return VT_EUNSUPPORTED;
  hipLaunchKernelGGL(k_bn_stats_eval, dim3(1), dim3(64), 0, (hipStream_t)s, running_mean, running_var, C, stat);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_transformer_ffn_fwd(const float *tokens, const float *n1_stat, const float *n1_gamma,
                                      const float *n1_beta, const float *w1, const float *b1, const float *w2,
                                      const float *b2, float *a, float *hdn, float *s2, float *part_n2, long rows,
                                      vt_stream s) {
  if (rows < 1) return VT_EINVAL;
  const int nb = (int)((rows + RB - 1) / RB);
  hipLaunchKernelGGL(k_tr_ffn_fwd, dim3(nb), dim3(TPB), 0, (hipStream_t)s, tokens, n1_stat, n1_gamma, n1_beta, w1, b1,
                     w2, b2, a, hdn, s2, part_n2, rows);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_projector_tokens_fwd(const float *s2, const float *n2_stat, const float *n2_gamma,
                                       const float *n2_beta, const float *wq, const float *wk, const float *wv,
                                       const float *bv, float *o, float *pk, int N, vt_stream s) {
  if (N < 1) return VT_EINVAL;
  hipLaunchKernelGGL(k_proj_tokens_fwd, dim3(N), dim3(TPB), 0, (hipStream_t)s, s2, n2_stat, n2_gamma, n2_beta, wq, wk,
                     wv, bv, o, pk);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_metrics(const float *probs, const int *labels, int N, int K, float *loss_sum, int *argmax,
                          vt_stream s) {
  if (N < 1 || K < 1) return VT_EINVAL;
  hipLaunchKernelGGL(k_metrics, dim3(1), dim3(TPB), 0, (hipStream_t)s, probs, labels, N, K, loss_sum, argmax);
  VT_CHECK_LAUNCH();
  return 0;
}
