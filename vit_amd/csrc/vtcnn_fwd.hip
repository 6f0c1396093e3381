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
#include "vtcnn_tail.h"

namespace vt {

// ---------------------------------------------------------------------------------------------
// Conv2DLayer::forward 3x3/s1/p1 (reference layers/conv2d_layer.hpp:35-87) + per-sample partial
// statistics for BatchNorm2D (reference layers/batch_normalization_layer.hpp:274-276).
// One workgroup per image; the zero-padded image tile is staged in LDS.
// ---------------------------------------------------------------------------------------------
// Index arithmetic without integer division: rmagic = 2^32 / d + 1 gives n / d = umulhi(n, rmagic)
// exactly for the sizes here (n < 2^13, d < 2^11: checked exhaustively).
template <bool TAIL>  // TAIL: the last workgroup closes the BatchNorm2D statistics (vtcnn_tail.h)
__global__ __launch_bounds__(TPB) void k_conv_fwd(const float *__restrict__ X, const float *__restrict__ Wc,
                                                  const float *__restrict__ bc, float *__restrict__ conv,
                                                  float *__restrict__ part, int Ci, int H, int W, unsigned w_magic,
                                                  unsigned tw_magic, const BnFwdTail tail) {
  extern __shared__ float sm[];
  const int P = H * W, TH = H + 2, TW = W + 2, plane = TH * TW;
  const int tile_n = Ci * plane;
  float *tile = sm;
  float *red = sm + ((tile_n + 3) & ~3);
  const int n = blockIdx.x, tid = threadIdx.x;
  // every lane's first ST_K tile elements are fetched before any is stored: one memory round trip per
  // channel instead of one per element (all workgroups of the launch sit in this phase together)
  constexpr int ST_K = 5;
  for (int ci = 0; ci < Ci; ++ci) {
    const float *__restrict__ src = X + (size_t)(n * Ci + ci) * P;
    float v[ST_K];
#pragma unroll
    for (int k = 0; k < ST_K; ++k) {
      const int i = min(tid + k * TPB, plane - 1);
      const int r = (int)__umulhi((unsigned)i, tw_magic), hh = r - 1, ww = i - r * TW - 1;
      const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
      const float x = src[in ? hh * W + ww : 0];
      v[k] = in ? x : 0.f;
    }
#pragma unroll
    for (int k = 0; k < ST_K; ++k)
      if (tid + k * TPB < plane) tile[ci * plane + tid + k * TPB] = v[k];
    for (int i = tid + ST_K * TPB; i < plane; i += TPB) {
      const int r = (int)__umulhi((unsigned)i, tw_magic), hh = r - 1, ww = i - r * TW - 1;
      const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
      const float x = src[in ? hh * W + ww : 0];
      tile[ci * plane + i] = in ? x : 0.f;
    }
  }
  __syncthreads();
  float vals[PPT][C1];
  float s[C1];
#pragma unroll
  for (int co = 0; co < C1; ++co) s[co] = 0.f;
#pragma unroll
  for (int it = 0; it < PPT; ++it) {
    if (it * TPB + (tid & ~63) < P) {  // wave-uniform: waves past the last pixel skip the arithmetic
      const int p = tid + it * TPB;
      const bool ok = p < P;
      const int pc = ok ? p : 0;
      const int h = (int)__umulhi((unsigned)pc, w_magic), w = pc - h * W;
      float acc[C1];
#pragma unroll
      for (int co = 0; co < C1; ++co) acc[co] = bc[co];
      for (int ci = 0; ci < Ci; ++ci) {
        const float *__restrict__ tp = tile + ci * plane + h * TW + w;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float v = tp[kh * TW + kw];
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
    } else {
#pragma unroll
      for (int co = 0; co < C1; ++co) vals[it][co] = 0.f;
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
  if constexpr (!TAIL) {
    if (tid < C1) {
      part[(size_t)n * 2 * C1 + tid] = s[tid];
      part[(size_t)n * 2 * C1 + C1 + tid] = m2[tid];
    }
  } else {
    if (tid < C1) {
      tail_store(part + (size_t)n * 2 * C1 + tid, s[tid]);
      tail_store(part + (size_t)n * 2 * C1 + C1 + tid, m2[tid]);
    }
    bn_fwd_tail(tail, part, n, gridDim.x, C1);
  }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm statistics finalisation (reference batch_normalization_layer.hpp:152-159, 274-283):
// merges per-block (sum, centred M2) partials with Chan's formula in double precision, writes
// stat = (mean, 1/sqrt(var+eps)) and updates running = 0.1*batch + 0.9*running (biased var).
//   S = sum_b sum_b, Q = sum_b M2_b, U = sum_b sum_b^2/cnt_b;  mean = S/R,  M2 = Q + U - S^2/R
// One workgroup per channel: a thread takes every FIN_T-th block, its loads are independent (eight
// in flight), 1/cnt_b is a constant except for the last block, so the loop has no division; the
// three sums then cross the workgroup through wave shuffles and one LDS hop.
// ---------------------------------------------------------------------------------------------
constexpr int FIN_T = 256;
__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
// Grouped form (blockIdx.y = statistics group): group g owns blocks [g*nbpg, (g+1)*nbpg) and
// rows [g*rows_group, ...); its statistics go to stat[g][2C] and, instead of touching the running
// buffers, its (mean, biased var) to batch_mv[g][2C] for k_bn_running_seq.  With one group the
// running buffers are updated here.
__global__ __launch_bounds__(FIN_T) void k_bn_stats_finalize(const float *__restrict__ part, int nb_total, int nbpg, int C,
                                                             long rows_per_block, long rows_group, long rows_total,
                                                             float *__restrict__ run_mean, float *__restrict__ run_var,
                                                             float *__restrict__ stat, float *__restrict__ batch_mv) {
  __shared__ double sh[3][FIN_T / 64];
  const int tid = threadIdx.x, c = blockIdx.x, grp = blockIdx.y;
  const int bfirst = grp * nbpg, nb = min(nbpg, nb_total - bfirst);
  const long rows_here = min(rows_group, rows_total - (long)grp * rows_group);
  part += (size_t)bfirst * 2 * C;
  const long last_cnt = rows_here - (long)(nb - 1) * rows_per_block;
  const double rc = 1.0 / (double)rows_per_block, rc_last = 1.0 / (double)last_cnt;
  // the running buffers are only needed by the very last instruction: requested now, their round trip runs
  // under the partial loads instead of after the reduction
  float rm_old = 0.f, rv_old = 0.f;
  if (tid == 0 && batch_mv == nullptr) {
    rm_old = run_mean[c];
    rv_old = run_var[c];
  }
  double S = 0.0, Q = 0.0, U = 0.0;
  for (int b0 = tid; b0 < nb; b0 += 8 * FIN_T) {
    float sv[8], qv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = min(b0 + k * FIN_T, nb - 1);
      sv[k] = part[(size_t)b * 2 * C + c];
      qv[k] = part[(size_t)b * 2 * C + C + c];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = b0 + k * FIN_T;
      if (b < nb) {
        const double s = (double)sv[k];
        S += s;
        Q += (double)qv[k];
        U += s * s * (b == nb - 1 ? rc_last : rc);
      }
    }
  }
  S = wave_sum_d(S);
  Q = wave_sum_d(Q);
  U = wave_sum_d(U);
  if ((tid & 63) == 0) {
    sh[0][tid >> 6] = S;
    sh[1][tid >> 6] = Q;
    sh[2][tid >> 6] = U;
  }
  __syncthreads();
  if (tid == 0) {
    double s = 0.0, q = 0.0, u = 0.0;
#pragma unroll
    for (int k = 0; k < FIN_T / 64; ++k) {
      s += sh[0][k];
      q += sh[1][k];
      u += sh[2][k];
    }
    const double R = (double)rows_here;
    const double mu = s / R;
    double m2 = q + u - s * s / R;
    if (m2 < 0.0) m2 = 0.0;
    const float mean = (float)mu;
    const float var = (float)(m2 / R);
    stat[(size_t)grp * 2 * C + c] = mean;
    stat[(size_t)grp * 2 * C + C + c] = 1.0f / sqrtf(var + BN_EPS);
    if (batch_mv != nullptr) {
      batch_mv[(size_t)grp * 2 * C + c] = mean;
      batch_mv[(size_t)grp * 2 * C + C + c] = var;
    } else {
      run_mean[c] = mean * BN_MOM + rm_old * (1 - BN_MOM);
      run_var[c] = var * BN_MOM + rv_old * (1 - BN_MOM);
    }
  }
}

// running = 0.1*batch + 0.9*running applied once per group, in group order: what the reference's
// loop over mini-batches leaves in the running buffers (batch_normalization_layer.hpp:156-159).
__global__ void k_bn_running_seq(const float *__restrict__ batch_mv, int groups, int C, float *__restrict__ run_mean,
                                 float *__restrict__ run_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float rm = run_mean[c], rv = run_var[c];
  for (int g0 = 0; g0 < groups; g0 += 16) {  // 32 independent loads, then the ordered recurrence
    float bm[16], bv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int g = min(g0 + k, groups - 1);
      bm[k] = batch_mv[(size_t)g * 2 * C + c];
      bv[k] = batch_mv[(size_t)g * 2 * C + C + c];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (g0 + k < groups) {
        rm = bm[k] * BN_MOM + rm * (1 - BN_MOM);
        rv = bv[k] * BN_MOM + rv * (1 - BN_MOM);
      }
  }
  run_mean[c] = rm;
  run_var[c] = rv;
}

// Eval-mode statistics from the running buffers (batch_normalization_layer.hpp:287-291).
__global__ void k_bn_stats_eval(const float *__restrict__ run_mean, const float *__restrict__ run_var, int C,
                                float *__restrict__ stat) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c < C) {
    stat[c] = run_mean[c];
    stat[C + c] = 1.0f / sqrtf(run_var[c] + BN_EPS);
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

// DataLoader::next_batch on the device (reference utils/load_dataset.hpp:129-151 with the pixel
// conversion of loadImages2D, :47-51, fused in): the dataset stays resident as the IDX file's
// bytes, a batch is the rows named by a slice of the shuffled index vector, pixel = u8 / 255.0f
// (IEEE division, as on the host).  One thread per 4 pixels.
__global__ __launch_bounds__(TPB) void k_loader_gather(const unsigned char *__restrict__ images,
                                                       const int *__restrict__ labels, const int *__restrict__ indices,
                                                       int n, int pixels, float *__restrict__ x, int *__restrict__ y) {
  const int quads = pixels >> 2;  // pixels % 4 == 0 (checked by the launcher)
  const long t = (long)blockIdx.x * TPB + threadIdx.x;
  if (t >= (long)n * quads) return;
  const int i = (int)(t / quads), q = (int)(t % quads);
  const int src = indices[i];
  const uchar4 b = *reinterpret_cast<const uchar4 *>(images + (size_t)src * pixels + 4 * q);
  float4 o;
  o.x = __fdiv_rn((float)b.x, 255.0f);
  o.y = __fdiv_rn((float)b.y, 255.0f);
  o.z = __fdiv_rn((float)b.z, 255.0f);
  o.w = __fdiv_rn((float)b.w, 255.0f);
  *reinterpret_cast<float4 *>(x + (size_t)i * pixels + 4 * q) = o;
  if (q == 0 && y != nullptr) y[i] = labels[src];
}

// Trainer::evaluate / evaluate_test bookkeeping for one batch without leaving the device
// (reference utils/trainer.hpp:104-127, tensor_cpu.hpp:1137-1173, 1294-1357): arg-max (first
// maximum), the batch's cross-entropy sum and number of correct predictions into slot `slot` of
// two per-batch arrays, and the confusion matrix cm[truth][pred] += 1 (integer atomics: the result
// does not depend on their order).  One workgroup per mini-batch of the sweep.
__global__ __launch_bounds__(TPB) void k_eval_accumulate(const float *__restrict__ probs, const int *__restrict__ labels,
                                                         int N, int K, int group, int slot0, float *__restrict__ loss_sums,
                                                         int *__restrict__ correct, int *__restrict__ cm,
                                                         int *__restrict__ argmax) {
  __shared__ float red[NW];
  __shared__ int scm[KMAX * KMAX];
  __shared__ int sok;
  // workgroup = one mini-batch of the sweep: images [first, first + cnt)
  const int first = blockIdx.x * group, cnt = min(group, N - first);
  const bool staged = K <= KMAX;  // larger class counts add to the global matrix directly
  for (int i = threadIdx.x; staged && i < K * K; i += TPB) scm[i] = 0;
  if (threadIdx.x == 0) sok = 0;
  __syncthreads();
  float part[1] = {0.f};
  int ok = 0;
  for (int i = threadIdx.x; i < cnt; i += TPB) {
    const int n = first + i;
    int best = 0;
    float bv = probs[(size_t)n * K];
    for (int k = 1; k < K; ++k) {
      const float v = probs[(size_t)n * K + k];
      if (v > bv) {
        bv = v;
        best = k;
      }
    }
    const int truth = labels[n];
    if (argmax) argmax[n] = best;
    part[0] += -logf(probs[(size_t)n * K + truth] + 1e-8f);
    ok += best == truth ? 1 : 0;
    if (truth >= 0 && truth < K) {  // integer adds: order-independent
      if (staged) atomicAdd(&scm[truth * K + best], 1);
      else if (cm != nullptr) atomicAdd(&cm[(size_t)truth * K + best], 1);
    }
  }
  atomicAdd(&sok, ok);
  block_sum<1>(part, red);
  if (threadIdx.x == 0) {
    loss_sums[slot0 + blockIdx.x] = part[0];
    correct[slot0 + blockIdx.x] = sok;
  }
  if (cm != nullptr && staged)
    for (int i = threadIdx.x; i < K * K; i += TPB)
      if (scm[i] != 0) atomicAdd(&cm[i], scm[i]);
}

}  // namespace vt

// ---------------------------------------------------------------------------------------------
// extern "C" launchers
// ---------------------------------------------------------------------------------------------
using namespace vt;

int vt::conv3x3_fwd_tail(const float *x, const float *w, const float *b, float *conv_out, float *part, int N, int Ci, int H,
                         int W, const BnFwdTail &tail, hipStream_t s) {
  if (N < 1 || Ci < 1 || H * W > PPT * TPB || H < 1 || W < 1) return VT_EUNSUPPORTED;
  const size_t lds = (size_t)(((Ci * (H + 2) * (W + 2) + 3) & ~3) + NW * C1) * sizeof(float);
  if (lds > 56 * 1024) return VT_EUNSUPPORTED;  // the closing step adds static LDS of its own
  const unsigned wm = (unsigned)(0x100000000ull / (unsigned)W + 1), twm = (unsigned)(0x100000000ull / (unsigned)(W + 2) + 1);
  if (part && tail.ticket)
    hipLaunchKernelGGL(k_conv_fwd<true>, dim3(N), dim3(TPB), lds, s, x, w, b, conv_out, part, Ci, H, W, wm, twm, tail);
  else
    hipLaunchKernelGGL(k_conv_fwd<false>, dim3(N), dim3(TPB), lds, s, x, w, b, conv_out, part, Ci, H, W, wm, twm, BnFwdTail{});
  VT_CHECK_LAUNCH();
  return 0;
}
extern "C" int vt_conv3x3_fwd(const float *x, const float *w, const float *b, float *conv_out, float *part, int N, int Ci,
                              int H, int W, vt_stream s) {
  return conv3x3_fwd_tail(x, w, b, conv_out, part, N, Ci, H, W, BnFwdTail{}, (hipStream_t)s);
}

extern "C" int vt_bn_stats_finalize(const float *part, int nblocks, int C, long rows_per_block, long rows_total,
                                    float *running_mean, float *running_var, float *stat, vt_stream s) {
  if (C < 1 || C > 1024 || nblocks < 1) return VT_EUNSUPPORTED;
  if ((long)(nblocks - 1) * rows_per_block >= rows_total || (long)nblocks * rows_per_block < rows_total) return VT_EINVAL;
  hipLaunchKernelGGL(k_bn_stats_finalize, dim3(C, 1), dim3(FIN_T), 0, (hipStream_t)s, part, nblocks, nblocks, C,
                     rows_per_block, rows_total, rows_total, running_mean, running_var, stat, (float *)nullptr);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_bn_stats_finalize_grouped(const float *part, int nblocks, int blocks_per_group, int C,
                                            long rows_per_block, long rows_total, float *running_mean,
                                            float *running_var, float *stat, float *batch_mv, vt_stream s) {
  if (C < 1 || C > 1024 || nblocks < 1 || blocks_per_group < 1 || !batch_mv) return VT_EUNSUPPORTED;
  if ((long)(nblocks - 1) * rows_per_block >= rows_total || (long)nblocks * rows_per_block < rows_total) return VT_EINVAL;
  const int groups = (nblocks + blocks_per_group - 1) / blocks_per_group;
  hipLaunchKernelGGL(k_bn_stats_finalize, dim3(C, groups), dim3(FIN_T), 0, (hipStream_t)s, part, nblocks, blocks_per_group,
                     C, rows_per_block, (long)blocks_per_group * rows_per_block, rows_total, running_mean, running_var,
                     stat, batch_mv);
  VT_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_bn_running_seq, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)s, batch_mv, groups, C, running_mean,
                     running_var);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_bn_stats_eval(const float *running_mean, const float *running_var, int C, float *stat, vt_stream s) {
  if (C < 1) return VT_EUNSUPPORTED;
  hipLaunchKernelGGL(k_bn_stats_eval, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)s, running_mean, running_var, C, stat);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_loader_gather(const unsigned char *images_u8, const int *labels, const int *indices, int n, int pixels,
                                float *x, int *y, vt_stream s) {
  if (n < 1 || pixels < 4 || pixels % 4 != 0 || !images_u8 || !indices || !x || (y && !labels)) return VT_EINVAL;
  const long threads = (long)n * (pixels / 4);
  hipLaunchKernelGGL(k_loader_gather, dim3((unsigned)((threads + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)s, images_u8,
                     labels, indices, n, pixels, x, y);
  VT_CHECK_LAUNCH();
  return 0;
}

extern "C" int vt_eval_accumulate_grouped(const float *probs, const int *labels, int N, int K, int group, int slot0,
                                          float *loss_sums, int *correct, int *confusion, int *argmax, vt_stream s) {
  if (N < 1 || K < 1 || group < 1 || slot0 < 0 || !probs || !labels || !loss_sums || !correct) return VT_EINVAL;
  hipLaunchKernelGGL(k_eval_accumulate, dim3((N + group - 1) / group), dim3(TPB), 0, (hipStream_t)s, probs, labels, N, K,
                     group, slot0, loss_sums, correct, confusion, argmax);
  VT_CHECK_LAUNCH();
  return 0;
}
extern "C" int vt_eval_accumulate(const float *probs, const int *labels, int N, int K, int slot, float *loss_sums,
                                  int *correct, int *confusion, int *argmax, vt_stream s) {
  return vt_eval_accumulate_grouped(probs, labels, N, K, N, slot, loss_sums, correct, confusion, argmax, s);
}

extern "C" int vt_metrics(const float *probs, const int *labels, int N, int K, float *loss_sum, int *argmax,
                          vt_stream s) {
  if (N < 1 || K < 1) return VT_EINVAL;
  hipLaunchKernelGGL(k_metrics, dim3(1), dim3(TPB), 0, (hipStream_t)s, probs, labels, N, K, loss_sum, argmax);
  VT_CHECK_LAUNCH();
  return 0;
}
