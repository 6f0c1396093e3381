// BatchNorm reductions finished inside the kernel that produces their partial sums.
//
// Every mini-batch-wide BatchNorm statistic of the step (reference layers/batch_normalization_layer.hpp:
// 152-168 forward, 187-199 / 316-326 backward) is a sum over all workgroups of a producer kernel, followed
// by a few flops per channel.  As a kernel of its own that closing step costs a dependent launch (about
// 2 us of boundary plus a 4 us latency-bound kernel) eight times per training step.  Here the workgroup
// that finishes LAST does it, inside the producer's launch:
//   * every workgroup writes its partial record with write-through (agent-scope) stores, each storing wave
//     waits for them (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, and one lane takes a ticket
//     with a returning agent-scope atomic add;
//   * tickets are sharded (32 counters on lines of their own, then one top counter taken by the last
//     arriver of each shard), so a thousand workgroups finishing together do not queue on one word;
//   * the workgroup whose top ticket completes the count knows every other workgroup's record is in
//     memory; it invalidates its L1 (agent-scope acquire), reads all records and closes the reduction.
// No workgroup ever waits for another one, so there is nothing that can hang; the counters are left at
// zero for the next launch by the workgroups that complete them.
// Sums over the records run in double precision in a fixed order (thread = record column x row lane,
// row lanes folded in order), so results do not depend on which workgroup happens to be last.
#pragma once
#include "vtcnn_common.h"

namespace vt {

constexpr int TAIL_SHARDS = 32;
constexpr int TAIL_WORDS = (TAIL_SHARDS + 1) * 32;  // unsigned words of one ticket set (each counter on its own 128 bytes)

// closing step of a forward statistic: stat = (mean, 1/sqrt(var+eps)), running <- 0.1 batch + 0.9 running
struct BnFwdTail {
  unsigned *ticket = nullptr;  // nullptr: the kernel stops after its partial records (finalize launched separately)
  float *run_mean = nullptr, *run_var = nullptr, *stat = nullptr;
  long rows_per_block = 0, rows_total = 0;
};
// closing step of a backward reduction: grad gamma / beta and coef = (scale, shift), delta_in = delta*scale + shift
struct BnBwdTail {
  unsigned *ticket = nullptr;
  const float *gamma = nullptr, *stat = nullptr, *in_row0 = nullptr;
  float *ggamma = nullptr, *gbeta = nullptr, *coef = nullptr;
  long rows_total = 0;
};

// Launchers of the producer kernels with the closing step attached (an empty tail = the plain launcher).
int conv3x3_fwd_tail(const float *x, const float *w, const float *b, float *conv_out, float *part, int N, int Ci, int H,
                     int W, const BnFwdTail &tail, hipStream_t s);
int tokenizer_fwd_tail(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                       const float *w1, const float *b1, const float *w2, const float *b2, float *tok_A, float *tok_stat,
                       float *tokens, float *part_n1, int N, int P, const BnFwdTail &tail, hipStream_t s, int sgroup = 0);
int transformer_ffn_fwd_tail(const float *tok, const float *n1_stat, const float *n1_gamma, const float *n1_beta,
                             const float *w1, const float *b1, const float *w2, const float *b2, float *a, float *hdn, float *s2,
                             float *part, long rows, const BnFwdTail &tail, hipStream_t s);
int projector_fwd_tail(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                       const float *pk, const float *wd, const float *bd, const float *pivot, float *proj_A, float *z,
                       float *part_pn, int N, int P, const BnFwdTail &tail, hipStream_t s, int sgroup = 0);

int projector_tokens_bwd_transformer_a_tail(const float *o, const float *dv, const float *dk, const float *wv, const float *wk,
                                            float *pdt, float *g_wk_wv_bv, float *scratch, const float *s2,
                                            const float *n2_stat, const float *tok_w2, float *part, float *dT0, long rows,
                                            const BnBwdTail &tail, hipStream_t s, long group_rows = 0);
int transformer_bwd_b_tail(const float *o, const float *pdt, const float *n2_coef, const float *hdn, const float *a,
                           const float *tok, const float *n1_stat, const float *w1, const float *w2, float *ds2, float *da,
                           float *g_ffn, float *scratch, float *part, long rows, const BnBwdTail &tail, hipStream_t s,
                           long group_rows = 0);
int tokenizer_bwd_tail(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                       const float *tok_stat, const float *dT0, const float *w1, const float *b1, const float *pdx, float *dy,
                       float *part, int N, int P, const BnBwdTail &tail, hipStream_t s, int sgroup = 0);
int head_bwd_windows_tail(const float *probs, const int *labels, const float *wfc, const unsigned char *pool_idx,
                          const float *pooled, const float *z, const float *pn_stat, const float *pn_gamma,
                          const float *pn_beta, float *d_logits, float *dpair, float *part, int N, int H, int K,
                          const BnBwdTail &tail, hipStream_t s, int sgroup = 0);
int head_bwd_windows_blocks_grouped(int N, int sgroup, int *per_group);

// Launchers with one statistics record per group of `sgroup` consecutive images (sgroup = 0: one group), for steps
// whose mini-batch is several independent reference mini-batches side by side.
int projector_bwd_grouped(const float *du, int windows, const float *pn_coef, const float *conv_out, const float *bn_stat,
                          const float *bn_gamma, const float *bn_beta, const float *pk, const float *wq, const float *wd,
                          float *pdx, float *dv, float *dk, float *gslab, int N, int P, int sgroup, hipStream_t s);
int transformer_bwd_c_grouped(const float *da, const float *n1_coef, const float *wo, const float *tok, const float *wqkv,
                              float *dr, float *dc, float *qkv, long rows, long group_rows, hipStream_t s);
int conv3x3_wgrad_grouped(const float *x, const float *dy, const float *bn_coef, float *gslab, int N, int Ci, int H, int W,
                          int sgroup, hipStream_t s);

// write-through store / L1-bypassing load of one float of a partial record
__device__ __forceinline__ void tail_store(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float tail_load(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by EVERY thread of EVERY workgroup after the workgroup's tail_store calls.  Returns true (to all
// threads of the workgroup) in exactly one workgroup of the launch: the one that arrived last.
// block = linear workgroup index, nblocks = workgroups in the launch.
__device__ __forceinline__ bool tail_arrive(unsigned *ticket, int block, int nblocks) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's record stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nsh = min(TAIL_SHARDS, nblocks), sh = block % nsh;
    const unsigned in_shard = (unsigned)((nblocks - sh + nsh - 1) / nsh);
    bool last = false;
    const unsigned got = __hip_atomic_fetch_add(ticket + sh * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (got == in_shard - 1) {
      ticket[sh * 32] = 0;  // every workgroup of the shard has been here: ready for the next launch
      const unsigned top = __hip_atomic_fetch_add(ticket + TAIL_SHARDS * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (top == (unsigned)nsh - 1) {
        ticket[TAIL_SHARDS * 32] = 0;
        last = true;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop this CU's L1 copies of the records
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    s_last = last ? 1 : 0;
  }
  __syncthreads();
  return s_last != 0;
}

// Column sums of the nb records of 2C floats, by the whole workgroup (TPB threads, 2C a divisor of TPB),
// in double precision: thread = (column j, row lane r) takes records r, r + lanes, ...; lanes are folded in
// order.  Also the between-block term of Chan's variance merge, U[j] = sum_b rec[b][j]^2 / cnt_b, when
// rc > 0 (forward statistics; cnt_b = rows_per_block except in the last record).
// Results in LDS: sh[j] = column sum, sh[64 + j] = U[j] for j < 2C.  sh: TPB doubles.
__device__ __forceinline__ void tail_column_sums(const float *part, int nb, int C, double rc, double rc_last, double *sh) {
  const int tid = threadIdx.x, cols = 2 * C, lanes = TPB / cols, j = tid % cols, r = tid / cols;
  double a = 0.0, u = 0.0;
  for (int b0 = r; b0 < nb; b0 += 8 * lanes) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tail_load(part + (size_t)min(b0 + k * lanes, nb - 1) * cols + j);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = b0 + k * lanes;
      if (b < nb) {
        const double x = (double)v[k];
        a += x;
        u += x * x * (b == nb - 1 ? rc_last : rc);
      }
    }
  }
  double colsum = 0.0, colu = 0.0;
  sh[tid] = a;
  __syncthreads();
  if (tid < cols)
    for (int q = 0; q < lanes; ++q) colsum += sh[q * cols + tid];
  __syncthreads();
  sh[tid] = u;
  __syncthreads();
  if (tid < cols)
    for (int q = 0; q < lanes; ++q) colu += sh[q * cols + tid];
  __syncthreads();
  if (tid < cols) {
    sh[tid] = colsum;
    sh[64 + tid] = colu;
  }
  __syncthreads();
}

// The closing arithmetic of k_bn_stats_finalize for one channel, from S = sum of sums, Q = sum of centred
// M2, U = sum of sum_b^2 / cnt_b:  mean = S/R,  M2 = Q + U - S^2/R  (Chan's merge).
__device__ __forceinline__ void bn_fwd_close(double S, double Q, double U, long rows, int C, int c, float *run_mean,
                                             float *run_var, float *stat) {
  const double R = (double)rows;
  const double mu = S / R;
  double m2 = Q + U - S * S / R;
  if (m2 < 0.0) m2 = 0.0;
  const float mean = (float)mu;
  const float var = (float)(m2 / R);
  stat[c] = mean;
  stat[C + c] = 1.0f / sqrtf(var + BN_EPS);
  run_mean[c] = mean * BN_MOM + run_mean[c] * (1 - BN_MOM);
  run_var[c] = var * BN_MOM + run_var[c] * (1 - BN_MOM);
}

// The closing arithmetic of k_bn_bwd_finalize for one channel (S1 = sum delta, S2 = sum delta * x_hat).
__device__ __forceinline__ void bn_bwd_close(float S1, float S2, long rows_total, int C, int c, const float *gamma,
                                             const float *stat, const float *in_row0, float *ggamma, float *gbeta,
                                             float *coef) {
  const float gm = gamma[c], mean = stat[c], inv = stat[C + c], x0 = in_row0[c];
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

// Forward statistic closed by the last workgroup.  Every thread of every workgroup calls this after the
// workgroup's record (part[block][2C] = per-channel sum, centred M2) went out through tail_store.
__device__ __forceinline__ void bn_fwd_tail(const BnFwdTail &t, const float *part, int block, int nblocks, int C) {
  if (t.ticket == nullptr) return;
  if (!tail_arrive(t.ticket, block, nblocks)) return;
  __shared__ double sh[TPB];
  const long last_cnt = t.rows_total - (long)(nblocks - 1) * t.rows_per_block;
  tail_column_sums(part, nblocks, C, 1.0 / (double)t.rows_per_block, 1.0 / (double)last_cnt, sh);
  const int tid = threadIdx.x;
  if (tid < C) bn_fwd_close(sh[tid], sh[C + tid], sh[64 + tid], t.rows_total, C, tid, t.run_mean, t.run_var, t.stat);
}

// Backward reduction closed by the last workgroup (part[block][2C] = sum delta, sum delta * x_hat).
__device__ __forceinline__ void bn_bwd_tail(const BnBwdTail &t, const float *part, int block, int nblocks, int C) {
  if (t.ticket == nullptr) return;
  if (!tail_arrive(t.ticket, block, nblocks)) return;
  __shared__ double sh[TPB];
  tail_column_sums(part, nblocks, C, 0.0, 0.0, sh);
  const int tid = threadIdx.x;
  if (tid < C)
    bn_bwd_close((float)sh[tid], (float)sh[C + tid], t.rows_total, C, tid, t.gamma, t.stat, t.in_row0, t.ggamma, t.gbeta,
                 t.coef);
}

}  // namespace vt
