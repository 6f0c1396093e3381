// MultiHeadAttention::backward, per-head part (reference layers/multi_head_attention.hpp:139-181),
// for gfx950.
//
// The layer's "batch" axis is the token index and its "sequence" axis is the mini-batch, because
// Tensor::permute relabels the shape without moving data (utils/tensor/tensor_cpu.hpp:719-747):
// attention group b = rows b*T .. b*T+T-1 of the flat [N*L, D] token matrix, T = N.  Only the
// weight gradients of the per-head query / key / value Dense layers survive: the forward output
// is multiplied by a zero concat (tensor_cpu.hpp:251) and the layer's input delta is discarded
// (layers/visual_transformer.hpp:160).  So the attention backward is restated row-wise:
//   scores[i][j] = K_i.Q_j / sqrt(d);  attn = softmax_j;  dattn[i][j] = dhead_i . V_j
//   dot_i = sum_j dattn*attn;  dsc = attn * (dattn - dot_i)     (no 1/sqrt(d) in backward, :165-170)
//   Y_i = sum_j attn[i][j] X_j;  Z_i = sum_j dsc[i][j] X_j
//   gWv = sum_i Y_i^T dhead_i;  gWq = sum_i Z_i^T K_i;  gWk = sum_i X_i^T (Z_i Wq)
// which equals X^T dV, X^T dQ, X^T dK of the reference without ever forming the T x T matrices
// (flash-style: at T = 1024 they would be 4 MB per group and head).
//
// Mapping: one workgroup = 64 query-side rows i (lane = row) of one (head, group); its four waves
// sweep disjoint quarters of the j range and merge their unnormalised sums through LDS.  Each
// wave stages tiles of 32 rows j (Q_j, V_j, X_j: 64 floats per row) into its own LDS region with
// coalesced 16-byte loads and consumes them with wave-uniform (broadcast) ds_read_b128.
// Softmax uses a per-row upper bound of the scores instead of the exact row maximum
// (|K_i.Q_j| <= |K_i| max_j |Q_j|): softmax is shift invariant, so this changes nothing but the
// rounding, and it removes a whole pass over j.
#include "vtcnn_common.h"

namespace vt {

constexpr int MH_STRIDE = 3 * D * HD;  // per-workgroup partials: gq[D*HD] | gk[D*HD] | gv[D*HD]
constexpr int TJ = 32;                 // rows j per staged tile
constexpr int ROWF = 2 * HD + D;       // floats per staged row: Q(16) V(16) X(32)

__global__ __launch_bounds__(TPB) void k_mha_bwd(const float *__restrict__ qkv, const float *__restrict__ dc,
                                                 const float *__restrict__ tok, const float *__restrict__ Wqkv,
                                                 float *__restrict__ scratch, int T, long rows, int groups) {
  // LDS: staging (4 waves x TJ x 64 floats = 32 KB) is dead once the j sweep ends and is reused
  // for the cross-wave merge (3 x 64 x 67 floats = 51 KB).
  __shared__ __align__(16) float smem[(NW - 1) * 64 * (2 * D + 3)];
  __shared__ float sqmax[NW];
  __shared__ float sout[MH_STRIDE];
  const int tiles = (T + 63) / 64;
  const int tile = blockIdx.x % tiles, bgrp = (blockIdx.x / tiles) % groups, h = blockIdx.x / (tiles * groups);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = tile * 64 + lane;
  const bool ok = i < T;
  const long gbase = (long)bgrp * T;
  const long ri = gbase + (ok ? i : 0);
  const float *__restrict__ Qm = qkv, *__restrict__ Km = qkv + rows * D, *__restrict__ Vm = qkv + 2 * rows * D;
  const float scale = 1.0f / sqrtf((float)HD);
  float Ki[HD], dh[HD];
  float kn2 = 0.f;
#pragma unroll
  for (int q = 0; q < HD / 4; ++q) {
    const float4 kv = *reinterpret_cast<const float4 *>(Km + ri * D + h * HD + 4 * q);
    const float4 dv = *reinterpret_cast<const float4 *>(dc + ri * D + h * HD + 4 * q);
    Ki[4 * q] = kv.x; Ki[4 * q + 1] = kv.y; Ki[4 * q + 2] = kv.z; Ki[4 * q + 3] = kv.w;
    dh[4 * q] = dv.x; dh[4 * q + 1] = dv.y; dh[4 * q + 2] = dv.z; dh[4 * q + 3] = dv.w;
    kn2 += kv.x * kv.x + kv.y * kv.y + kv.z * kv.z + kv.w * kv.w;
  }
  const int jchunk = (T + NW - 1) / NW;
  const int j0 = wave * jchunk, j1 = min(T, j0 + jchunk);
  // largest |Q_j|^2 of the group (all four waves together cover every j)
  float qn2 = 0.f;
  for (int j = j0 + lane; j < j1; j += 64) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < HD / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(Qm + (gbase + j) * D + h * HD + 4 * q);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    qn2 = fmaxf(qn2, s);
  }
  qn2 = wave_max(qn2);
  if (lane == 0) sqmax[wave] = qn2;
  __syncthreads();
  qn2 = fmaxf(fmaxf(sqmax[0], sqmax[1]), fmaxf(sqmax[2], sqmax[3]));
  const float mx = scale * sqrtf(kn2) * sqrtf(qn2);  // >= every scaled score of this row

  float Yt[D], Wt[D], ssum = 0.f, tsum = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) { Yt[d] = 0.f; Wt[d] = 0.f; }
  float *stage = smem + wave * (TJ * ROWF);
  const int ntile = (jchunk + TJ - 1) / TJ;  // same trip count for every wave (barriers inside)
  for (int t = 0; t < ntile; ++t) {
    const int jt = j0 + t * TJ;
    const int cnt = max(0, min(TJ, j1 - jt));
    __syncthreads();  // previous tile fully consumed
    // stage: row r of the tile = [Q(16) | V(16) | X(32)], 16 float4 per row, 512 float4 per tile
#pragma unroll
    for (int rep = 0; rep < TJ * ROWF / 4 / 64; ++rep) {
      const int idx = rep * 64 + lane;
      const int r = idx >> 4, part = idx & 15;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < cnt) {
        const long row = gbase + jt + r;
        const float *src = part < 4 ? Qm + row * D + h * HD + 4 * part
                         : part < 8 ? Vm + row * D + h * HD + 4 * (part - 4)
                                    : tok + row * D + 4 * (part - 8);
        v = *reinterpret_cast<const float4 *>(src);
      }
      *reinterpret_cast<float4 *>(stage + r * ROWF + 4 * part) = v;
    }
    __syncthreads();
    for (int jj = 0; jj < cnt; ++jj) {
      const float4 *rowp = reinterpret_cast<const float4 *>(stage + jj * ROWF);
      float s = 0.f, da = 0.f;
#pragma unroll
      for (int q = 0; q < HD / 4; ++q) {
        const float4 qv = rowp[q], vv = rowp[4 + q];
        s += Ki[4 * q] * qv.x + Ki[4 * q + 1] * qv.y + Ki[4 * q + 2] * qv.z + Ki[4 * q + 3] * qv.w;
        da += dh[4 * q] * vv.x + dh[4 * q + 1] * vv.y + dh[4 * q + 2] * vv.z + dh[4 * q + 3] * vv.w;
      }
      const float ev = __expf(s * scale - mx);
      const float ed = ev * da;
      ssum += ev;
      tsum += ed;
#pragma unroll
      for (int q = 0; q < D / 4; ++q) {
        const float4 xv = rowp[8 + q];
        Yt[4 * q] += ev * xv.x; Yt[4 * q + 1] += ev * xv.y; Yt[4 * q + 2] += ev * xv.z; Yt[4 * q + 3] += ev * xv.w;
        Wt[4 * q] += ed * xv.x; Wt[4 * q + 1] += ed * xv.y; Wt[4 * q + 2] += ed * xv.z; Wt[4 * q + 3] += ed * xv.w;
      }
    }
  }
  __syncthreads();  // staging region is reused below
  if (wave > 0) {
    float *r = smem + ((wave - 1) * 64 + lane) * (2 * D + 3);
#pragma unroll
    for (int d = 0; d < D; ++d) { r[d] = Yt[d]; r[D + d] = Wt[d]; }
    r[2 * D] = ssum;
    r[2 * D + 1] = tsum;
  }
  __syncthreads();
  // wave 0 merges the four j ranges and finishes its 64 rows
  if (wave == 0) {
    float s = ssum, t = tsum;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) {
      const float *r = smem + (w * 64 + lane) * (2 * D + 3);
#pragma unroll
      for (int d = 0; d < D; ++d) { Yt[d] += r[d]; Wt[d] += r[D + d]; }
      s += r[2 * D];
      t += r[2 * D + 1];
    }
    const float invs = ok ? 1.0f / s : 0.f;
    const float dot = t * invs;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      Yt[d] = Yt[d] * invs;                // Y_i
      Wt[d] = Wt[d] * invs - dot * Yt[d];  // Z_i
    }
    // dK_i = Z_i . Wq_h   (delta_scores . query with query = X.Wq)
    const float *__restrict__ Wq_h = Wqkv + (size_t)(h * 3 + 0) * D * HD;
    float dK[HD], Xi[D];
#pragma unroll
    for (int e = 0; e < HD; ++e) dK[e] = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      Xi[d] = ok ? tok[ri * D + d] : 0.f;
#pragma unroll
      for (int e = 0; e < HD; ++e) dK[e] += Wt[d] * Wq_h[d * HD + e];
    }
    if (!ok) {
#pragma unroll
      for (int e = 0; e < HD; ++e) { Ki[e] = 0.f; dh[e] = 0.f; dK[e] = 0.f; }
    }
    // outer products summed over the 64 rows with the transposing butterfly, 64 outputs per call
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
      for (int blk = 0; blk < D / 4; ++blk) {
        float acc[64];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd)
#pragma unroll
          for (int e = 0; e < HD; ++e) {
            const int d = blk * 4 + dd;
            acc[dd * HD + e] = (m == 0) ? Wt[d] * Ki[e] : (m == 1) ? Xi[d] * dK[e] : Yt[d] * dh[e];
          }
        wave_xreduce<64>(acc);
        sout[m * D * HD + blk * 64 + lane] = acc[0];
      }
    }
  }
  __syncthreads();
  float *row = scratch + (size_t)blockIdx.x * MH_STRIDE;
  for (int idx = threadIdx.x; idx < MH_STRIDE; idx += TPB) row[idx] = sout[idx];
}

}  // namespace vt

using namespace vt;

extern "C" int vt_reduce_rows(const float *slab, long rows, int cols, int ld, float *out, vt_stream s);

// g_qkv: [q0 | k0 | v0 | q1 | k1 | v1] gradients (arena order);
// scratch >= heads * groups * ceil(N/64) * 1536 floats.
extern "C" int vt_mha_bwd(const float *qkv, const float *dc, const float *tok, const float *wqkv, float *g_qkv,
                          float *scratch, int N, int groups, vt_stream s) {
  if (N < 1 || groups < 1) return VT_EINVAL;
  const int tiles = (N + 63) / 64;
  const long rows = (long)N * groups;
  hipLaunchKernelGGL(k_mha_bwd, dim3(HEADS * groups * tiles), dim3(TPB), 0, (hipStream_t)s, qkv, dc, tok, wqkv, scratch,
                     N, rows, groups);
  VT_CHECK_LAUNCH();
  // workgroups are ordered (head, group, tile): the partials of one head are contiguous rows
  for (int h = 0; h < HEADS; ++h) {
    const int rc = vt_reduce_rows(scratch + (size_t)h * groups * tiles * MH_STRIDE, (long)groups * tiles, MH_STRIDE,
                                  MH_STRIDE, g_qkv + (size_t)h * MH_STRIDE, s);
    if (rc) return rc;
  }
  return 0;
}
