// Per-pixel kernels of the VTCNN hot path on the gfx950 matrix cores.
//
// The per-pixel contractions of the tokenizer and the projector are 8-, 16- and 32-wide: too narrow
// to be GEMMs worth a library call, but on the vector ALUs every multiply needs a wave-uniform
// weight operand (scalar registers or LDS broadcasts), and delivering those operands, not the
// arithmetic or the HBM traffic, bounds the kernels.  v_mfma_f32_16x16x4_f32 removes the problem:
// the weights sit distributed over the lanes of a few vector registers, a tile of 16 pixels is the
// other operand, accumulation is exact fp32, and the matrix pipe runs beside the vector pipe that
// does the softmax.
//
// Register layouts of v_mfma_f32_16x16x4_f32 (lane = 16*g + j, g = 0..3, j = 0..15):
//   A[i][k]: lane holds i = j, k = g            B[k][j']: lane holds j' = j, k = g
//   D[i][j']: register v of the lane holds i = 4*g + v, j' = j
// Two facts are used throughout:
//   * a contraction may visit its terms in any order, so a D tile ("row 4g+v in register v") is
//     directly the B (or A) operand of the NEXT product if step v of that product is defined to
//     contract over the rows {4g+v : g = 0..3} and the other operand is permuted to match;
//   * the same register is an A operand (pixel = output row) or a B operand (pixel = output column).
// "T layout" below: lane = pixel j, registers = features 4g+v.  "N layout": lane = feature j,
// registers = pixels 4g+v (what a contraction over pixels needs).
#include "vtcnn_common.h"
#include "vtcnn_tail.h"

namespace vt {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int TK_MAXT = 13;  // tiles per wave: P <= 16 * NW * TK_MAXT = 832

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// A 32-deep fp32 contraction on the bf16 matrix pipe: every operand is carried as bf16 hi = bf16(x) and
// lo = bf16(x - hi) (16 mantissa bits together) and the product is hi.hi + hi.lo + lo.hi, three
// v_mfma_f32_16x16x32_bf16 (16 cycles each) instead of eight v_mfma_f32_16x16x4_f32 (32 cycles each), with
// fp32 accumulation; the relative error per product is 2^-17.  A lane holds 8 consecutive contraction
// slots of its lane group (k = 8g .. 8g+7) in both operands.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float floatx2_t __attribute__((ext_vector_type(2)));
struct Split8 {
  bf16x8_t hi, lo;
};
__device__ __forceinline__ void split_pair(float a, float b, unsigned &hi, unsigned &lo) {
  const floatx2_t f = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_t));
  const floatx2_t r = f - floatx2_t{__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
}
__device__ __forceinline__ Split8 split8(const float (&x)[8]) {
  uint4 h, l;
  split_pair(x[0], x[1], h.x, l.x);
  split_pair(x[2], x[3], h.y, l.y);
  split_pair(x[4], x[5], h.z, l.z);
  split_pair(x[6], x[7], h.w, l.w);
  return Split8{__builtin_bit_cast(bf16x8_t, h), __builtin_bit_cast(bf16x8_t, l)};
}
struct Split4 {  // four values: packed bf16 hi and lo halves (two 32-bit words each)
  uint2 hi, lo;
};
__device__ __forceinline__ Split4 split4(const f4 &x) {
  Split4 r;
  split_pair(x[0], x[1], r.hi.x, r.lo.x);
  split_pair(x[2], x[3], r.hi.y, r.lo.y);
  return r;
}
// operand of a 32-deep product from two groups of four slots (either may be all zero)
__device__ __forceinline__ Split8 join8(const Split4 &a, const Split4 &b) {
  return Split8{__builtin_bit_cast(bf16x8_t, make_uint4(a.hi.x, a.hi.y, b.hi.x, b.hi.y)),
                __builtin_bit_cast(bf16x8_t, make_uint4(a.lo.x, a.lo.y, b.lo.x, b.lo.y))};
}
__device__ __forceinline__ f4 mfma32_split(const Split8 &a, const Split8 &b, f4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, c, 0, 0, 0);  // smallest partial products first
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, c, 0, 0, 0);
}

// Forces a value loaded in a kernel's prologue to have arrived (the compiler places the wait for
// its load here).  Without it the wait lands at the first use inside the tile loop, behind that
// tile's stores, and since stores share the loads' counter the wave would then sit out the
// acknowledgement of its own stores once per tile.
template <class T>
__device__ __forceinline__ void settle(T &x) { asm volatile("" : "+v"(x)); }
template <class T, int N>
__device__ __forceinline__ void settle(T (&x)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(x[i]));
}
__device__ __forceinline__ void settle(f4 &x) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t = x[i];
    asm volatile("" : "+v"(t));
    x[i] = t;
  }
}

// sum / max over the four lane groups g (lanes j, j+16, j+32, j+48)
__device__ __forceinline__ float gsum(float x) {
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}
__device__ __forceinline__ float gmax(float x) {
  x = fmaxf(x, __shfl_xor(x, 16));
  x = fmaxf(x, __shfl_xor(x, 32));
  return x;
}
// sum over the 16 lanes j of one lane group
__device__ __forceinline__ float jsum(float x) {
  x += __shfl_xor(x, 1);
  x += __shfl_xor(x, 2);
  x += __shfl_xor(x, 4);
  x += __shfl_xor(x, 8);
  return x;
}

// BatchNorm2D apply for the two channels 2g, 2g+1 of pixel j (x_norm * gamma + beta,
// batch_normalization_layer.hpp:279,296); same operation order as load_x.
struct BnPair {
  float mean[2], inv[2], gam[2], bet[2];
  __device__ __forceinline__ void init(const float *st, const float *g, const float *b, int gq) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ch = 2 * gq + s;
      mean[s] = st[ch];
      inv[s] = st[C1 + ch];
      gam[s] = g[ch];
      bet[s] = b[ch];
    }
  }
  __device__ __forceinline__ void settle_all() {
    settle(mean);
    settle(inv);
    settle(gam);
    settle(bet);
  }
  __device__ __forceinline__ float apply(float raw, int s) const { return ((raw - mean[s]) * inv[s]) * gam[s] + bet[s]; }
};

// ---------------------------------------------------------------------------------------------
// Pixel side of ProjectorLayer::forward (reference layers/projector_layer.hpp:65-104).
//   a = softmax_L(x.Mq);  z = (x.Wd + bd) + a.v    (+ per-sample partial statistics of z)
// One workgroup per image, one wave per 16-pixel tile.  Per tile: 2 MFMAs for the scores (T
// layout, so the softmax over the 16 tokens is 4 registers x 4 lane groups), 2 for the downsample
// and 4 for a.v, the softmax output feeding the last product straight from its registers.
// The partial statistics are accumulated around a pivot (the layer's running mean: any value near
// the batch mean works):  M2 = sum (z-piv)^2 - (sum (z-piv))^2 / P.
// projA (the attention map) is written only when the pointer is non-null (debug / tests); the
// backward pass recomputes it.
// ---------------------------------------------------------------------------------------------
template <bool TAIL>  // TAIL: the last workgroup closes the projector BatchNorm1D statistics
__global__ __launch_bounds__(TPB) void k_proj_fwd(const float *__restrict__ conv, const float *__restrict__ st,
                                                  const float *__restrict__ g, const float *__restrict__ b,
                                                  const float *__restrict__ pk, const float *__restrict__ Wd,
                                                  const float *__restrict__ bd, const float *__restrict__ piv,
                                                  float *__restrict__ projA, float *__restrict__ z,
                                                  float *__restrict__ part_pn, int P, int sgroup, const BnFwdTail tail) {
  __shared__ float red[NW][2 * C2];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, gq = lane >> 4;
  if (sgroup > 0) st += (size_t)(n / sgroup) * 2 * C1;  // BatchNorm2D statistics of this image's mini-batch
  const float *__restrict__ mq = pk + (size_t)n * PK_STRIDE;
  const float *__restrict__ vv = mq + C1 * L;
  float amq[2], awd[2], av[4], pv[4];
  f4 cinit;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    amq[s] = mq[(2 * gq + s) * L + j];   // A[token j][channel 2g+s]
    awd[s] = Wd[(2 * gq + s) * C2 + j];  // A[c2 j][channel 2g+s]
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    av[v] = vv[(4 * gq + v) * C2 + j];  // A[c2 j][token 4g+v]
    pv[v] = piv[4 * gq + v];
    cinit[v] = bd[4 * gq + v];
  }
  BnPair bn;
  bn.init(st, g, b, gq);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int ntiles = (P + 15) >> 4;
  // The wave's tiles are independent, and with only four waves per SIMD resident the HBM latency
  // of a tile's load would be paid once per tile: all loads of the wave are issued up front
  // (unconditional, clamped addresses; 2 registers per tile), the arithmetic then drains them.
  const float *__restrict__ xin = conv + (size_t)n * P * C1 + 2 * gq;
  float2 raws[TK_MAXT];
#pragma unroll
  for (int i = 0; i < TK_MAXT; ++i)
    raws[i] = *reinterpret_cast<const float2 *>(xin + (size_t)min((wave + NW * i) * 16 + j, P - 1) * C1);
  settle(amq);
  settle(pv);
  settle(cinit);
  bn.settle_all();
#pragma unroll
  for (int i = 0; i < TK_MAXT; ++i) {
    settle(raws[i].x);
    settle(raws[i].y);
  }
  // constant operand of the split product below (formed after every load of the prologue is under way)
  const Split8 wz = join8(split4(f4{awd[0], awd[1], 0.f, 0.f}), split4(f4{av[0], av[1], av[2], av[3]}));
#pragma unroll
  for (int i = 0; i < TK_MAXT; ++i) {
    const int t = wave + NW * i;
    if (t >= ntiles) break;
    const int p = t * 16 + j;
    const bool ok = p < P;
    const float2 raw = raws[i];
    const float x0 = ok ? bn.apply(raw.x, 0) : 0.f, x1 = ok ? bn.apply(raw.y, 1) : 0.f;
    f4 sc = {0.f, 0.f, 0.f, 0.f};
    sc = mfma16(amq[0], x0, sc);
    sc = mfma16(amq[1], x1, sc);
    const float mx = gmax(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])));
    f4 a;
#pragma unroll
    for (int v = 0; v < 4; ++v) a[v] = __expf(sc[v] - mx);
    const float rsum = __builtin_amdgcn_rcpf(gsum((a[0] + a[1]) + (a[2] + a[3])));
#pragma unroll
    for (int v = 0; v < 4; ++v) a[v] = a[v] * rsum;
    // z = bd + Wd . x + v . a: one 24-deep contraction (slots 0..1 = channels 2g, 2g+1, slots 4..7 = tokens
    // 4g..4g+3) as bf16 hi/lo partial products, 3 matrix instructions of 16 cycles instead of 6 of 32
    Split4 sx;
    split_pair(x0, x1, sx.hi.x, sx.lo.x);
    sx.hi.y = 0u;
    sx.lo.y = 0u;
    const f4 zc = mfma32_split(wz, join8(sx, split4(a)), cinit);
    if (ok) {
      if (projA) *reinterpret_cast<f4 *>(projA + ((size_t)n * P + p) * L + 4 * gq) = a;
      *reinterpret_cast<f4 *>(z + ((size_t)n * P + p) * C2 + 4 * gq) = zc;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float dlt = zc[v] - pv[v];
        s1[v] += dlt;
        s2[v] += dlt * dlt;
      }
    }
  }
  if (part_pn == nullptr) return;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    s1[v] = jsum(s1[v]);
    s2[v] = jsum(s2[v]);
  }
  if (j == 0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      red[wave][4 * gq + v] = s1[v];
      red[wave][C2 + 4 * gq + v] = s2[v];
    }
  }
  __syncthreads();
  if (tid < C2) {
    float d1 = 0.f, d2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      d1 += red[w][tid];
      d2 += red[w][C2 + tid];
    }
    const float rs = d1 + (float)P * piv[tid], rq = fmaxf(d2 - d1 * d1 / (float)P, 0.f);
    if constexpr (!TAIL) {
      part_pn[(size_t)n * 2 * C2 + tid] = rs;
      part_pn[(size_t)n * 2 * C2 + C2 + tid] = rq;
    } else {
      tail_store(part_pn + (size_t)n * 2 * C2 + tid, rs);
      tail_store(part_pn + (size_t)n * 2 * C2 + C2 + tid, rq);
    }
  }
  if constexpr (TAIL) bn_fwd_tail(tail, part_pn, n, gridDim.x, C2);
}


// ---------------------------------------------------------------------------------------------
// ProjectorLayer::backward, pixel side (reference layers/projector_layer.hpp:133-184, 204-213).
// Per pixel (dz = BatchNorm1D input delta, a = softmax_L(x.Mq) recomputed):
//   dx  = dz . Wd^T                                   downsample Dense backward
//   da  = dz . v^T;  ds = a * (da - sum_l da*a)       softmax_backward over tokens (tensor_cpu.hpp:946-950)
//   dx += ds . Mq^T                                   (= x_q_delta . Wq^T with Mq = Wq k^T folded)
// Per-sample sums over pixels:
//   G[c][l]   = sum_p x[c] ds[l]      ->  dk = G^T Wq (token key delta),  gWq_n = G k
//   dv[l][c2] = sum_p a[l] dz[c2]     (value delta)
//   gWd_n     = sum_p x^T dz,  gbd_n = sum_p dz
// gslab record per image: gWq_n[C1*D] | gWd_n[C1*C2] | gbd_n[C2]  (GS_STRIDE floats).
//
// One wave per 16-pixel tile.  The per-pixel part runs in T layout (26 MFMAs per tile in total):
// scores (2), da (4), dx (8, the token and the c2 contraction chained into one accumulator).
// The sums over pixels need the pixel as the contraction index, i.e. x, a, dz, ds in N layout:
// the wave passes the four tiles through its private LDS patch (row stride 60 floats: the b128
// writes of 8 consecutive lanes cover all banks) and accumulates G, gWd|gbd (a row of ones in the
// x operand yields the bias gradient) and dv in three D tiles that stay in registers over all
// tiles of the wave; the four waves are summed through LDS at the end.
// ---------------------------------------------------------------------------------------------
constexpr int GS_STRIDE = C1 * D + C1 * C2 + C2;  // 400
constexpr int PB_RS = 60;                          // LDS row stride of the transposition patch
constexpr int PB_X = 0, PB_A = 8, PB_DZ = 24, PB_DS = 40;
// WIN: du is not the dense [N,P,C2] delta but one float per horizontally adjacent element pair, [N, P*C2/2]
// (k_head_bwd_win): zero, or the delta of the pair's pooling window with the arg-max column in its lowest
// mantissa bit.  A thread's four elements 4g..4g+3 of a pixel record are two such pairs, one 8-byte load.
template <bool WIN>
__global__ __launch_bounds__(TPB) void k_proj_bwd(const float *__restrict__ du, const float *__restrict__ coef,
                                                  const float *__restrict__ conv, const float *__restrict__ st,
                                                  const float *__restrict__ g, const float *__restrict__ b,
                                                  const float *__restrict__ pk, const float *__restrict__ Wq,
                                                  const float *__restrict__ Wd, float *__restrict__ pdx,
                                                  float *__restrict__ dv_out, float *__restrict__ dk_out,
                                                  float *__restrict__ gslab, int P, int sgroup) {
  __shared__ __attribute__((aligned(16))) float patch[NW][16 * PB_RS];
  __shared__ float sred[NW][3][256];
  __shared__ float sG[C1 * L];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, gq = lane >> 4;
  if (sgroup > 0) {  // statistics and backward coefficients of this image's mini-batch
    st += (size_t)(n / sgroup) * 2 * C1;
    coef += (size_t)(n / sgroup) * 2 * C2;
  }
  const float *__restrict__ mq = pk + (size_t)n * PK_STRIDE;
  const float *__restrict__ vv = mq + C1 * L;
  const float *__restrict__ kk = vv + L * C2;
  float amq[2], av[4], awd[4], amt[4], cf0[4], cf1[4];
#pragma unroll
  for (int s = 0; s < 2; ++s) amq[s] = mq[(2 * gq + s) * L + j];  // A[token j][channel 2g+s]
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    av[v] = vv[j * C2 + 4 * gq + v];                       // A[token j][c2 4g+v]
    awd[v] = j < C1 ? Wd[j * C2 + 4 * gq + v] : 0.f;       // A[channel j][c2 4g+v]
    amt[v] = j < C1 ? mq[j * L + 4 * gq + v] : 0.f;        // A[channel j][token 4g+v]
    cf0[v] = coef[4 * gq + v];
    cf1[v] = coef[C2 + 4 * gq + v];
  }
  // constant operands of the two split products: slots 0..3 = c2 4g..4g+3, slots 4..7 = tokens 4g..4g+3
  const Split4 zero4 = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
  const Split8 wda = join8(split4(f4{av[0], av[1], av[2], av[3]}), zero4);
  const Split8 wdx = join8(split4(f4{awd[0], awd[1], awd[2], awd[3]}), split4(f4{amt[0], amt[1], amt[2], amt[3]}));
  BnPair bn;
  bn.init(st, g, b, gq);
  f4 accG = {0.f, 0.f, 0.f, 0.f}, accW = accG, accV = accG;
  float *pt = patch[wave];
  const int ntiles = (P + 15) >> 4;
  // The operands of a tile are requested one trip ahead (the loads of tile t + NW are in flight during the
  // arithmetic of tile t); rows past the image re-read the last pixel and are masked where they are used.
  struct Fetch {
    float2 raw;
    f4 u;       // dense delta (WIN == false)
    float2 e2;  // the two pair entries (WIN == true)
  };
  auto fetch = [&](int t) {
    Fetch f;
    const int p = min(t * 16 + j, P - 1);
    f.raw = *reinterpret_cast<const float2 *>(conv + ((size_t)n * P + p) * C1 + 2 * gq);
    if (WIN) f.e2 = *reinterpret_cast<const float2 *>(du + ((size_t)n * P + p) * (C2 / 2) + 2 * gq);
    else f.u = *reinterpret_cast<const f4 *>(du + ((size_t)n * P + p) * C2 + 4 * gq);
    return f;
  };
  Fetch nxt = fetch(wave);
  for (int t = wave; t < ntiles; t += NW) {
    const int p = t * 16 + j;
    const bool ok = p < P;
    const Fetch cur = nxt;
    if (t + NW < ntiles) nxt = fetch(t + NW);
    float x0 = 0.f, x1 = 0.f;
    f4 dz = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      f4 u;
      if (WIN) {
        const unsigned b0 = __float_as_uint(cur.e2.x), b1 = __float_as_uint(cur.e2.y);
        const float v0 = __uint_as_float(b0 & ~1u), v1 = __uint_as_float(b1 & ~1u);
        u[0] = (b0 & 1u) ? 0.f : v0;
        u[1] = (b0 & 1u) ? v0 : 0.f;
        u[2] = (b1 & 1u) ? 0.f : v1;
        u[3] = (b1 & 1u) ? v1 : 0.f;
      } else {
        u = cur.u;
      }
      x0 = bn.apply(cur.raw.x, 0);
      x1 = bn.apply(cur.raw.y, 1);
#pragma unroll
      for (int v = 0; v < 4; ++v) dz[v] = u[v] * cf0[v] + cf1[v];
    }
    f4 sc = {0.f, 0.f, 0.f, 0.f};
    sc = mfma16(amq[0], x0, sc);
    sc = mfma16(amq[1], x1, sc);
    const float mx = gmax(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])));
    f4 a;
#pragma unroll
    for (int v = 0; v < 4; ++v) a[v] = __expf(sc[v] - mx);
    const float rsum = __builtin_amdgcn_rcpf(gsum((a[0] + a[1]) + (a[2] + a[3])));
#pragma unroll
    for (int v = 0; v < 4; ++v) a[v] = a[v] * rsum;
    // da = v . dz (16 deep) and, below, dx = Wd . dz + Mq . ds (32 deep) as bf16 hi/lo partial products:
    // 6 matrix instructions of 16 cycles instead of 12 of 32; dz is split once for both
    const Split4 sdz = split4(dz);
    f4 da = mfma32_split(wda, join8(sdz, zero4), f4{0.f, 0.f, 0.f, 0.f});
    const float dot = gsum((da[0] * a[0] + da[1] * a[1]) + (da[2] * a[2] + da[3] * a[3]));
    f4 ds;
#pragma unroll
    for (int v = 0; v < 4; ++v) ds[v] = a[v] * (da[v] - dot);
    const f4 dx = mfma32_split(wdx, join8(sdz, split4(ds)), f4{0.f, 0.f, 0.f, 0.f});
    if (ok && gq < 2) *reinterpret_cast<f4 *>(pdx + ((size_t)n * P + p) * C1 + 4 * gq) = dx;
    // T -> N through the wave's LDS patch
    *reinterpret_cast<float2 *>(pt + j * PB_RS + PB_X + 2 * gq) = make_float2(x0, x1);
    *reinterpret_cast<f4 *>(pt + j * PB_RS + PB_A + 4 * gq) = a;
    *reinterpret_cast<f4 *>(pt + j * PB_RS + PB_DZ + 4 * gq) = dz;
    *reinterpret_cast<f4 *>(pt + j * PB_RS + PB_DS + 4 * gq) = ds;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float xn[4], an[4], dzn[4], dsn[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float *row = pt + (4 * gq + v) * PB_RS;
      xn[v] = j < C1 ? row[PB_X + j] : (j == C1 ? 1.f : 0.f);
      an[v] = row[PB_A + j];
      dzn[v] = row[PB_DZ + j];
      dsn[v] = row[PB_DS + j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      accG = mfma16(xn[v], dsn[v], accG);  // [channel | .][token]
      accW = mfma16(xn[v], dzn[v], accW);  // [channel | ones][c2]
      accV = mfma16(an[v], dzn[v], accV);  // [token][c2]
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    sred[wave][0][(4 * gq + v) * 16 + j] = accG[v];
    sred[wave][1][(4 * gq + v) * 16 + j] = accW[v];
    sred[wave][2][(4 * gq + v) * 16 + j] = accV[v];
  }
  __syncthreads();
  float *srow = gslab + (size_t)n * GS_STRIDE;
  for (int o = tid; o < 3 * 256; o += TPB) {
    const int which = o >> 8, idx = o & 255, r = idx >> 4, c = idx & 15;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += sred[w][which][idx];
    if (which == 0) {
      if (r < C1) sG[r * L + c] = s;                       // G[channel r][token c]
    } else if (which == 1) {
      if (r < C1) srow[C1 * D + r * C2 + c] = s;           // gWd_n[channel r][c2]
      else if (r == C1) srow[C1 * D + C1 * C2 + c] = s;    // gbd_n[c2]
    } else {
      dv_out[((size_t)n * L + r) * C2 + c] = s;            // dv[token r][c2]
    }
  }
  __syncthreads();
  // token key delta dk[l][d] = sum_c G[c][l] Wq[c][d]  (t_q_delta = x_q^T . softmax_delta, :183-184)
  for (int o = tid; o < L * D; o += TPB) {
    const int l = o / D, d = o % D;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C1; ++c) s += sG[c * L + l] * Wq[c * D + d];
    dk_out[(size_t)n * L * D + o] = s;
  }
  // gWq_n[c][d] = sum_l G[c][l] k[l][d]  (linear1 grad = x^T . (softmax_delta . t_q), :180, 205-206)
  for (int o = tid; o < C1 * D; o += TPB) {
    const int c = o / D, d = o % D;
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) s += sG[c * L + l] * kk[l * D + d];
    srow[o] = s;
  }
}


// ---------------------------------------------------------------------------------------------
// FilterTokenizer::forward (reference layers/tokenizer.hpp:22-48) fused with the BatchNorm2D apply
// and the NCHW->[N,P,C] transpose (model/vit_cnn.hpp:127-134).  One workgroup per image.
//   S = x.W1 + b1;  A = softmax over the P pixels per token;  T0 = A^T.x;  t = T0.W2 + b2
// S is produced in N layout (lane = token, registers = pixels), so the softmax over pixels reduces
// inside the lane first and crosses lanes once per workgroup.  The image is read once: a wave fetches
// all of its tiles together, keeps their score tiles in registers (52 per lane, the kernel stays
// under 128) and the normalised pixels in LDS;
//   pass 1: scores and the per-token maximum;   pass 2: e = exp(S - max), sum e, and T0~ = e^T.x
//   accumulated in D tiles per wave; T0 = T0~ / sum at the end.
// (The kernel is bound by its vector instructions, not by bytes: recomputing S and re-reading x in a
// second sweep cost 40 % more of them.)
// tok_stat[n] = (max_l[L], sum_l[L]) lets the backward pass rebuild A in a single sweep; tokA is
// written only when non-null (debug / tests), by a third sweep.
// ---------------------------------------------------------------------------------------------
template <bool WITH_MAP, bool TAIL>  // WITH_MAP: also write the attention map tokA (tests); its stores cost registers the step cannot spare.  TAIL: the last workgroup closes norm1's statistics
__global__ __launch_bounds__(TPB, 4) void k_tokenizer_fwd(const float *__restrict__ conv, const float *__restrict__ st,
                                                       const float *__restrict__ g, const float *__restrict__ b,
                                                       const float *__restrict__ W1, const float *__restrict__ b1,
                                                       const float *__restrict__ W2, const float *__restrict__ b2,
                                                       float *__restrict__ tokA, float *__restrict__ tok_stat,
                                                       float *__restrict__ tok, float *__restrict__ part_n1, int P,
                                                       int sgroup, const BnFwdTail tail) {
  __shared__ float redm[NW][L], reds[NW][L];
  __shared__ float sT[NW][256];
  __shared__ float sT0[L * C1];
  __shared__ float st_[L * D];
  // normalised pixels of the whole image, [pixel][TKX] with a pad column (the second product reads
  // them with the pixel in the registers, four rows apart per lane group: stride 9 keeps those reads
  // and the 16 lanes' writes on distinct banks)
  constexpr int TKX = C1 + 1;
  __shared__ float sx[16 * NW * TK_MAXT * TKX];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, gq = lane >> 4;
  const int ntiles = (P + 15) >> 4;
  if (sgroup > 0) st += (size_t)(n / sgroup) * 2 * C1;  // BatchNorm2D statistics of this image's mini-batch
  float bw[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) bw[s] = W1[(2 * gq + s) * L + j];  // B[channel 2g+s][token j]
  const float bias = b1[j];
  BnPair bn;
  bn.init(st, g, b, gq);
  const float *__restrict__ xin = conv + (size_t)n * P * C1 + 2 * gq;
  // One pass over the image: every tile of the wave (t = wave + NW*u) is fetched at once, normalised,
  // left in LDS for the second product, and its score tile S = x.W1 + b1 stays in registers
  // (N layout: lane = token j, registers = pixels 4g+v), so nothing is recomputed or re-read.
  float2 raw[TK_MAXT];
#pragma unroll
  for (int u = 0; u < TK_MAXT; ++u)
    raw[u] = *reinterpret_cast<const float2 *>(xin + (size_t)min((wave + NW * u) * 16 + j, P - 1) * C1);
  f4 sc[TK_MAXT];
  float cm = -INFINITY;
#pragma unroll
  for (int u = 0; u < TK_MAXT; ++u) {
    const int t = wave + NW * u;
    const bool ok = t < ntiles && t * 16 + j < P;
    const float x0 = ok ? bn.apply(raw[u].x, 0) : 0.f, x1 = ok ? bn.apply(raw[u].y, 1) : 0.f;
    sx[(t * 16 + j) * TKX + 2 * gq] = x0;
    sx[(t * 16 + j) * TKX + 2 * gq + 1] = x1;
    sc[u] = f4{bias, bias, bias, bias};
    sc[u] = mfma16(x0, bw[0], sc[u]);
    sc[u] = mfma16(x1, bw[1], sc[u]);
    // rows past the image leave the softmax here: exp(-inf - max) = 0, and the second pass needs no masks
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const bool valid = t < ntiles && t * 16 + 4 * gq + v < P;
      sc[u][v] = valid ? sc[u][v] : -INFINITY;
      cm = fmaxf(cm, sc[u][v]);
    }
    __builtin_amdgcn_sched_barrier(0);  // one tile at a time: see the note in the second pass
  }
  // maximum over the pixels, per token (lane j)
  cm = gmax(cm);
  if (gq == 0) redm[wave][j] = cm;
  __syncthreads();
  cm = fmaxf(fmaxf(redm[0][j], redm[1][j]), fmaxf(redm[2][j], redm[3][j]));
  // e = exp(S - max), its sum, and T0~[l][c] = sum_p e[p][l] x[p][c] (D[token][channel]); two
  // accumulators so that consecutive matrix instructions do not wait for each other
  const int ch = j & (C1 - 1);
  float cs = 0.f;
  f4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = acc;
#pragma unroll
  for (int u = 0; u < TK_MAXT; ++u) {
    const int t = wave + NW * u;
    if (t < ntiles) {  // wave-uniform
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float e = __expf(sc[u][v] - cm);
        const float xn = j < C1 ? sx[(t * 16 + 4 * gq + v) * TKX + ch] : 0.f;  // rows past P hold zeros
        cs += e;
        if (v & 1) acc1 = mfma16(e, xn, acc1);
        else acc = mfma16(e, xn, acc);
      }
    }
    // keep the instruction scheduler from hoisting every tile's LDS reads to the top of the pass
    // (52 more live registers would halve the occupancy)
    if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) acc[v] += acc1[v];
  cs = gsum(cs);
  if (gq == 0) reds[wave][j] = cs;
#pragma unroll
  for (int v = 0; v < 4; ++v) sT[wave][(4 * gq + v) * 16 + j] = acc[v];
  __syncthreads();
  cs = (reds[0][j] + reds[1][j]) + (reds[2][j] + reds[3][j]);
  if (tid < L) {
    tok_stat[(size_t)n * 2 * L + tid] = cm;
    tok_stat[(size_t)n * 2 * L + L + tid] = cs;
  }
  {
    const int l = tid >> 4, c = tid & 15;
    const float tot = (reds[0][l] + reds[1][l]) + (reds[2][l] + reds[3][l]);
    if (c < C1) sT0[l * C1 + c] = ((sT[0][tid] + sT[1][tid]) + (sT[2][tid] + sT[3][tid])) * (1.0f / tot);
  }
  if (WITH_MAP) {  // debug output: the attention map itself
    const float rcs = 1.0f / cs;
#pragma unroll
    for (int u = 0; u < TK_MAXT; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int t = wave + NW * u, pr = t * 16 + 4 * gq + v;
        if (t < ntiles && pr < P) tokA[((size_t)n * P + pr) * L + j] = __expf(sc[u][v] - cm) * rcs;
      }
  }
  __syncthreads();
  // t = T0 . W2 + b2  (linear2, tokenizer.hpp:42-44)
  for (int o = tid; o < L * D; o += TPB) {
    const int l = o / D, d = o % D;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C1; ++c) s += sT0[l * C1 + c] * W2[c * D + d];
    s += b2[d];
    st_[o] = s;
    tok[(size_t)n * L * D + o] = s;
  }
  __syncthreads();
  if (tid < D && part_n1 != nullptr) {  // per-sample partial statistics for the transformer's norm1
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) s += st_[l * D + tid];
    const float mu = s * (1.0f / L);
    float q = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float dlt = st_[l * D + tid] - mu;
      q += dlt * dlt;
    }
    if constexpr (!TAIL) {
      part_n1[(size_t)n * 2 * D + tid] = s;
      part_n1[(size_t)n * 2 * D + D + tid] = q;
    } else {
      tail_store(part_n1 + (size_t)n * 2 * D + tid, s);
      tail_store(part_n1 + (size_t)n * 2 * D + D + tid, q);
    }
  }
  if constexpr (TAIL) bn_fwd_tail(tail, part_n1, n, gridDim.x, D);
}

// ---------------------------------------------------------------------------------------------
// FilterTokenizer::backward2 (reference layers/tokenizer.hpp:50-107).  Per pixel:
//   dA[l] = dT0[l] . x;  dx = A . dT0;  dS = A * (dA - sum_l dA*A)   (softmax_backward reduces over
//   the token axis although the forward softmax ran over pixels, tokenizer.hpp:36,91)
//   dx += dS . W1^T;  dx += projector input delta                         (tokenizer.hpp:97,105)
// The tokenizer's own Dense gradients are never applied by the reference (no optimizer is ever
// attached to them, tokenizer.hpp:18-19) and are not computed.  Fused: partial sums of the
// BatchNorm2D backward over the result (vit_cnn.hpp:236-238).
// T layout throughout, 12 MFMAs per 16-pixel tile: S (2) and dA (2) share the x operand, dx chains
// the two token contractions onto the projector's input delta loaded as the accumulator (8).
// A is rebuilt from the per-sample (max, sum) the forward pass left in tok_stat.
// ---------------------------------------------------------------------------------------------
template <bool TAIL>  // TAIL: the last workgroup closes the BatchNorm2D backward reduction
__global__ __launch_bounds__(TPB) void k_tokenizer_bwd(const float *__restrict__ conv, const float *__restrict__ st,
                                                       const float *__restrict__ g, const float *__restrict__ b,
                                                       const float *__restrict__ tok_stat, const float *__restrict__ dT0,
                                                       const float *__restrict__ W1, const float *__restrict__ b1,
                                                       const float *__restrict__ pdx, float *__restrict__ dy,
                                                       float *__restrict__ part, int P, int sgroup, const BnBwdTail tail) {
  __shared__ float red[NW][2 * C1];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, gq = lane >> 4;
  if (sgroup > 0) st += (size_t)(n / sgroup) * 2 * C1;  // BatchNorm2D statistics of this image's mini-batch
  const float *__restrict__ dt = dT0 + (size_t)n * L * C1;
  const float *__restrict__ ts = tok_stat + (size_t)n * 2 * L;
  float adt[2], aw1[2], a1[4], a2[4], cmv[4], csv[4], hm[4], hi[4];
  f4 cS;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    adt[s] = dt[j * C1 + 2 * gq + s];    // A[token j][channel 2g+s]
    aw1[s] = W1[(2 * gq + s) * L + j];   // A[token j][channel 2g+s]
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    a1[v] = j < C1 ? dt[(4 * gq + v) * C1 + j] : 0.f;  // A[channel j][token 4g+v]
    a2[v] = j < C1 ? W1[j * L + 4 * gq + v] : 0.f;     // A[channel j][token 4g+v]
    cS[v] = b1[4 * gq + v];
    cmv[v] = ts[4 * gq + v];
    csv[v] = 1.0f / ts[L + 4 * gq + v];
    const int c = (4 * gq + v) & (C1 - 1);
    hm[v] = st[c];
    hi[v] = st[C1 + c];
  }
  // the constant operand of the 32-deep product: slots 0..3 = dT0^T (tokens 4g..4g+3), 4..7 = W1
  const float wop[8] = {a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3]};
  const Split8 wts = split8(wop);
  BnPair bn;
  bn.init(st, g, b, gq);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int ntiles = (P + 15) >> 4;
  for (int t = wave; t < ntiles; t += NW) {
    const int p = t * 16 + j;
    const bool ok = p < P, lo = ok && gq < 2;
    float x0 = 0.f, x1 = 0.f;
    f4 dx = {0.f, 0.f, 0.f, 0.f}, raw4 = dx;
    if (ok) {
      const float2 raw = *reinterpret_cast<const float2 *>(conv + ((size_t)n * P + p) * C1 + 2 * gq);
      x0 = bn.apply(raw.x, 0);
      x1 = bn.apply(raw.y, 1);
    }
    if (lo) {
      dx = *reinterpret_cast<const f4 *>(pdx + ((size_t)n * P + p) * C1 + 4 * gq);
      raw4 = *reinterpret_cast<const f4 *>(conv + ((size_t)n * P + p) * C1 + 4 * gq);
    }
    f4 sc = cS, dA = {0.f, 0.f, 0.f, 0.f};
    sc = mfma16(aw1[0], x0, sc);
    sc = mfma16(aw1[1], x1, sc);
    dA = mfma16(adt[0], x0, dA);
    dA = mfma16(adt[1], x1, dA);
    f4 A;
#pragma unroll
    for (int v = 0; v < 4; ++v) A[v] = ok ? __expf(sc[v] - cmv[v]) * csv[v] : 0.f;
    const float dot = gsum((dA[0] * A[0] + dA[1] * A[1]) + (dA[2] * A[2] + dA[3] * A[3]));
    f4 dS;
#pragma unroll
    for (int v = 0; v < 4; ++v) dS[v] = A[v] * (dA[v] - dot);
    // dx += dT0^T . A + W1 . dS: one 32-deep contraction (16 tokens of each product)
    const float bop[8] = {A[0], A[1], A[2], A[3], dS[0], dS[1], dS[2], dS[3]};
    dx = mfma32_split(wts, split8(bop), dx);
    if (lo) {
      *reinterpret_cast<f4 *>(dy + ((size_t)n * P + p) * C1 + 4 * gq) = dx;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        s1[v] += dx[v];
        s2[v] += dx[v] * ((raw4[v] - hm[v]) * hi[v]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    s1[v] = jsum(s1[v]);
    s2[v] = jsum(s2[v]);
  }
  if (j == 0 && gq < 2) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      red[wave][4 * gq + v] = s1[v];
      red[wave][C1 + 4 * gq + v] = s2[v];
    }
  }
  __syncthreads();
  if (tid < 2 * C1) {
    const float r = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    if constexpr (!TAIL) part[(size_t)n * 2 * C1 + tid] = r;
    else tail_store(part + (size_t)n * 2 * C1 + tid, r);
  }
  if constexpr (TAIL) bn_bwd_tail(tail, part, n, gridDim.x, C1);
}


// ---------------------------------------------------------------------------------------------
// BatchNorm2D backward apply + Conv2DLayer::backward gradients (reference
// layers/conv2d_layer.hpp:115-147): gb[co] = sum delta, gW[co][ci][kh][kw] = sum xpad * delta.
// The conv input delta (:150-190) is never consumed (first layer) and is not computed.
// A contraction over pixels: D[tap][co] += A[tap][pixel] . B[pixel][co], with the shifted image
// value as the A operand (lane = tap; lane 9 carries ones and yields the bias gradient) and the
// delta in N layout as the B operand; 4 MFMAs per 16 pixels and input channel.  The image is read
// straight from global memory (3 KB per channel and sample: it lives in L1).
// Per-image partials [C1*Ci*9 | C1].
// ---------------------------------------------------------------------------------------------
constexpr int CW_MAXCI = 4;
// W4: the row length is a multiple of 4, so the 4 pixels a lane owns in a tile (4g .. 4g+3) lie in
// one image row: one row/column split per tile instead of four, one base address per operand.
template <bool W4>
__global__ __launch_bounds__(TPB) void k_conv_wgrad(const float *__restrict__ X, const float *__restrict__ dy,
                                                    const float *__restrict__ coef, float *__restrict__ gslab, int Ci,
                                                    int H, int W, int ld, int sgroup) {
  __shared__ float sred[NW][CW_MAXCI][256];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, gq = lane >> 4;
  if (sgroup > 0) coef += (size_t)(n / sgroup) * 2 * C1;  // BatchNorm2D backward coefficients of this image's mini-batch
  const int P = H * W, ntiles = (P + 15) >> 4;
  const int kh = j / 3 - 1, kw = j % 3 - 1;  // tap offsets of lanes j < 9
  const float c0 = j < C1 ? coef[j] : 0.f, c1 = j < C1 ? coef[C1 + j] : 0.f;
  const float invW = 1.0f / (float)W;
  f4 acc[CW_MAXCI];
#pragma unroll
  for (int ci = 0; ci < CW_MAXCI; ++ci) acc[ci] = f4{0.f, 0.f, 0.f, 0.f};
  const float *__restrict__ dyb = dy + (size_t)n * P * C1 + (j < C1 ? j : 0);
  const float *__restrict__ xb = X + (size_t)n * Ci * P;
  // W4: the zero-padded image ((H+2) x (W+2) per channel) is staged in LDS once, so that a shifted
  // image value is one LDS read with no clamps or border masks (the launcher checks that it fits)
  extern __shared__ float xt[];
  const int TW = W + 2, plane = (H + 2) * TW;
  if (W4) {
    for (int i = tid; i < Ci * plane; i += TPB) {
      const int ci = i / plane, r = i - ci * plane, hh = r / TW - 1, ww = r - (hh + 1) * TW - 1;
      const bool in = hh >= 0 && hh < H && ww >= 0 && ww < W;
      const float v = xb[(size_t)ci * P + (in ? hh * W + ww : 0)];
      xt[i] = in ? v : 0.f;
    }
    __syncthreads();
  }
  for (int t = wave; t < ntiles; t += NW) {
    float dv[4];
    if (W4) {
      const int p4 = t * 16 + 4 * gq;  // P is a multiple of 4 too: the 4 pixels are valid together
      const bool valid = p4 < P;
      int h = (int)(((float)p4 + 0.5f) * invW);
      h -= (h * W > p4) ? 1 : 0;
      h += ((h + 1) * W <= p4) ? 1 : 0;
      const int w0 = p4 - h * W;
      const int pc = valid ? p4 : 0;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float d = dyb[(size_t)(pc + v) * C1];
        dv[v] = (valid && j < C1) ? d * c0 + c1 : 0.f;
      }
      // lanes j < 9: tap (kh, kw) of the padded tile; lane 9: the ones row (bias gradient); lanes past
      // it contribute rows that are never read.  Pixels past the image have a zero delta.
      const int toff = valid ? (h + kh + 1) * TW + w0 + kw + 1 : TW + 1;
#pragma unroll
      for (int ci = 0; ci < CW_MAXCI; ++ci) {
        if (ci < Ci) {
          const float *__restrict__ xr = xt + ci * plane + (j < 9 ? toff : TW + 1);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float xv = j == 9 ? 1.f : xr[v];
            acc[ci] = mfma16(xv, dv[v], acc[ci]);
          }
        }
      }
    } else {
      int hh[4], ww[4];
      bool inb[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int pr = t * 16 + 4 * gq + v;
        const bool valid = pr < P;
        int h = (int)(((float)pr + 0.5f) * invW);
        h -= (h * W > pr) ? 1 : 0;
        h += ((h + 1) * W <= pr) ? 1 : 0;
        const int w = pr - h * W;
        hh[v] = h + kh;
        ww[v] = w + kw;
        inb[v] = valid && j < 9 && hh[v] >= 0 && hh[v] < H && ww[v] >= 0 && ww[v] < W;
        dv[v] = (valid && j < C1) ? dy[((size_t)n * P + pr) * C1 + j] * c0 + c1 : 0.f;
      }
#pragma unroll
      for (int ci = 0; ci < CW_MAXCI; ++ci) {
        if (ci < Ci) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            float xv = inb[v] ? X[((size_t)(n * Ci + ci) * H + hh[v]) * W + ww[v]] : 0.f;
            if (j == 9) xv = 1.f;
            acc[ci] = mfma16(xv, dv[v], acc[ci]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int ci = 0; ci < CW_MAXCI; ++ci)
#pragma unroll
    for (int v = 0; v < 4; ++v) sred[wave][ci][(4 * gq + v) * 16 + j] = acc[ci][v];
  __syncthreads();
  float *row = gslab + (size_t)n * ld;
  for (int o = tid; o < Ci * 9 * C1 + C1; o += TPB) {
    int ci, tap, co;
    if (o < Ci * 9 * C1) {
      co = o / (Ci * 9);
      ci = (o / 9) % Ci;
      tap = o % 9;
    } else {
      co = o - Ci * 9 * C1;
      ci = 0;
      tap = 9;
    }
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += sred[w][ci][tap * 16 + co];
    row[o] = s;
  }
}


// ---------------------------------------------------------------------------------------------
// fc DenseLayer::backward weight / bias gradients (reference layers/dense_layer.hpp:92-96):
// gW[f][k] = sum_n pooled[n][f] * delta[n][k], gb[k] = sum_n delta[n][k].
// A contraction over samples: D[k][f] += A[k][n] . B[n][f] with the delta as the A operand (lane =
// class) and 16 consecutive features of the pooled rows as the B operand.  One wave owns a tile of
// 16 features and a chunk of FCW_CH samples; all of its 2 * FCW_CH/4 loads are issued before the
// first MFMA.  One extra tile whose B operand is all ones yields the bias gradient.  Chunks are
// summed afterwards (partial slab + column reduction).
// ---------------------------------------------------------------------------------------------
constexpr int FCW_CH = 128;
__global__ __launch_bounds__(TPB) void k_fc_wgrad(const float *__restrict__ pooled, const float *__restrict__ dl,
                                                  float *__restrict__ out, int N, int Q, int K, int ld) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, gq = lane >> 4;
  const int qt = (Q + 15) >> 4, tile = blockIdx.x * NW + wave;
  if (tile > qt) return;
  const bool bias_tile = tile == qt;
  const int n0 = blockIdx.y * FCW_CH, f = tile * 16 + j;
  const bool fok = !bias_tile && f < Q;
  float a[FCW_CH / 4], b[FCW_CH / 4];
#pragma unroll
  for (int s = 0; s < FCW_CH / 4; ++s) {
    const int n = n0 + 4 * s + gq, nn = min(n, N - 1);
    a[s] = j < K ? dl[(size_t)nn * K + j] : 0.f;
    b[s] = fok ? pooled[(size_t)nn * Q + f] : 1.f;
    if (n >= N) a[s] = 0.f;
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < FCW_CH / 4; ++s) acc = mfma16(a[s], b[s], acc);
  float *row = out + (size_t)blockIdx.y * ld;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int k = 4 * gq + v;
    if (k < K) {
      if (fok) row[(size_t)f * K + k] = acc[v];
      else if (bias_tile && j == 0) row[(size_t)Q * K + k] = acc[v];
    }
  }
}

}  // namespace vt

using namespace vt;

int vt::projector_fwd_tail(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                           const float *pk, const float *wd, const float *bd, const float *pivot, float *proj_a, float *z,
                           float *part_pn, int N, int P, const BnFwdTail &tail, hipStream_t s, int sgroup) {
  if (N < 1 || P < 1 || P > 16 * NW * TK_MAXT || !pivot || sgroup < 0) return VT_EUNSUPPORTED;
  if (part_pn && tail.ticket && sgroup == 0)
    hipLaunchKernelGGL(k_proj_fwd<true>, dim3(N), dim3(TPB), 0, s, conv_out, bn_stat, bn_gamma, bn_beta, pk, wd, bd, pivot, proj_a,
                       z, part_pn, P, 0, tail);
  else
    hipLaunchKernelGGL(k_proj_fwd<false>, dim3(N), dim3(TPB), 0, s, conv_out, bn_stat, bn_gamma, bn_beta, pk, wd, bd, pivot, proj_a,
                       z, part_pn, P, sgroup, BnFwdTail{});
  VT_CHECK_LAUNCH();
  return 0;
}
extern "C" int vt_projector_fwd(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                                const float *pk, const float *wd, const float *bd, const float *pivot, float *proj_a,
                                float *z, float *part_pn, int N, int P, vt_stream s) {
  return projector_fwd_tail(conv_out, bn_stat, bn_gamma, bn_beta, pk, wd, bd, pivot, proj_a, z, part_pn, N, P, BnFwdTail{},
                            (hipStream_t)s, 0);
}


// windows: du is the pair-form delta of k_head_bwd_win (P must be 784); sgroup: images per statistics group
// (pn_coef and bn_stat then hold one record per group), 0 = the whole mini-batch is one group
int vt::projector_bwd_grouped(const float *du, int windows, const float *pn_coef, const float *conv_out, const float *bn_stat,
                              const float *bn_gamma, const float *bn_beta, const float *pk, const float *wq, const float *wd,
                              float *pdx, float *dv, float *dk, float *gslab, int N, int P, int sgroup, hipStream_t s) {
  if (N < 1 || P < 1 || sgroup < 0 || (windows && P != 784)) return VT_EUNSUPPORTED;
  if (windows)
    hipLaunchKernelGGL(k_proj_bwd<true>, dim3(N), dim3(TPB), 0, s, du, pn_coef, conv_out, bn_stat, bn_gamma, bn_beta, pk, wq, wd,
                       pdx, dv, dk, gslab, P, sgroup);
  else
    hipLaunchKernelGGL(k_proj_bwd<false>, dim3(N), dim3(TPB), 0, s, du, pn_coef, conv_out, bn_stat, bn_gamma, bn_beta, pk, wq, wd,
                       pdx, dv, dk, gslab, P, sgroup);
  VT_CHECK_LAUNCH();
  return 0;
}
extern "C" int vt_projector_bwd(const float *du, const float *pn_coef, const float *conv_out, const float *bn_stat,
                                const float *bn_gamma, const float *bn_beta, const float *pk, const float *wq,
                                const float *wd, float *pdx, float *dv, float *dk, float *gslab, int N, int P,
                                vt_stream s) {
  return projector_bwd_grouped(du, 0, pn_coef, conv_out, bn_stat, bn_gamma, bn_beta, pk, wq, wd, pdx, dv, dk, gslab, N, P, 0,
                               (hipStream_t)s);
}

extern "C" int vt_projector_bwd_windows(const float *dpair, const float *pn_coef, const float *conv_out,
                                        const float *bn_stat, const float *bn_gamma, const float *bn_beta, const float *pk,
                                        const float *wq, const float *wd, float *pdx, float *dv, float *dk, float *gslab,
                                        int N, int P, vt_stream s) {
  return projector_bwd_grouped(dpair, 1, pn_coef, conv_out, bn_stat, bn_gamma, bn_beta, pk, wq, wd, pdx, dv, dk, gslab, N, P, 0,
                               (hipStream_t)s);
}


int vt::tokenizer_fwd_tail(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                           const float *w1, const float *b1, const float *w2, const float *b2, float *tok_A, float *tok_stat,
                           float *tokens, float *part_n1, int N, int P, const BnFwdTail &tail, hipStream_t s, int sgroup) {
  if (N < 1 || P < 1 || P > 16 * NW * TK_MAXT || !tok_stat || sgroup < 0) return VT_EUNSUPPORTED;
  const bool tl = part_n1 != nullptr && tail.ticket != nullptr && sgroup == 0;
#define VT_TOKFWD(MAP, TL)                                                                                                  \
  hipLaunchKernelGGL((k_tokenizer_fwd<MAP, TL>), dim3(N), dim3(TPB), 0, s, conv_out, bn_stat, bn_gamma, bn_beta, w1, b1, w2, \
                     b2, tok_A, tok_stat, tokens, part_n1, P, sgroup, TL ? tail : BnFwdTail{})
  if (tok_A != nullptr && tl) VT_TOKFWD(true, true);
  else if (tok_A != nullptr) VT_TOKFWD(true, false);
  else if (tl) VT_TOKFWD(false, true);
  else VT_TOKFWD(false, false);
#undef VT_TOKFWD
  VT_CHECK_LAUNCH();
  return 0;
}
extern "C" int vt_tokenizer_fwd(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                                const float *w1, const float *b1, const float *w2, const float *b2, float *tok_A,
                                float *tok_stat, float *tokens, float *part_n1, int N, int P, vt_stream s) {
  return tokenizer_fwd_tail(conv_out, bn_stat, bn_gamma, bn_beta, w1, b1, w2, b2, tok_A, tok_stat, tokens, part_n1, N, P,
                            BnFwdTail{}, (hipStream_t)s, 0);
}

int vt::tokenizer_bwd_tail(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                           const float *tok_stat, const float *dT0, const float *w1, const float *b1, const float *pdx,
                           float *dy, float *part, int N, int P, const BnBwdTail &tail, hipStream_t s, int sgroup) {
  if (N < 1 || P < 1 || !tok_stat || sgroup < 0) return VT_EUNSUPPORTED;
  if (tail.ticket && sgroup == 0)
    hipLaunchKernelGGL(k_tokenizer_bwd<true>, dim3(N), dim3(TPB), 0, s, conv_out, bn_stat, bn_gamma, bn_beta, tok_stat, dT0, w1,
                       b1, pdx, dy, part, P, 0, tail);
  else
    hipLaunchKernelGGL(k_tokenizer_bwd<false>, dim3(N), dim3(TPB), 0, s, conv_out, bn_stat, bn_gamma, bn_beta, tok_stat, dT0, w1,
                       b1, pdx, dy, part, P, sgroup, BnBwdTail{});
  VT_CHECK_LAUNCH();
  return 0;
}
extern "C" int vt_tokenizer_bwd(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                                const float *tok_stat, const float *dT0, const float *w1, const float *b1,
                                const float *pdx, float *dy, float *part, int N, int P, vt_stream s) {
  return tokenizer_bwd_tail(conv_out, bn_stat, bn_gamma, bn_beta, tok_stat, dT0, w1, b1, pdx, dy, part, N, P, BnBwdTail{},
                            (hipStream_t)s, 0);
}


// sgroup: images per statistics group (bn_coef then holds one record per group), 0 = one group
int vt::conv3x3_wgrad_grouped(const float *x, const float *dy, const float *bn_coef, float *gslab, int N, int Ci, int H, int W,
                              int sgroup, hipStream_t s) {
  if (N < 1 || Ci < 1 || Ci > CW_MAXCI || H < 1 || W < 1 || sgroup < 0) return VT_EUNSUPPORTED;
  const int ld = C1 * Ci * 9 + C1;
  const size_t lds = (size_t)Ci * (H + 2) * (W + 2) * sizeof(float);
  if (W % 4 == 0 && lds <= 32 * 1024)
    hipLaunchKernelGGL(k_conv_wgrad<true>, dim3(N), dim3(TPB), lds, s, x, dy, bn_coef, gslab, Ci, H, W, ld, sgroup);
  else
    hipLaunchKernelGGL(k_conv_wgrad<false>, dim3(N), dim3(TPB), 0, s, x, dy, bn_coef, gslab, Ci, H, W, ld, sgroup);
  VT_CHECK_LAUNCH();
  return 0;
}
extern "C" int vt_conv3x3_wgrad(const float *x, const float *dy, const float *bn_coef, float *gslab, int N, int Ci,
                                int H, int W, vt_stream s) {
  return conv3x3_wgrad_grouped(x, dy, bn_coef, gslab, N, Ci, H, W, 0, (hipStream_t)s);
}


// scratch: at least ceil(N/128) * (Q*K + K) floats (unused when N <= 128 and gb follows gw).
extern "C" int vt_fc_wgrad(const float *pooled, const float *d_logits, float *gw, float *gb, float *scratch, int N,
                           int Q, int K, vt_stream s) {
  if (N < 1 || Q < 1 || K < 1 || K > KMAX) return VT_EUNSUPPORTED;
  const int ns = (N + FCW_CH - 1) / FCW_CH, ld = Q * K + K;
  const bool direct = (ns == 1 && gb == gw + (size_t)Q * K);
  if (!direct && !scratch) return VT_EINVAL;
  float *dst = direct ? gw : scratch;
  const int tiles = (Q + 15) / 16 + 1;
  VT_LAUNCH_SIGNALLING(k_fc_wgrad, dim3((tiles + NW - 1) / NW, ns), dim3(TPB), 0, (hipStream_t)s, pooled, d_logits, dst, N, Q, K,
                       ld);
  VT_CHECK_LAUNCH();
  if (!direct) {
    int rc = reduce_rows(scratch, ns, Q * K, ld, gw, (hipStream_t)s);
    if (rc) return rc;
    rc = reduce_rows(scratch + (size_t)Q * K, ns, K, ld, gb, (hipStream_t)s);
    if (rc) return rc;
  }
  return 0;
}


// ---------------------------------------------------------------------------------------------
// vt_fc_wgrad for callers that cannot allocate (code under HIP graph capture, tight serving loops,
// harnesses that forbid hipMalloc): the per-chunk partial slab that vt_fc_wgrad takes as its
// `scratch` argument lives in a file-scope device array instead.  Every GPU the library runs on holds
// its own copy of the array at its own address until the process ends; the address is looked up once
// per device (first call on that device, with that device current) and only that pointer is ever
// handed to kernels.  k_fc_wgrad writes every slab entry the reduction reads (all Q*K + K columns of
// each of the ceil(N/128) rows), so the previous call's contents never have to be cleared.  Calls that
// overlap on one GPU (two streams, two threads) share the array and race.
// ---------------------------------------------------------------------------------------------
constexpr int FCW_STATIC_MAX_FLOATS = 16 * (3136 * KMAX + KMAX);  // 2048 images of the reference's Q = 3136, 16 classes: 3.06 MiB
static __device__ __align__(256) float fcw_static_scratch[FCW_STATIC_MAX_FLOATS];
constexpr int FCW_MAX_DEVICES = 64;
static float *g_fcw_scratch_ptr[FCW_MAX_DEVICES];  // per device: address of that device's copy, 0 = not looked up yet

extern "C" int vt_fc_wgrad_static(const float *pooled, const float *d_logits, float *gw, float *gb, int N, int Q, int K,
                                  vt_stream s) {
  if (!pooled || !d_logits || !gw || !gb) return VT_EINVAL;
  if (N < 1 || Q < 1 || K < 1 || K > KMAX) return VT_EUNSUPPORTED;
  const long ns = (N + FCW_CH - 1) / FCW_CH, ld = (long)Q * K + K;
  if (ns * ld > (long)FCW_STATIC_MAX_FLOATS) return VT_ELIMIT;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FCW_MAX_DEVICES) return VT_EINVAL;
  float *scratch = __atomic_load_n(&g_fcw_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!scratch) {  // set-up, once per device: two racing threads look up the same address
    void *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(fcw_static_scratch));
    if (e != hipSuccess) return (int)e;
    if (!ptr) return VT_EINVAL;
    scratch = static_cast<float *>(ptr);
    __atomic_store_n(&g_fcw_scratch_ptr[dev], scratch, __ATOMIC_RELEASE);
  }
  const int tiles = (Q + 15) / 16 + 1;
  hipLaunchKernelGGL(k_fc_wgrad, dim3((tiles + NW - 1) / NW, (int)ns), dim3(TPB), 0, (hipStream_t)s, pooled, d_logits,
                     scratch, N, Q, K, (int)ld);
  VT_CHECK_LAUNCH();
  // the same two column reductions as vt_fc_wgrad (kernel launches only)
  int rc = reduce_rows(scratch, ns, Q * K, (int)ld, gw, (hipStream_t)s);
  if (rc) return rc;
  return reduce_rows(scratch + (size_t)Q * K, ns, K, (int)ld, gb, (hipStream_t)s);
}
