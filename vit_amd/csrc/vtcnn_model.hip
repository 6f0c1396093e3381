// Model handle: owns the parameter arena and the activation workspace, and chains the per-layer
// launchers into VTCNN::forward / backward / update_weights (reference model/vit_cnn.hpp).
// Host-side bit-exact pieces live here too: weight initialisation, .bin checkpoints, IDX files,
// loader shuffling.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "vtcnn_common.h"
#include "vtcnn_tail.h"
#include "vtcnn_generic.h"

namespace vt {
thread_local hipEvent_t g_stop_event = nullptr;
}
using namespace vt;

#define HIPCHK(x)                          \
  do {                                     \
    hipError_t e__ = (x);                  \
    if (e__ != hipSuccess) return (int)e__; \
  } while (0)
#define VTCHK(x)            \
  do {                      \
    int r__ = (x);          \
    if (r__ != 0) return r__; \
  } while (0)

struct vt_model {
  vt_dims d;
  vt_layout lo;
  int maxN = 0, device = 0, lastN = 0, training = 1;
  int P = 0, Q = 0, K = 0;
  hipStream_t gstream = nullptr;  // capture stream of the step graph
  // Backward runs two independent launch chains after the projector (the transformer/attention rows
  // and the tokenizer/BatchNorm2D/conv pixels); the second one is issued on this stream between a
  // fork event and a join event so that latency-bound row kernels overlap HBM-bound pixel kernels.
  hipStream_t side = nullptr;
  hipEvent_t ev_rows = nullptr, ev_join = nullptr;
  bool concurrent = true, side_open = false;
  bool gen_attention_done = false;  // general shapes: the last backward_main also ran the attention block
  bool join_signalled = false;  // the side chain's last kernel signals ev_join itself (no record needed)
  bool signal_launches = [] { const char *e = getenv("VT_SIGNAL_LAUNCHES"); return !(e && *e == '0'); }();  // VT_SIGNAL_LAUNCHES=0: event-record packets
  // Deferred attention backward (vt_model_set_deferred_attention): the attention block's backward and the
  // update of its weights run on attn_stream beside the NEXT step's forward, which never reads those
  // weights.  tokbuf[2]: the token matrix is double-buffered so that the next forward can write its tokens
  // while the attention backward still reads this step's.
  hipStream_t attn_stream = nullptr;
  hipEvent_t ev_attn_go = nullptr, ev_attn_done = nullptr;
  bool defer_attn = false, attn_inflight = false, attn_update_pending = false, capturing = false;
  float *tokbuf[2] = {nullptr, nullptr};
  int tok_parity = 0, attn_tok_parity = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  bool keep_internals = false;  // also store the two attention maps (test aid)
  bool own_arena = false, own_grads = false;
  float *arena = nullptr, *grads = nullptr;
  std::vector<void *> allocs;
  std::map<std::string, std::pair<float *, long>> named;  // name -> (device ptr, floats per sample or total)
  // forward workspace
  float *conv, *tokA = nullptr, *tokstat, *tok, *a, *hdn, *s2, *o, *pk, *projA = nullptr, *z, *pooled, *logits, *probs;
  unsigned char *pool_idx;
  float *part, *part_side = nullptr, *stat_bn, *stat_n1, *stat_n2, *stat_pn, *batch_mv;
  bool last_grouped = false;  // the last forward was a grouped evaluation sweep: no backward for it
  int last_group = 0;         // images per statistics group of the last (training-mode) forward; 0 = one mini-batch
  float *gpart_pn = nullptr, *gpart_n2 = nullptr, *gpart_n1 = nullptr, *gpart_bn = nullptr;  // per-group BatchNorm gradients
  bool attention_pending = false;  // vt_model_backward_main ran, vt_model_backward_attention has not yet
  const float *last_images = nullptr;
  float *last_probs = nullptr;  // where the last forward left the probabilities (the caller's buffer or m->probs)
  // backward workspace
  unsigned *tickets = nullptr;  // eight ticket sets of the BatchNorm closing steps that run inside their producer kernels
  bool tails = false;           // those closing steps in place of the separate finalize launches (opt-in: measured no faster)
  float *dpair = nullptr;   // delta per element pair [N, P*C2/2]: what the head backward hands the projector on 28x28 inputs
  bool du_windows = false;  // the last backward left dpair, not the dense du
  float *dlogits, *du, *pdx, *pdt, *dv, *dk, *d0, *ds2, *dh, *da, *dr, *dc, *dT0, *dy, *qkv, *gslab, *gslab_conv;
  float *sc_fc, *sc_pt, *sc_ffn, *sc_mha;  // partial-gradient slabs, one per launcher: they stay live until the fused reduction
  float *coef_pn, *coef_n2, *coef_n1, *coef_bn;
  float *loss_buf;
  // optional per-launcher timing with HIP events on the caller's stream
  std::string prof_filter;  // "" = off, "*" = every launcher, else one launcher name
  int prof_period = 1, prof_seen = 0;  // time every prof_period-th matching launch only
  std::vector<hipEvent_t> prof_ev;
  std::vector<int> prof_name;  // name index per event pair
  std::vector<std::string> prof_names;
  size_t prof_used = 0;

  float *adam_m1 = nullptr, *adam_m2 = nullptr;  // Adam moments over the trainable region, allocated on first use
  int adam_step = 0;
  // shapes outside the compiled headline kernels run the general, GEMM-based step instead
  GenModel *gen = nullptr;

  float *P_(const char *name) { return arena + lo.t[vt_find(&lo, name)].offset; }
  float *G_(const char *name) { return grads + lo.t[vt_find(&lo, name)].offset; }
};

extern "C" const char *vt_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case VT_EINVAL: return "invalid argument";
    case VT_EUNSUPPORTED: return "configuration not supported by the compiled gfx950 kernels";
    case VT_ENOMEM: return "out of memory";
    case VT_EIO: return "file i/o error";
    case VT_EFORMAT: return "checkpoint / file format mismatch";
    case VT_ELIMIT: return "size exceeds the fixed scratch capacity of the allocation-free entry point";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}
extern "C" int vt_version(void) { return 1; }

// Makes the model's device current for the lifetime of the object and restores the caller's afterwards.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) ok = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    int now = -1;
    if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev);
  }
};

// Host copies of the arenas wait for every stream of the device first: the step may have been issued on
// non-blocking streams (the caller's, the model's side streams, a deferred attention update), which a
// blocking copy on the null stream does not order against.
static int quiesce(vt_model *m) {
  HIPCHK(hipDeviceSynchronize());
  m->attn_inflight = false;
  m->attn_update_pending = false;
  return 0;
}

static int dalloc(vt_model *m, float **p, size_t floats, const char *name = nullptr) {
  void *q = nullptr;
  if (hipMalloc(&q, std::max<size_t>(floats, 4) * sizeof(float)) != hipSuccess) return VT_ENOMEM;
  m->allocs.push_back(q);
  *p = (float *)q;
  if (name) m->named[name] = {*p, (long)floats};
  return 0;
}

extern "C" int vt_model_create(const vt_dims *dims, int max_batch, int device, float *ext_arena, float *ext_grads,
                               vt_model **out) {
  if (!dims || !out || max_batch < 1) return VT_EINVAL;
  vt_layout lo;
  if (vt_make_layout(dims, &lo) != 0) return VT_EINVAL;
  // VT_GENERAL_PATH=1 (read at every create): the general-shape launchers also for the shape the tuned kernels are built
  // for (measurement aid: the same step on another set of kernels, and with vt_set_exact_products on exact fp32
  // matrix instructions throughout)
  const char *gp = getenv("VT_GENERAL_PATH");
  const bool headline = dims->vt_channels == C2 && dims->tokens == L && dims->token_channels == D && dims->heads == HEADS &&
                        dims->input_dim * dims->input_dim <= PPT * TPB && dims->num_classes <= KMAX &&
                        dims->image_channels <= 4 && !(gp && *gp && *gp != '0');
  DeviceGuard dg(device);
  if (!dg.ok) return VT_EINVAL;
  vt_model *m = new vt_model();
  m->d = *dims;
  m->lo = lo;
  m->maxN = max_batch;
  m->device = device;
  m->P = dims->input_dim * dims->input_dim;
  m->Q = dims->vt_channels * 196;
  m->K = dims->num_classes;
  const size_t N = max_batch, P = m->P, NL = N * L, Q = m->Q, K = m->K;
  int rc = 0;
  if (ext_arena) m->arena = ext_arena; else { rc |= dalloc(m, &m->arena, lo.n_total); m->own_arena = true; }
  if (ext_grads) m->grads = ext_grads; else { rc |= dalloc(m, &m->grads, lo.n_trainable); m->own_grads = true; }
  if (!headline) {
    m->gen = new GenModel();
    auto alloc = [](void *ctx, float **p, size_t floats, const char *name) { return dalloc((vt_model *)ctx, p, floats, name); };
    const int grc = rc ? VT_ENOMEM : gen_create(m->gen, dims, max_batch, alloc, m);
    if (grc != 0) {
      vt_model_destroy(m);
      return grc;
    }
    m->probs = m->gen->probs;
    rc |= dalloc(m, &m->loss_buf, 4);
    // the side chain of the general backward (gen_backward): stream and fork / join events
    if (rc == 0 && (hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&m->ev_rows, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming) != hipSuccess))
      rc = VT_ENOMEM;
    if (rc != 0) {
      vt_model_destroy(m);
      return VT_ENOMEM;
    }
    (void)hipMemset(m->arena, 0, lo.n_total * sizeof(float));
    (void)hipMemset(m->grads, 0, lo.n_trainable * sizeof(float));
    *out = m;
    return 0;
  }
  rc |= dalloc(m, &m->conv, N * P * C1, "fwd.conv_out");
  rc |= dalloc(m, &m->tokstat, N * 2 * L, "fwd.tok_stat");
  rc |= dalloc(m, &m->tok, NL * D, "fwd.tok_out");
  m->tokbuf[0] = m->tok;
  rc |= dalloc(m, &m->a, NL * D, "fwd.tr_a");
  rc |= dalloc(m, &m->hdn, NL * D, "fwd.tr_hdn");
  rc |= dalloc(m, &m->s2, NL * D, "fwd.tr_s2");
  rc |= dalloc(m, &m->o, NL * D, "fwd.tr_out");
  rc |= dalloc(m, &m->pk, N * PK_STRIDE, "fwd.proj_pk");
  rc |= dalloc(m, &m->z, N * P * C2, "fwd.proj_z");
  rc |= dalloc(m, &m->pooled, N * Q, "fwd.pooled");
  rc |= dalloc(m, &m->logits, N * K, "fwd.logits");
  rc |= dalloc(m, &m->probs, N * K, "fwd.preds");
  float *pi = nullptr;
  rc |= dalloc(m, &pi, (N * Q + 3) / 4, "fwd.pool_idx");  // raw bytes, 4 slots per float word
  m->pool_idx = (unsigned char *)pi;
  const size_t nrb = (NL + RB - 1) / RB;
  // partial-sum records: one per image or row block, or the head backward's window-major count, which is not
  // monotonic in the batch size (at most 13 * 128 workgroups for any batch up to N)
  // (with statistics groups a chunk of the head backward divides the group: at most N/4 + 1 chunks of 13 records)
  rc |= dalloc(m, &m->part, std::max<size_t>(std::max<size_t>(std::max(N, nrb) * 64, std::min<size_t>(13 * N, 13 * 128) * 2 * C2),
                                              (size_t)13 * (N / 4 + 2) * 2 * C2));
  rc |= dalloc(m, &m->part_side, std::max(N, nrb) * 64);
  // one (mean, inv) record per statistics group: grouped forward runs up to maxN/4 groups
  const size_t max_groups = N / 4 + 1;
  rc |= dalloc(m, &m->stat_bn, max_groups * 2 * C1, "stat.bn");
  rc |= dalloc(m, &m->stat_n1, max_groups * 2 * D, "stat.norm1");
  rc |= dalloc(m, &m->stat_n2, max_groups * 2 * D, "stat.norm2");
  rc |= dalloc(m, &m->stat_pn, max_groups * 2 * C2, "stat.proj.norm");
  rc |= dalloc(m, &m->batch_mv, max_groups * 2 * D);
  rc |= dalloc(m, &m->dlogits, N * K, "bwd.d_logits");
  rc |= dalloc(m, &m->du, N * P * C2, "bwd.du");
  rc |= dalloc(m, &m->dpair, N * P * C2 / 2, "bwd.dpair");
  {
    float *tk = nullptr;
    rc |= dalloc(m, &tk, 8 * TAIL_WORDS);
    m->tickets = (unsigned *)tk;
    if (rc == 0 && hipMemset(tk, 0, 8 * TAIL_WORDS * sizeof(unsigned)) != hipSuccess) rc = VT_ENOMEM;
  }
  rc |= dalloc(m, &m->pdx, N * P * C1, "bwd.proj_dx");
  rc |= dalloc(m, &m->pdt, NL * D, "bwd.proj_dt");
  rc |= dalloc(m, &m->dv, NL * C2, "bwd.proj_dv");
  rc |= dalloc(m, &m->dk, NL * D, "bwd.proj_dk");
  rc |= dalloc(m, &m->d0, NL * D, "bwd.tr_d0");
  rc |= dalloc(m, &m->ds2, NL * D, "bwd.tr_ds2");
  rc |= dalloc(m, &m->dh, NL * D, "bwd.tr_dh");
  rc |= dalloc(m, &m->da, NL * D, "bwd.tr_da");
  rc |= dalloc(m, &m->dr, NL * D, "bwd.tr_dr");
  rc |= dalloc(m, &m->dc, NL * D, "bwd.tr_dc");
  rc |= dalloc(m, &m->dT0, NL * C1, "bwd.tok_dT0");
  rc |= dalloc(m, &m->dy, N * P * C1, "bwd.tok_dx");
  rc |= dalloc(m, &m->qkv, 3 * NL * D, "bwd.mha_qkv");
  rc |= dalloc(m, &m->gslab, N * 400, "bwd.gslab");
  rc |= dalloc(m, &m->gslab_conv, N * (C1 * 9 * dims->image_channels + C1));
  // partial slabs of the row-matrix launchers (sizes: see the scratch notes in vtcnn_hip.h)
  rc |= dalloc(m, &m->sc_ffn, nrb * 2112);
  rc |= dalloc(m, &m->sc_pt, nrb * 1552);
  rc |= dalloc(m, &m->sc_fc, ((N + 127) / 128) * (Q * K + K));
  // (the attention groups of a grouped step: L per statistics group, one [D] pivot each)
  rc |= dalloc(m, &m->sc_mha, (size_t)HEADS * 4 * (2 * D + 3) * NL + (size_t)HEADS * nrb * 1536 + (size_t)L * D * max_groups);
  rc |= dalloc(m, &m->coef_pn, max_groups * 2 * C2, "coef.proj.norm");
  rc |= dalloc(m, &m->coef_n2, max_groups * 2 * D, "coef.norm2");
  rc |= dalloc(m, &m->coef_n1, max_groups * 2 * D, "coef.norm1");
  rc |= dalloc(m, &m->coef_bn, max_groups * 2 * C1, "coef.bn");
  rc |= dalloc(m, &m->gpart_pn, max_groups * 2 * C2);
  rc |= dalloc(m, &m->gpart_n2, max_groups * 2 * D);
  rc |= dalloc(m, &m->gpart_n1, max_groups * 2 * D);
  rc |= dalloc(m, &m->gpart_bn, max_groups * 2 * C1);
  rc |= dalloc(m, &m->loss_buf, 4);
  if (rc == 0 && (hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess ||
                  hipEventCreateWithFlags(&m->ev_rows, hipEventDisableTiming) != hipSuccess ||
                  hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming) != hipSuccess))
    rc = VT_ENOMEM;
  if (rc != 0) {
    vt_model_destroy(m);
    return VT_ENOMEM;
  }
  (void)hipMemset(m->arena, 0, lo.n_total * sizeof(float));
  (void)hipMemset(m->grads, 0, lo.n_trainable * sizeof(float));
  *out = m;
  return 0;
}

extern "C" int vt_model_destroy(vt_model *m) {
  if (!m) return 0;
  DeviceGuard dg(m->device);
  (void)hipDeviceSynchronize();
  if (m->graph_exec) (void)hipGraphExecDestroy(m->graph_exec);
  if (m->graph) (void)hipGraphDestroy(m->graph);
  if (m->gstream) (void)hipStreamDestroy(m->gstream);
  if (m->side) (void)hipStreamDestroy(m->side);
  if (m->attn_stream) (void)hipStreamDestroy(m->attn_stream);
  for (hipEvent_t e : {m->ev_rows, m->ev_join, m->ev_attn_go, m->ev_attn_done})
    if (e) (void)hipEventDestroy(e);
  for (void *p : m->allocs) (void)hipFree(p);
  for (hipEvent_t e : m->prof_ev) (void)hipEventDestroy(e);
  delete m->gen;
  delete m;
  return 0;
}

extern "C" int vt_model_is_fused(const vt_model *m) { return !m ? VT_EINVAL : (m->gen ? 0 : 1); }
extern "C" int vt_model_layout(const vt_model *m, vt_layout *out) {
  if (!m || !out) return VT_EINVAL;
  *out = m->lo;
  return 0;
}

extern "C" float *vt_model_arena_ptr(vt_model *m) { return m ? m->arena : nullptr; }
extern "C" float *vt_model_grads_ptr(vt_model *m) { return m ? m->grads : nullptr; }

extern "C" int vt_model_set_arena(vt_model *m, const float *host) {
  if (!m || !host) return VT_EINVAL;
  DeviceGuard dg(m->device);
  if (!dg.ok) return VT_EINVAL;
  VTCHK(quiesce(m));
  HIPCHK(hipMemcpy(m->arena, host, m->lo.n_total * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}
extern "C" int vt_model_get_arena(vt_model *m, float *host) {
  if (!m || !host) return VT_EINVAL;
  DeviceGuard dg(m->device);
  if (!dg.ok) return VT_EINVAL;
  VTCHK(quiesce(m));
  HIPCHK(hipMemcpy(host, m->arena, m->lo.n_total * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int vt_model_get_grads(vt_model *m, float *host) {
  if (!m || !host) return VT_EINVAL;
  DeviceGuard dg(m->device);
  if (!dg.ok) return VT_EINVAL;
  VTCHK(quiesce(m));
  HIPCHK(hipMemcpy(host, m->grads, m->lo.n_trainable * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Weight initialisation: same generator, same distributions, same draw order as the reference.
//   conv: Tensor::kaiming_normal, sigma = sqrt(2/fan_in)       (conv2d_layer.hpp:27-29, tensor_cpu.hpp:98-106)
//   Dense: Tensor::rand_uniform(+-sqrt(6/(in+out)))            (dense_layer.hpp:22-23, tensor_cpu.hpp:69-76)
// Construction order: conv; tokenizer linear1, linear2 (tokenizer.hpp:18-19); per head q,k,v then
// the output projection (multi_head_attention.hpp:33-46); ffn linear1, linear2
// (transformer_layer.hpp:21-23); projector linear1..3, downsample (projector_layer.hpp:23-34);
// fc (vit_cnn.hpp:91).  BatchNorm: gamma 1, beta 0, running mean 0, running var 1.
// Uses libstdc++'s std::normal_distribution / std::uniform_real_distribution<float>, the same
// library code the reference is built against.
// ---------------------------------------------------------------------------------------------
static void init_arena_host(const vt_layout &lo, const vt_dims &d, unsigned seed, std::vector<float> &h) {
  h.assign(lo.n_total, 0.f);
  std::mt19937 gen(seed);
  auto T = [&](const std::string &n) { return h.data() + lo.t[vt_find(&lo, n.c_str())].offset; };
  auto SZ = [&](const std::string &n) { return lo.t[vt_find(&lo, n.c_str())].size; };
  auto dense = [&](const std::string &n, int in, int out) {
    const float limit = std::sqrt(6.0f / (in + out));
    std::uniform_real_distribution<float> dist(-limit, limit);
    float *p = T(n);
    for (int i = 0; i < in * out; ++i) p[i] = dist(gen);
  };
  auto fill = [&](const std::string &n, float v) { std::fill(T(n), T(n) + SZ(n), v); };
  const int Ci = d.image_channels, c1 = d.vt_channels / 2, c2 = d.vt_channels, l = d.tokens, dd = d.token_channels;
  const int hd = dd / d.heads;
  {
    const int fan_in = Ci * 3 * 3;
    const float scale = std::sqrt(2.0f / fan_in);
    std::normal_distribution<float> dist(0.0f, scale);
    float *p = T("conv.W");
    for (int i = 0; i < c1 * Ci * 9; ++i) p[i] = dist(gen);
  }
  dense("tok.linear1.W", c1, l);
  dense("tok.linear2.W", c1, dd);
  for (int i = 0; i < d.heads; ++i) {
    const std::string s = std::to_string(i);
    dense("mha.q" + s + ".W", dd, hd);
    dense("mha.k" + s + ".W", dd, hd);
    dense("mha.v" + s + ".W", dd, hd);
  }
  dense("mha.o.W", dd, dd);
  dense("ffn.linear1.W", dd, dd);
  dense("ffn.linear2.W", dd, dd);
  dense("proj.linear1.W", c1, dd);
  dense("proj.linear2.W", dd, dd);
  dense("proj.linear3.W", dd, c2);
  dense("proj.downsample.W", c1, c2);
  dense("fc.W", c2 * 14 * 14, d.num_classes);
  for (const char *n : {"bn.gamma", "norm1.gamma", "norm2.gamma", "proj.norm.gamma", "bn.running_var",
                        "norm1.running_var", "norm2.running_var", "proj.norm.running_var"})
    fill(n, 1.0f);
}

extern "C" int vt_model_init_seed(vt_model *m, unsigned seed) {
  if (!m) return VT_EINVAL;
  std::vector<float> h;
  init_arena_host(m->lo, m->d, seed, h);
  return vt_model_set_arena(m, h.data());
}

// ---------------------------------------------------------------------------------------------
// .bin checkpoint v1 (reference model/vit_cnn.hpp:276-546, batch_normalization_layer.hpp:384-505,
// visual_transformer.hpp:244-270): little-endian;
//   i32 4
//   i32 5 "Layer"             W[C1,Ci,3,3] b[C1]          (Conv2DLayer has no get_type override)
//   i32 11 "BatchNorm2D"      gamma beta running_mean running_var
//   i32 17 "VisualTransformer" i32 12, tokenizer (4 tensors) + projector (8 tensors)
//   i32 5 "Dense"             fc.W fc.b
// tensor record = i32 ndim, i32 dims[ndim], i32 count, f32 data[count].
// The transformer block and the BatchNorm1D running statistics are not part of the file.
// ---------------------------------------------------------------------------------------------
namespace {
struct CkEntry {
  const char *type;
  int count_field;  // write the "number of tensors" field (VisualTransformer only)
  std::vector<const char *> tensors;
};
const std::vector<CkEntry> &ck_entries() {
  static const std::vector<CkEntry> e = {
      {"Layer", 0, {"conv.W", "conv.b"}},
      {"BatchNorm2D", 0, {"bn.gamma", "bn.beta", "bn.running_mean", "bn.running_var"}},
      {"VisualTransformer", 1,
       {"tok.linear1.W", "tok.linear1.b", "tok.linear2.W", "tok.linear2.b", "proj.linear1.W", "proj.linear2.W",
        "proj.linear3.W", "proj.linear3.b", "proj.norm.gamma", "proj.norm.beta", "proj.downsample.W",
        "proj.downsample.b"}},
      {"Dense", 0, {"fc.W", "fc.b"}},
  };
  return e;
}
}  // namespace

extern "C" int vt_host_init_arena(const vt_dims *dims, unsigned seed, float *arena_host) {
  vt_layout lo;
  if (!dims || !arena_host || vt_make_layout(dims, &lo) != 0) return VT_EINVAL;
  std::vector<float> h;
  init_arena_host(lo, *dims, seed, h);
  std::memcpy(arena_host, h.data(), sizeof(float) * lo.n_total);
  return 0;
}

static const char kV2Magic[8] = {'V', 'T', 'C', 'N', 'N', 'v', '2', 0};

static bool in_v1(const char *name) {
  for (const CkEntry &e : ck_entries())
    for (const char *tn : e.tensors)
      if (std::strcmp(tn, name) == 0) return true;
  return false;
}

static void write_tensor_record(FILE *f, const vt_tensor_info &ti, const float *arena_host) {
  auto wi = [&](int v) { std::fwrite(&v, 4, 1, f); };
  wi(ti.ndim);
  for (int k = 0; k < ti.ndim; ++k) wi(ti.shape[k]);
  wi(ti.size);
  std::fwrite(arena_host + ti.offset, 4, ti.size, f);
}

static int checkpoint_write(const vt_dims *dims, const float *arena_host, const vt_train_state *st, bool v2,
                            const char *path) {
  vt_layout lo;
  if (!dims || !arena_host || !path || vt_make_layout(dims, &lo) != 0) return VT_EINVAL;
  FILE *f = std::fopen(path, "wb");
  if (!f) return VT_EIO;  // "No se pudo abrir el archivo para guardar el modelo" (vit_cnn.hpp:280)
  auto wi = [&](int v) { std::fwrite(&v, 4, 1, f); };
  wi(4);
  for (const CkEntry &e : ck_entries()) {
    const int len = (int)std::strlen(e.type);
    wi(len);
    std::fwrite(e.type, 1, len, f);
    if (e.count_field) wi((int)e.tensors.size());
    for (const char *tn : e.tensors) write_tensor_record(f, lo.t[vt_find(&lo, tn)], arena_host);
  }
  if (v2) {
    std::fwrite(kV2Magic, 1, 8, f);
    wi(2);
    int extra = 0;
    for (int k = 0; k < lo.n_tensors; ++k) extra += in_v1(lo.t[k].name) ? 0 : 1;
    wi(extra);
    for (int k = 0; k < lo.n_tensors; ++k) {
      if (in_v1(lo.t[k].name)) continue;
      const int len = (int)std::strlen(lo.t[k].name);
      wi(len);
      std::fwrite(lo.t[k].name, 1, len, f);
      write_tensor_record(f, lo.t[k], arena_host);
    }
    const vt_train_state zero = {2, 0, 0, 42u, 0};
    const vt_train_state &t = st ? *st : zero;
    std::fwrite(&t.step, 8, 1, f);
    wi(t.epoch);
    std::fwrite(&t.loader_seed, 4, 1, f);
    wi(t.loader_resets);
  }
  const bool ok = std::ferror(f) == 0;
  std::fclose(f);
  return ok ? 0 : VT_EIO;
}

extern "C" int vt_checkpoint_write(const vt_dims *dims, const float *arena_host, const char *path) {
  return checkpoint_write(dims, arena_host, nullptr, false, path);
}
extern "C" int vt_checkpoint_write_v2(const vt_dims *dims, const float *arena_host, const vt_train_state *state,
                                      const char *path) {
  return checkpoint_write(dims, arena_host, state, true, path);
}

extern "C" int vt_model_save(vt_model *m, const char *path) {
  if (!m || !path) return VT_EINVAL;
  std::vector<float> h(m->lo.n_total);
  VTCHK(vt_model_get_arena(m, h.data()));
  return vt_checkpoint_write(&m->d, h.data(), path);
}

static int checkpoint_read(const vt_layout &lo, std::vector<float> &h, const char *path, vt_train_state *st = nullptr);

extern "C" int vt_model_save_v2(vt_model *m, const vt_train_state *state, const char *path) {
  if (!m || !path) return VT_EINVAL;
  std::vector<float> h(m->lo.n_total);
  VTCHK(vt_model_get_arena(m, h.data()));
  return vt_checkpoint_write_v2(&m->d, h.data(), state, path);
}

// Tensors absent from the file (transformer block, BatchNorm1D running statistics) keep the
// values already in arena_host, exactly as a freshly constructed reference model keeps its
// seed-42 initialisation for them after load().
extern "C" int vt_checkpoint_read(const vt_dims *dims, float *arena_host, const char *path) {
  vt_layout lo;
  if (!dims || !arena_host || !path || vt_make_layout(dims, &lo) != 0) return VT_EINVAL;
  std::vector<float> h(arena_host, arena_host + lo.n_total);
  const int rc = checkpoint_read(lo, h, path);
  if (rc == 0) std::memcpy(arena_host, h.data(), sizeof(float) * lo.n_total);
  return rc;
}

extern "C" int vt_model_load(vt_model *m, const char *path) {
  if (!m || !path) return VT_EINVAL;
  std::vector<float> h(m->lo.n_total);
  VTCHK(vt_model_get_arena(m, h.data()));
  VTCHK(checkpoint_read(m->lo, h, path));
  return vt_model_set_arena(m, h.data());
}

extern "C" int vt_checkpoint_read_v2(const vt_dims *dims, float *arena_host, vt_train_state *state, const char *path) {
  vt_layout lo;
  if (!dims || !arena_host || !path || vt_make_layout(dims, &lo) != 0) return VT_EINVAL;
  std::vector<float> h(arena_host, arena_host + lo.n_total);
  vt_train_state st = {1, 0, 0, 42u, 0};
  const int rc = checkpoint_read(lo, h, path, &st);
  if (rc == 0) {
    std::memcpy(arena_host, h.data(), sizeof(float) * lo.n_total);
    if (state) *state = st;
  }
  return rc;
}

extern "C" int vt_model_load_v2(vt_model *m, vt_train_state *state, const char *path) {
  if (!m || !path) return VT_EINVAL;
  std::vector<float> h(m->lo.n_total);
  VTCHK(vt_model_get_arena(m, h.data()));
  vt_train_state st = {1, 0, 0, 42u, 0};
  VTCHK(checkpoint_read(m->lo, h, path, &st));
  if (state) *state = st;
  return vt_model_set_arena(m, h.data());
}

// Reads one v1 tensor record into the arena slot of ti; shape and count must match.
static int read_tensor_record(FILE *f, const vt_tensor_info &ti, float *dst) {
  auto ri = [&](int *v) { return std::fread(v, 4, 1, f) == 1; };
  int nd = 0, cnt = 0, dim = 0;
  if (!ri(&nd) || nd != ti.ndim) return VT_EFORMAT;
  for (int k = 0; k < nd; ++k)
    if (!ri(&dim) || dim != ti.shape[k]) return VT_EFORMAT;  // shape mismatch (vit_cnn.hpp:444)
  if (!ri(&cnt) || cnt != ti.size || std::fread(dst, 4, cnt, f) != (size_t)cnt) return VT_EFORMAT;
  return 0;
}

static int checkpoint_read(const vt_layout &lo, std::vector<float> &h, const char *path, vt_train_state *st) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return VT_EIO;  // "No se pudo abrir el archivo para cargar el modelo" (vit_cnn.hpp:400)
  int rc = 0;
  auto ri = [&](int *v) { return std::fread(v, 4, 1, f) == 1; };
  int v = 0;
  if (rc == 0 && (!ri(&v) || v != 4)) rc = VT_EFORMAT;  // "Cantidad de capas no coincide" (vit_cnn.hpp:405)
  for (const CkEntry &e : ck_entries()) {
    if (rc != 0) break;
    int len = 0;
    char buf[64] = {0};
    if (!ri(&len) || len < 0 || len > 63 || std::fread(buf, 1, len, f) != (size_t)len || std::strcmp(buf, e.type) != 0) {
      rc = VT_EFORMAT;
      break;
    }
    if (e.count_field && (!ri(&v) || v != (int)e.tensors.size())) {
      rc = VT_EFORMAT;
      break;
    }
    for (const char *tn : e.tensors) {
      const vt_tensor_info &ti = lo.t[vt_find(&lo, tn)];
      rc = read_tensor_record(f, ti, h.data() + ti.offset);
      if (rc != 0) break;
    }
  }
  // optional v2 trailer; only looked for when the caller asked for the training state
  if (rc == 0 && st) {
    char magic[8];
    if (std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, kV2Magic, 8) == 0) {
      int ver = 0, extra = 0;
      if (!ri(&ver) || ver != 2 || !ri(&extra) || extra < 0 || extra > VT_MAX_TENSORS) rc = VT_EFORMAT;
      for (int k = 0; k < extra && rc == 0; ++k) {
        int len = 0;
        char name[VT_NAME_LEN] = {0};
        if (!ri(&len) || len < 1 || len >= VT_NAME_LEN || std::fread(name, 1, len, f) != (size_t)len) { rc = VT_EFORMAT; break; }
        const int idx = vt_find(&lo, name);
        if (idx < 0 || in_v1(name)) { rc = VT_EFORMAT; break; }
        rc = read_tensor_record(f, lo.t[idx], h.data() + lo.t[idx].offset);
      }
      if (rc == 0) {
        st->version = 2;
        if (std::fread(&st->step, 8, 1, f) != 1 || !ri(&st->epoch) || std::fread(&st->loader_seed, 4, 1, f) != 1 ||
            !ri(&st->loader_resets))
          rc = VT_EFORMAT;
      }
    } else {
      st->version = 1;
    }
  }
  std::fclose(f);
  return rc;
}

// ---------------------------------------------------------------------------------------------
// IDX reader and loader shuffling (reference utils/load_dataset.hpp).
// ---------------------------------------------------------------------------------------------
static bool read_be32(FILE *f, int *v) {
  unsigned char b[4];
  if (std::fread(b, 1, 4, f) != 4) return false;
  *v = (int)((unsigned)b[0] << 24 | (unsigned)b[1] << 16 | (unsigned)b[2] << 8 | (unsigned)b[3]);
  return true;
}

extern "C" int vt_idx_read_images(const char *path, int max_images, float *out, int *n, int *rows, int *cols) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return VT_EIO;  // "No se pudo abrir el archivo de imágenes" (load_dataset.hpp:24)
  int magic, num, r, c;
  if (!read_be32(f, &magic) || !read_be32(f, &num) || !read_be32(f, &r) || !read_be32(f, &c)) {
    std::fclose(f);
    return VT_EFORMAT;
  }
  if (max_images > 0 && max_images < num) num = max_images;  // load_dataset.hpp:37-38
  // The magic number is read and ignored, as in the reference (:26-32).  Sizes that the reference's own
  // containers would refuse (negative counts, a pixel count beyond int) are a format error here.
  if (num < 0 || r < 0 || c < 0 || (long long)r * c > 0x7fffffffLL || (long long)num * r * c > (1LL << 40)) {
    std::fclose(f);
    return VT_EFORMAT;
  }
  if (n) *n = num;
  if (rows) *rows = r;
  if (cols) *cols = c;
  int rc = 0;
  if (out) {
    const size_t total = (size_t)num * r * c;
    try {
      std::vector<unsigned char> raw(total);
      const size_t got = std::fread(raw.data(), 1, total, f);
      // a short file leaves the remaining pixels at 0, as the reference's failed reads do (:48-50)
      for (size_t i = 0; i < total; ++i) out[i] = static_cast<float>(i < got ? raw[i] : 0) / 255.0f;
    } catch (const std::bad_alloc &) {
      rc = VT_ENOMEM;
    }
  }
  std::fclose(f);
  return rc;
}

extern "C" int vt_idx_read_labels(const char *path, int max_labels, int *out, int *n) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return VT_EIO;
  int magic, num;
  if (!read_be32(f, &magic) || !read_be32(f, &num)) {
    std::fclose(f);
    return VT_EFORMAT;
  }
  if (max_labels > 0 && max_labels < num) num = max_labels;
  if (num < 0) {
    std::fclose(f);
    return VT_EFORMAT;
  }
  if (n) *n = num;
  int rc = 0;
  if (out) {
    try {
      std::vector<unsigned char> raw(num);
      const size_t got = std::fread(raw.data(), 1, num, f);
      for (int i = 0; i < num; ++i) out[i] = (size_t)i < got ? (int)raw[i] : 0;
    } catch (const std::bad_alloc &) {
      rc = VT_ENOMEM;
    }
  }
  std::fclose(f);
  return rc;
}

extern "C" int vt_loader_indices(int n, unsigned seed, int resets, int *out) {
  if (n < 0 || resets < 0 || !out) return VT_EINVAL;
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::mt19937 rng(seed);
  std::shuffle(idx.begin(), idx.end(), rng);  // constructor (load_dataset.hpp:126)
  for (int r = 0; r < resets; ++r) std::shuffle(idx.begin(), idx.end(), rng);  // reset() (:158)
  for (int i = 0; i < n; ++i) out[i] = (int)idx[i];
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Forward / backward chains
// ---------------------------------------------------------------------------------------------
namespace {
constexpr size_t kMaxProfPairs = 16384;
// Records a start event; returns the pair slot or -1 when this launcher is not being timed.
int prof_begin(vt_model *m, const char *name, hipStream_t hs) {
  if (m->prof_filter.empty() || (m->prof_filter != "*" && m->prof_filter != name)) return -1;
  if (m->prof_period > 1 && (m->prof_seen++ % m->prof_period) != 0) return -1;
  if (m->prof_used >= kMaxProfPairs) return -1;
  if (m->prof_ev.size() < 2 * (m->prof_used + 1)) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    m->prof_ev.push_back(a);
    m->prof_ev.push_back(b);
    m->prof_name.push_back(0);
  }
  int ni = -1;
  for (size_t k = 0; k < m->prof_names.size(); ++k)
    if (m->prof_names[k] == name) ni = (int)k;
  if (ni < 0) {
    m->prof_names.push_back(name);
    ni = (int)m->prof_names.size() - 1;
  }
  const int slot = (int)m->prof_used++;
  m->prof_name[slot] = ni;
  (void)hipEventRecord(m->prof_ev[2 * slot], hs);
  return slot;
}
void prof_end(vt_model *m, int slot, hipStream_t hs) {
  if (slot >= 0) (void)hipEventRecord(m->prof_ev[2 * slot + 1], hs);
}
}  // namespace

#define STEP_ON(name, st, call)                               \
  do {                                                        \
    const int slot__ = prof_begin(m, name, (hipStream_t)(st)); \
    const int r__ = (call);                                   \
    prof_end(m, slot__, (hipStream_t)(st));                   \
    if (r__ != 0) return r__;                                 \
  } while (0)
#define STEP(name, call) STEP_ON(name, s, call)

extern "C" int vt_model_profile(vt_model *m, const char *filter) {
  if (!m) return VT_EINVAL;
  m->prof_filter = filter ? filter : "";
  m->prof_period = 1;
  m->prof_seen = 0;
  const size_t slash = m->prof_filter.find('/');
  if (slash != std::string::npos) {  // "name/period"
    m->prof_period = std::max(1, std::atoi(m->prof_filter.c_str() + slash + 1));
    m->prof_filter.resize(slash);
  }
  m->prof_used = 0;
  return 0;
}

// Aggregated timings since the last call (synchronises the device).  Returns the number of
// distinct launcher names; entry idx -> name, total milliseconds, number of launches.
extern "C" int vt_model_profile_read(vt_model *m, int idx, char *name_out, int name_cap, float *total_ms, int *count) {
  if (!m) return VT_EINVAL;
  if (hipDeviceSynchronize() != hipSuccess) return VT_EINVAL;
  if (idx >= 0 && idx < (int)m->prof_names.size()) {
    float tot = 0.f;
    int cnt = 0;
    for (size_t k = 0; k < m->prof_used; ++k)
      if (m->prof_name[k] == idx) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->prof_ev[2 * k], m->prof_ev[2 * k + 1]) == hipSuccess) {
          tot += ms;
          ++cnt;
        }
      }
    if (name_out && name_cap > 0) {
      std::strncpy(name_out, m->prof_names[idx].c_str(), name_cap - 1);
      name_out[name_cap - 1] = 0;
    }
    if (total_ms) *total_ms = tot;
    if (count) *count = cnt;
  }
  return (int)m->prof_names.size();
}
// Pair k of the launches timed since vt_model_profile: launcher name, start and end in microseconds
// after the first timed launch's start (pairs on the model's own stream included, so the overlap of
// the two backward chains can be read off).  Returns the number of pairs (synchronises).
extern "C" int vt_model_profile_timeline(vt_model *m, int k, char *name_out, int name_cap, float *start_us, float *end_us) {
  if (!m) return VT_EINVAL;
  if (hipDeviceSynchronize() != hipSuccess) return VT_EINVAL;
  if (k >= 0 && (size_t)k < m->prof_used) {
    float a = 0.f, b = 0.f;
    (void)hipEventElapsedTime(&a, m->prof_ev[0], m->prof_ev[2 * k]);
    (void)hipEventElapsedTime(&b, m->prof_ev[0], m->prof_ev[2 * k + 1]);
    if (name_out && name_cap > 0) {
      std::strncpy(name_out, m->prof_names[m->prof_name[k]].c_str(), name_cap - 1);
      name_out[name_cap - 1] = 0;
    }
    if (start_us) *start_us = a * 1e3f;
    if (end_us) *end_us = b * 1e3f;
  }
  return (int)m->prof_used;
}

extern "C" int vt_model_keep_internals(vt_model *m, int on) {
  if (!m) return VT_EINVAL;
  if (m->gen) return 0;  // the general step stores both maps anyway
  if (on && !m->tokA) {
    HIPCHK(hipSetDevice(m->device));
    const size_t n = (size_t)m->maxN * m->P * L;
    if (dalloc(m, &m->tokA, n, "fwd.tok_A") != 0 || dalloc(m, &m->projA, n, "fwd.proj_a") != 0) return VT_ENOMEM;
  }
  m->keep_internals = on != 0;
  return 0;
}

extern "C" int vt_model_set_matmul_precision(vt_model *m, int bf16) {
  if (!m) return VT_EINVAL;
  if (!m->gen) return bf16 ? VT_EUNSUPPORTED : 0;  // the fused headline kernels are fp32 only
  m->gen->bf16 = bf16 != 0;
  return 0;
}

extern "C" int vt_model_set_fused_reductions(vt_model *m, int on) {
  if (!m) return VT_EINVAL;
  if (m->gen) return on ? VT_EUNSUPPORTED : 0;
  m->tails = on != 0;
  return 0;
}

extern "C" int vt_model_set_concurrency(vt_model *m, int on) {
  if (!m) return VT_EINVAL;
  m->concurrent = on != 0;
  return 0;
}

extern "C" int vt_model_set_training(vt_model *m, int training) {
  if (!m) return VT_EINVAL;
  m->training = training ? 1 : 0;
  return 0;
}

// Makes stream s wait for a deferred attention backward (and the update of its weights) still in flight.
static int attention_join(vt_model *m, hipStream_t s) {
  if (!m->attn_inflight) return 0;
  HIPCHK(hipStreamWaitEvent(s, m->ev_attn_done, 0));
  m->attn_inflight = false;
  m->attn_update_pending = false;
  return 0;
}

extern "C" int vt_model_set_deferred_attention(vt_model *m, int on) {
  if (!m) return VT_EINVAL;
  if (m->gen) return on ? VT_EUNSUPPORTED : 0;
  DeviceGuard dg(m->device);
  if (!dg.ok) return VT_EINVAL;
  VTCHK(quiesce(m));
  if (on && !m->attn_stream) {
    if (dalloc(m, &m->tokbuf[1], (size_t)m->maxN * L * D) != 0) return VT_ENOMEM;
    HIPCHK(hipStreamCreateWithFlags(&m->attn_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&m->ev_attn_go, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->ev_attn_done, hipEventDisableTiming));
  }
  m->defer_attn = on != 0;
  return 0;
}

extern "C" int vt_model_attention_join(vt_model *m, vt_stream s) {
  if (!m) return VT_EINVAL;
  return attention_join(m, (hipStream_t)s);
}

// group = 0: the n images are one mini-batch.  group > 0 (eval mode only): the n images are
// consecutive mini-batches of `group` images (the last may be short), each with its own
// BatchNorm1D statistics, processed by the same launches: what Trainer::evaluate_test computes
// batch after batch (utils/trainer.hpp:104-127), at one launch set per sweep instead of one per batch.
static int forward_impl(vt_model *m, const float *d_images, int n, int group, float *d_probs, vt_stream s) {
  if (!m || !d_images || n < 1 || n > m->maxN) return VT_EINVAL;
  if (group > 0 && group % 4 != 0) return VT_EUNSUPPORTED;
  // training mode: the groups are independent mini-batches that will be trained side by side (every BatchNorm layer,
  // BatchNorm2D included, normalises each with its own batch statistics), and all of them must be full
  if (group > 0 && group < n && m->training && (n % group != 0 || m->d.input_dim != 28 || m->gen)) return VT_EUNSUPPORTED;
  if (m->gen) {
    if (group > 0) return VT_EUNSUPPORTED;
    m->lastN = n;
    m->last_images = d_images;
    m->last_grouped = false;
    STEP("generic_forward", gen_forward(m->gen, m->arena, &m->lo, m->arena, d_images, n, m->training, d_probs, (hipStream_t)s));
    return 0;
  }
  if (group >= n) group = 0;
  if (m->defer_attn && !m->capturing && m->tokbuf[1]) {
    // the deferred attention backward of the previous step may still be reading its token matrix: this
    // forward writes the other one (and waits only if that one, too, is still being read)
    const int next = m->tok_parity ^ 1;
    if (m->attn_inflight && m->attn_tok_parity == next) VTCHK(attention_join(m, (hipStream_t)s));
    m->tok_parity = next;
    m->tok = m->tokbuf[next];
    m->named["fwd.tok_out"].first = m->tok;
  }
  const int H = m->d.input_dim, P = m->P, Ci = m->d.image_channels, K = m->K;
  const long NL = (long)n * L;
  const int nrb = (int)((NL + RB - 1) / RB);
  m->lastN = n;
  m->last_images = d_images;
  m->last_grouped = group > 0 && !m->training;
  m->last_group = m->training ? group : 0;
  const int bn_group = m->last_group;  // BatchNorm2D: per-group batch statistics in training mode, running ones otherwise
  // conv + BatchNorm2D statistics (training: batch stats; eval: running stats)
  // With m->tails the last workgroup of each producer kernel closes the statistic it feeds (vtcnn_tail.h) and the
  // separate finalize launch is skipped; grouped sweeps keep the launches (their statistics are per group).
  const bool tails = m->tails && group == 0;
  hipStream_t hs = (hipStream_t)s;
  auto fwd_tail = [&](int site, const char *mean, const char *var, float *stat, long rows_per_block, long rows_total) {
    BnFwdTail t;
    if (tails) {
      t.ticket = m->tickets + site * TAIL_WORDS;
      t.run_mean = m->P_(mean);
      t.run_var = m->P_(var);
      t.stat = stat;
      t.rows_per_block = rows_per_block;
      t.rows_total = rows_total;
    }
    return t;
  };
  STEP("conv3x3_fwd", conv3x3_fwd_tail(d_images, m->P_("conv.W"), m->P_("conv.b"), m->conv, m->training ? m->part : nullptr, n, Ci, H, H,
                                       fwd_tail(0, "bn.running_mean", "bn.running_var", m->stat_bn, P, (long)n * P), hs));
  if (m->training && tails) {
  } else if (m->training && bn_group > 0)
    STEP("bn_stats_finalize", vt_bn_stats_finalize_grouped(m->part, n, bn_group, C1, P, (long)n * P, m->P_("bn.running_mean"), m->P_("bn.running_var"),
                               m->stat_bn, m->batch_mv, s));
  else if (m->training)
    STEP("bn_stats_finalize", vt_bn_stats_finalize(m->part, n, C1, P, (long)n * P, m->P_("bn.running_mean"), m->P_("bn.running_var"), m->stat_bn, s));
  else
    STEP("bn_stats_eval", vt_bn_stats_eval(m->P_("bn.running_mean"), m->P_("bn.running_var"), C1, m->stat_bn, s));
  // tokenizer; the three BatchNorm1D layers always use batch statistics (visual_transformer.hpp:221)
  STEP("tokenizer_fwd", tokenizer_fwd_tail(m->conv, m->stat_bn, m->P_("bn.gamma"), m->P_("bn.beta"), m->P_("tok.linear1.W"),
                         m->P_("tok.linear1.b"), m->P_("tok.linear2.W"), m->P_("tok.linear2.b"), m->keep_internals ? m->tokA : nullptr, m->tokstat,
                         m->tok, m->part, n, P,
                         fwd_tail(1, "norm1.running_mean", "norm1.running_var", m->stat_n1, L, NL), hs, bn_group));
  if (tails) {
  } else if (group > 0)
    STEP("bn_stats_finalize", vt_bn_stats_finalize_grouped(m->part, n, group, D, L, NL, m->P_("norm1.running_mean"), m->P_("norm1.running_var"), m->stat_n1, m->batch_mv, s));
  else
    STEP("bn_stats_finalize", vt_bn_stats_finalize(m->part, n, D, L, NL, m->P_("norm1.running_mean"), m->P_("norm1.running_var"), m->stat_n1, s));
  if (tails)
    STEP("transformer_ffn_fwd", transformer_ffn_fwd_tail(m->tok, m->stat_n1, m->P_("norm1.gamma"), m->P_("norm1.beta"), m->P_("ffn.linear1.W"),
                                 m->P_("ffn.linear1.b"), m->P_("ffn.linear2.W"), m->P_("ffn.linear2.b"), m->a, m->hdn, m->s2, m->part, NL,
                                 fwd_tail(2, "norm2.running_mean", "norm2.running_var", m->stat_n2, RB, NL), hs));
  else
    STEP("transformer_ffn_fwd", vt_transformer_ffn_fwd_grouped(m->tok, m->stat_n1, m->P_("norm1.gamma"), m->P_("norm1.beta"), m->P_("ffn.linear1.W"),
                                 m->P_("ffn.linear1.b"), m->P_("ffn.linear2.W"), m->P_("ffn.linear2.b"), m->a, m->hdn, m->s2,
                                 m->part, NL, (long)group * L, s));
  if (tails) {
  } else if (group > 0)
    STEP("bn_stats_finalize", vt_bn_stats_finalize_grouped(m->part, nrb, group * L / RB, D, RB, NL, m->P_("norm2.running_mean"), m->P_("norm2.running_var"), m->stat_n2, m->batch_mv, s));
  else
    STEP("bn_stats_finalize", vt_bn_stats_finalize(m->part, nrb, D, RB, NL, m->P_("norm2.running_mean"), m->P_("norm2.running_var"), m->stat_n2, s));
  STEP("projector_tokens_fwd", vt_projector_tokens_fwd_grouped(m->s2, m->stat_n2, m->P_("norm2.gamma"), m->P_("norm2.beta"), m->P_("proj.linear1.W"),
                                m->P_("proj.linear2.W"), m->P_("proj.linear3.W"), m->P_("proj.linear3.b"), m->o, m->pk, n, group, s));
  STEP("projector_fwd", projector_fwd_tail(m->conv, m->stat_bn, m->P_("bn.gamma"), m->P_("bn.beta"), m->pk, m->P_("proj.downsample.W"),
                         m->P_("proj.downsample.b"), m->P_("proj.norm.running_mean"),
                         m->keep_internals ? m->projA : nullptr, m->z, m->part, n, P,
                         fwd_tail(3, "proj.norm.running_mean", "proj.norm.running_var", m->stat_pn, P, (long)n * P), hs, bn_group));
  if (tails) {
  } else if (group > 0)
    STEP("bn_stats_finalize", vt_bn_stats_finalize_grouped(m->part, n, group, C2, P, (long)n * P, m->P_("proj.norm.running_mean"), m->P_("proj.norm.running_var"),
                               m->stat_pn, m->batch_mv, s));
  else
    STEP("bn_stats_finalize", vt_bn_stats_finalize(m->part, n, C2, P, (long)n * P, m->P_("proj.norm.running_mean"), m->P_("proj.norm.running_var"),
                               m->stat_pn, s));
  STEP("head_fwd", vt_head_fwd_grouped(m->z, m->stat_pn, m->P_("proj.norm.gamma"), m->P_("proj.norm.beta"), m->P_("fc.W"), m->P_("fc.b"),
                    m->pooled, m->pool_idx, m->logits, d_probs ? d_probs : m->probs, n, H, K, group, s));
  m->last_probs = d_probs ? d_probs : m->probs;  // written in place: the backward reads it there
  return 0;
}

extern "C" int vt_model_forward(vt_model *m, const float *d_images, int n, float *d_probs, vt_stream s) {
  return forward_impl(m, d_images, n, 0, d_probs, s);
}
extern "C" int vt_model_forward_grouped(vt_model *m, const float *d_images, int n, int group, float *d_probs, vt_stream s) {
  if (group < 1) return VT_EINVAL;
  return forward_impl(m, d_images, n, group, d_probs, s);
}

// VTCNN::backward (reference model/vit_cnn.hpp:215-242): softmax -> fc -> flatten -> maxpool ->
// VisualTransformer::backward (projector -> transformer -> tokenizer.backward2,
// layers/visual_transformer.hpp:127-199) -> BatchNorm2D -> conv.
// every weight-gradient slab reduction requested inside a scope is queued and runs as one launch at flush
struct BatchScope {
  ReduceBatch b;
  bool on;
  // the general step reuses one partial buffer between reductions, so its reductions run at once
  explicit BatchScope(const vt_model *m) : on(m->gen == nullptr) { if (on) reduce_batch_begin(&b); }
  ~BatchScope() { if (on) reduce_batch_abandon(); }
};

// Backward of everything except the attention block: head, projector, feed-forward half of the
// transformer layer, tokenizer, BatchNorm2D and the convolution.  The tokenizer only needs the
// transformer's input delta (transformer_bwd_a) and the projector's input delta, so it runs here
// and the attention block, which nothing else waits for, closes the step.
// whole (general shapes only): the attention block too, on the side chain; backward_attention then has nothing left to do
static int backward_main(vt_model *m, const int *d_labels, vt_stream s, bool whole = false) {
  if (m->gen) {
    const bool fork = m->concurrent && m->side != nullptr;
    m->gen_attention_done = whole && fork;
    STEP("generic_backward", gen_backward(m->gen, m->arena, &m->lo, m->grads, m->last_images, d_labels, m->lastN,
                                          m->gen_attention_done ? 3 : 1, (hipStream_t)s, fork ? m->side : nullptr,
                                          fork ? m->ev_rows : nullptr, fork ? m->ev_join : nullptr));
    return 0;
  }
  const int n = m->lastN, H = m->d.input_dim, P = m->P, Ci = m->d.image_channels, K = m->K, Q = m->Q;
  const long NL = (long)n * L;
  const int nrb = (int)((NL + RB - 1) / RB);
  hipStream_t hs = (hipStream_t)s;
  // The side chain: the tokenizer / BatchNorm2D / conv backward (needs the projector's input delta and
  // transformer_bwd_a's token delta), then the fc weight gradient (needs only the head's delta).  Its
  // launchers write buffers nothing on the caller's stream touches before join_side().
  const bool fork = m->concurrent && m->side != nullptr;
  const vt_stream t = fork ? (vt_stream)m->side : s;
  // head: softmax/fc/pool/relu backward + projector BatchNorm1D backward sums
  // On 28x28 inputs (2x2 pooling) the delta between the two is one float per element pair, not the dense [N,P,C2]
  // tensor, three quarters of which are zeros.
  static const bool dense_delta = [] {  // measurement aid: VT_HEAD_DENSE=1 keeps the dense [N,P,C2] delta on every input size
    const char *e = getenv("VT_HEAD_DENSE");
    return e && *e && *e != '0';
  }();
  // statistics groups of the last forward (independent mini-batches side by side): sg images, gr token rows each
  const int sg = m->last_group;
  const long gr = (long)sg * L;
  const int ngroups = sg > 0 ? n / sg : 1;
  const bool windows = H == 28 && (!dense_delta || sg > 0);
  if (sg > 0 && !windows) return VT_EUNSUPPORTED;
  m->du_windows = windows;
  // closes one BatchNorm backward reduction: one finalize launch, per group where the step has groups (the groups'
  // gamma / beta gradients are then added up by a queued slab reduction)
  auto bn_bwd_close_launch = [&](vt_stream st, const float *part, int nblocks, int blocks_per_group, int C, long rows_group,
                                 long rows_total, const char *gamma, const char *beta, const float *stat, const float *in,
                                 float *gpart, float *coef) -> int {
    if (sg == 0) return vt_bn_bwd_finalize(part, nblocks, C, rows_total, m->P_(gamma), stat, in, m->G_(gamma), m->G_(beta), coef, st);
    VTCHK(vt_bn_bwd_finalize_grouped(part, nblocks, blocks_per_group, C, rows_group, rows_total, m->P_(gamma), stat, in, gpart, coef, st));
    return vt_reduce_rows(gpart, ngroups, 2 * C, 2 * C, m->G_(gamma), st);  // queued; gamma | beta are neighbours in the arena
  };
  auto bwd_tail = [&](int site, const char *gamma, const float *stat, const float *in_row0, float *coef, long rows_total) {
    BnBwdTail t;
    t.ticket = m->tickets + site * TAIL_WORDS;
    t.gamma = m->P_(gamma);
    t.stat = stat;
    t.in_row0 = in_row0;
    t.ggamma = m->G_(gamma);
    t.gbeta = m->G_((std::string(gamma).substr(0, std::string(gamma).size() - 5) + "beta").c_str());
    t.coef = coef;
    t.rows_total = rows_total;
    return t;
  };
  const bool tails = m->tails && sg == 0;
  if (windows && tails)
    STEP("head_bwd", head_bwd_windows_tail(m->last_probs, d_labels, m->P_("fc.W"), m->pool_idx, m->pooled, m->z, m->stat_pn,
                      m->P_("proj.norm.gamma"), m->P_("proj.norm.beta"), m->dlogits, m->dpair, m->part, n, H, K,
                      bwd_tail(4, "proj.norm.gamma", m->stat_pn, m->z, m->coef_pn, (long)n * P), hs));
  else if (windows)
    STEP("head_bwd", head_bwd_windows_tail(m->last_probs, d_labels, m->P_("fc.W"), m->pool_idx, m->pooled, m->z, m->stat_pn,
                      m->P_("proj.norm.gamma"), m->P_("proj.norm.beta"), m->dlogits, m->dpair, m->part, n, H, K, BnBwdTail{}, hs, sg));
  else
    STEP("head_bwd", vt_head_bwd(m->last_probs, d_labels, m->P_("fc.W"), m->pool_idx, m->z, m->stat_pn, m->P_("proj.norm.gamma"),
                      m->P_("proj.norm.beta"), m->dlogits, m->du, m->part, n, H, K, s));
  if (!(windows && tails)) {
    int per_group = 0;
    const int nblk = windows ? head_bwd_windows_blocks_grouped(n, sg, &per_group) : n;
    STEP("bn_bwd_finalize", bn_bwd_close_launch(s, m->part, nblk, per_group, C2, (long)sg * P, (long)n * P, "proj.norm.gamma", "proj.norm.beta",
                                                m->stat_pn, m->z, m->gpart_pn, m->coef_pn));
  }
  // projector
  if (windows)
    STEP("projector_bwd", projector_bwd_grouped(m->dpair, 1, m->coef_pn, m->conv, m->stat_bn, m->P_("bn.gamma"),
                           m->P_("bn.beta"), m->pk, m->P_("proj.linear1.W"), m->P_("proj.downsample.W"), m->pdx, m->dv, m->dk,
                           m->gslab, n, P, sg, hs));
  else
    STEP("projector_bwd", vt_projector_bwd(m->du, m->coef_pn, m->conv, m->stat_bn, m->P_("bn.gamma"), m->P_("bn.beta"), m->pk,
                           m->P_("proj.linear1.W"), m->P_("proj.downsample.W"), m->pdx, m->dv, m->dk, m->gslab, n, P, s));
  VTCHK(vt_reduce_rows(m->gslab, n, C1 * D, 400, m->G_("proj.linear1.W"), s));  // queued
  VTCHK(vt_reduce_rows(m->gslab + C1 * D, n, C1 * C2 + C2, 400, m->G_("proj.downsample.W"), s));
  // projector token side + transformer step a (targets = projector token delta, same bytes:
  // visual_transformer.hpp:153-157) in one launch
  // the fork event is signalled by this launch itself where the stream is not being captured: an event-record
  // packet behind it would hold up the stream's next kernel
  bool capturing = false;
  if (fork) {
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    capturing = hipStreamIsCapturing(hs, &cst) != hipSuccess || cst != hipStreamCaptureStatusNone;
    if (!capturing && m->signal_launches) vt::g_stop_event = m->ev_rows;
  }
  STEP("projector_tokens_bwd", projector_tokens_bwd_transformer_a_tail(m->o, m->dv, m->dk, m->P_("proj.linear3.W"), m->P_("proj.linear2.W"),
                                m->pdt, m->G_("proj.linear2.W"), m->sc_pt, m->s2, m->stat_n2, m->P_("tok.linear2.W"), m->part,
                                m->dT0, NL, tails ? bwd_tail(5, "norm2.gamma", m->stat_n2, m->s2, m->coef_n2, NL) : BnBwdTail{}, hs, gr));
  if (fork) {
    const bool signalled = !capturing && m->signal_launches && vt::g_stop_event == nullptr;
    vt::g_stop_event = nullptr;
    if (!signalled) HIPCHK(hipEventRecord(m->ev_rows, hs));
    HIPCHK(hipStreamWaitEvent(m->side, m->ev_rows, 0));
    m->side_open = true;
  }
  // tokenizer.backward2(projector token delta, projector input delta) + BatchNorm2D + conv, with
  // partial sums of their own (norm2's are still waiting in m->part for their finalize)
  STEP_ON("tokenizer_bwd", t, tokenizer_bwd_tail(m->conv, m->stat_bn, m->P_("bn.gamma"), m->P_("bn.beta"), m->tokstat, m->dT0,
                         m->P_("tok.linear1.W"), m->P_("tok.linear1.b"), m->pdx, m->dy, m->part_side, n, P,
                         tails ? bwd_tail(6, "bn.gamma", m->stat_bn, m->conv, m->coef_bn, (long)n * P) : BnBwdTail{}, (hipStream_t)t, sg));
  if (!tails)
    STEP_ON("bn_bwd_finalize", t, bn_bwd_close_launch(t, m->part_side, n, sg, C1, (long)sg * P, (long)n * P, "bn.gamma", "bn.beta", m->stat_bn,
                                                     m->conv, m->gpart_bn, m->coef_bn));
  STEP_ON("conv3x3_wgrad", t, conv3x3_wgrad_grouped(m->last_images, m->dy, m->coef_bn, m->gslab_conv, n, Ci, H, H, sg, (hipStream_t)t));
  VTCHK(vt_reduce_rows(m->gslab_conv, n, C1 * Ci * 9 + C1, C1 * Ci * 9 + C1, m->G_("conv.W"), s));  // queued
  // the fc weight gradient only needs the head's delta: it rides at the end of the shorter chain
  m->join_signalled = false;
  if (fork && !capturing && m->signal_launches) vt::g_stop_event = m->ev_join;  // the side chain's last kernel
  STEP_ON("fc_wgrad", t, vt_fc_wgrad(m->pooled, m->dlogits, m->G_("fc.W"), m->G_("fc.b"), m->sc_fc, n, Q, K, t));
  if (fork && !capturing && m->signal_launches) m->join_signalled = vt::g_stop_event == nullptr;
  vt::g_stop_event = nullptr;
  // feed-forward half of the transformer layer
  if (!tails)
    STEP("bn_bwd_finalize", bn_bwd_close_launch(s, m->part, nrb, (int)(gr / RB), D, gr, NL, "norm2.gamma", "norm2.beta", m->stat_n2, m->s2,
                                                m->gpart_n2, m->coef_n2));
  STEP("transformer_bwd_b", transformer_bwd_b_tail(m->o, m->pdt, m->coef_n2, m->hdn, m->a, m->tok, m->stat_n1, m->P_("ffn.linear1.W"),
                             m->P_("ffn.linear2.W"), m->ds2, m->da, m->G_("ffn.linear1.W"), m->sc_ffn, m->part, NL,
                             tails ? bwd_tail(7, "norm1.gamma", m->stat_n1, m->tok, m->coef_n1, NL) : BnBwdTail{}, hs, gr));
  if (!tails)
    STEP("bn_bwd_finalize", bn_bwd_close_launch(s, m->part, nrb, (int)(gr / RB), D, gr, NL, "norm1.gamma", "norm1.beta", m->stat_n1, m->tok,
                                                m->gpart_n1, m->coef_n1));
  // the output projection's cached input is the zero concat: its gradient is exactly zero
  // (a reduction over zero slab rows writes the zeros: it rides in the fused reduction launch and keeps
  // memset nodes, which did not replay reliably, out of the captured step graph)
  VTCHK(reduce_rows(m->gslab, 0, D * D, D * D, m->G_("mha.o.W"), hs));
  return 0;
}

// Backward of the attention block (MultiHeadAttention::backward's q/k/v weight gradients): reads the
// delta of the first residual (transformer_bwd_b) and writes only the mha.q*/k*/v* gradients.
static int backward_attention(vt_model *m, float *g_out, vt_stream s) {
  if (m->gen) {
    if (g_out) return VT_EUNSUPPORTED;
    if (m->gen_attention_done) {  // ran beside the tokenizer / convolution half (backward_main, whole)
      m->gen_attention_done = false;
      return 0;
    }
    STEP("generic_backward_attention", gen_backward(m->gen, m->arena, &m->lo, m->grads, m->last_images, nullptr, m->lastN, 2, (hipStream_t)s));
    return 0;
  }
  const int n = m->lastN, sg = m->last_group;
  const long NL = (long)n * L;
  // the attention groups are chunks of T consecutive token rows, T = the mini-batch size (the reference's permute
  // relabels, it does not move data): with statistics groups of sg images, L chunks of sg rows per group
  const int T = sg > 0 ? sg : n, chunks = sg > 0 ? L * (n / sg) : L;
  STEP("transformer_bwd_c", transformer_bwd_c_grouped(m->da, m->coef_n1, m->P_("mha.o.W"), m->tok, m->P_("mha.q0.W"), m->dr, m->dc, m->qkv, NL,
                                                      (long)sg * L, (hipStream_t)s));
  STEP("mha_bwd_rows", vt_mha_bwd_rows(m->qkv, m->dc, m->tok, m->sc_mha, T, chunks, s));
  STEP("mha_bwd_wgrad", vt_mha_bwd_wgrad(m->qkv, m->dc, m->tok, m->P_("mha.q0.W"), g_out ? g_out : m->G_("mha.q0.W"), m->sc_mha, T, chunks, s));
  return 0;
}

// Makes the caller's stream wait for the side chain; every gradient slab is complete after it.
static int join_side(vt_model *m, vt_stream s) {
  if (!m->side_open) return 0;
  m->side_open = false;
  if (!m->join_signalled) HIPCHK(hipEventRecord(m->ev_join, m->side));
  m->join_signalled = false;
  HIPCHK(hipStreamWaitEvent((hipStream_t)s, m->ev_join, 0));
  return 0;
}

static bool backward_ready(const vt_model *m) { return m && m->lastN >= 1 && m->last_images && !m->last_grouped; }

extern "C" int vt_model_backward(vt_model *m, const int *d_labels, vt_stream s) {
  if (!backward_ready(m) || !d_labels) return VT_EINVAL;
  // the previous step's deferred attention read buffers this backward is about to overwrite
  VTCHK(attention_join(m, (hipStream_t)s));
  const bool defer = m->defer_attn && !m->gen && !m->capturing && m->attn_stream != nullptr;
  BatchScope batch(m);
  VTCHK(backward_main(m, d_labels, s, m->gen != nullptr));
  if (defer) HIPCHK(hipEventRecord(m->ev_attn_go, (hipStream_t)s));  // everything the attention block reads exists
  else VTCHK(backward_attention(m, nullptr, s));
  VTCHK(join_side(m, s));
  if (batch.on) STEP("reduce_rows", reduce_batch_flush((hipStream_t)s));
  m->attention_pending = false;
  if (defer) {
    // the attention block on its own stream: it overlaps the rest of this step and the next forward
    HIPCHK(hipStreamWaitEvent(m->attn_stream, m->ev_attn_go, 0));
    BatchScope abatch(m);
    VTCHK(backward_attention(m, nullptr, (vt_stream)m->attn_stream));
    if (abatch.on) STEP_ON("reduce_rows", m->attn_stream, reduce_batch_flush(m->attn_stream));
    HIPCHK(hipEventRecord(m->ev_attn_done, m->attn_stream));
    m->attn_inflight = true;
    m->attn_update_pending = true;
    m->attn_tok_parity = m->tok_parity;
  }
  return 0;
}

static int train_step_impl(vt_model *m, const float *d_images, int n, int group, const int *d_labels, float lr, float divisor,
                           float *d_probs, vt_stream s) {
  if (!m || !d_labels || divisor == 0.f) return VT_EINVAL;
  if (group > 0 && !m->training) return VT_EINVAL;
  VTCHK(forward_impl(m, d_images, n, group, d_probs, s));
  if (m->gen || m->defer_attn) {  // no queued reductions to fold the update into / the attention update trails
    VTCHK(vt_model_backward(m, d_labels, s));
    return vt_model_sgd_step(m, lr, divisor, s);
  }
  VTCHK(attention_join(m, (hipStream_t)s));
  BatchScope batch(m);
  VTCHK(backward_main(m, d_labels, s));
  VTCHK(backward_attention(m, nullptr, s));
  VTCHK(join_side(m, s));
  STEP("reduce_rows_sgd", reduce_batch_flush_sgd(m->arena, m->grads, m->lo.n_trainable, lr, divisor, (hipStream_t)s));
  m->attention_pending = false;
  return 0;
}

extern "C" int vt_model_train_step(vt_model *m, const float *d_images, int n, const int *d_labels, float lr, float divisor,
                                   float *d_probs, vt_stream s) {
  return train_step_impl(m, d_images, n, 0, d_labels, lr, divisor, d_probs, s);
}
extern "C" int vt_model_train_step_grouped(vt_model *m, const float *d_images, int n, int group, const int *d_labels, float lr,
                                           float divisor, float *d_probs, vt_stream s) {
  if (group < 1) return VT_EINVAL;
  return train_step_impl(m, d_images, n, group, d_labels, lr, divisor, d_probs, s);
}

extern "C" int vt_model_backward_main(vt_model *m, const int *d_labels, vt_stream s) {
  if (!backward_ready(m) || !d_labels) return VT_EINVAL;
  VTCHK(attention_join(m, (hipStream_t)s));
  BatchScope batch(m);
  VTCHK(backward_main(m, d_labels, s));
  VTCHK(join_side(m, s));
  if (batch.on) STEP("reduce_rows", reduce_batch_flush((hipStream_t)s));
  m->attention_pending = true;
  return 0;
}

extern "C" int vt_model_backward_attention(vt_model *m, float *d_grads_out, vt_stream s) {
  if (!backward_ready(m) || !m->attention_pending) return VT_EINVAL;
  BatchScope batch(m);
  VTCHK(backward_attention(m, d_grads_out, s));
  if (batch.on) STEP("reduce_rows", reduce_batch_flush((hipStream_t)s));
  m->attention_pending = false;
  return 0;
}

extern "C" int vt_model_attention_range(const vt_model *m, int *first, int *count) {
  if (!m || !first || !count) return VT_EINVAL;
  const int a = m->lo.t[vt_find(&m->lo, "mha.q0.W")].offset, b = m->lo.t[vt_find(&m->lo, "mha.o.W")].offset;
  *first = a;
  *count = b - a;
  return 0;
}

extern "C" int vt_model_sgd_step(vt_model *m, float lr, float divisor, vt_stream s) {
  if (!m) return VT_EINVAL;
  if (m->attn_inflight && m->attn_update_pending) {
    // the attention gradients are still being computed on the model's attention stream: everything else is
    // updated here, their range there, right behind the kernels that produce it
    int first = 0, count = 0;
    VTCHK(vt_model_attention_range(m, &first, &count));
    STEP("sgd_update", sgd_update_hole(m->arena, m->grads, m->lo.n_trainable, lr, divisor, first, count, (hipStream_t)s));
    STEP_ON("sgd_update", m->attn_stream, vt_sgd_update(m->arena + first, m->grads + first, count, lr, divisor, 1, (vt_stream)m->attn_stream));
    HIPCHK(hipEventRecord(m->ev_attn_done, m->attn_stream));
    m->attn_update_pending = false;
    return 0;
  }
  STEP("sgd_update", vt_sgd_update(m->arena, m->grads, m->lo.n_trainable, lr, divisor, 1, s));
  return 0;
}

extern "C" int vt_model_adam_step(vt_model *m, float lr, float beta1, float beta2, float eps, float divisor, vt_stream s) {
  if (!m || divisor == 0.f) return VT_EINVAL;
  VTCHK(attention_join(m, (hipStream_t)s));
  if (!m->adam_m1) {
    HIPCHK(hipSetDevice(m->device));
    if (dalloc(m, &m->adam_m1, m->lo.n_trainable) != 0 || dalloc(m, &m->adam_m2, m->lo.n_trainable) != 0) return VT_ENOMEM;
    HIPCHK(hipMemsetAsync(m->adam_m1, 0, m->lo.n_trainable * sizeof(float), (hipStream_t)s));
    HIPCHK(hipMemsetAsync(m->adam_m2, 0, m->lo.n_trainable * sizeof(float), (hipStream_t)s));
    m->adam_step = 0;
  }
  STEP("adam_update", vt_adam_update(m->arena, m->grads, m->adam_m1, m->adam_m2, m->lo.n_trainable, lr, beta1, beta2, eps,
                                     ++m->adam_step, divisor, s));
  return 0;
}

extern "C" int vt_model_sgd_step_range(vt_model *m, float lr, float divisor, int first, int count, vt_stream s) {
  if (!m || first < 0 || count < 0 || (first & 3) || (long)first + count > m->lo.n_trainable) return VT_EINVAL;
  if (count == 0) return 0;
  VTCHK(attention_join(m, (hipStream_t)s));
  STEP("sgd_update", vt_sgd_update(m->arena + first, m->grads + first, count, lr, divisor, 1, s));
  return 0;
}

// Captures `body` (launches on m->gstream) as the model's step graph.
template <typename Body>
static int capture_graph(vt_model *m, Body body) {
  HIPCHK(hipSetDevice(m->device));
  if (!m->gstream) HIPCHK(hipStreamCreateWithFlags(&m->gstream, hipStreamNonBlocking));
  if (m->graph_exec) { (void)hipGraphExecDestroy(m->graph_exec); m->graph_exec = nullptr; }
  if (m->graph) { (void)hipGraphDestroy(m->graph); m->graph = nullptr; }
  const std::string saved = m->prof_filter;
  m->prof_filter.clear();  // no timing events inside the captured region
  VTCHK(quiesce(m));
  m->capturing = true;  // a captured step keeps its attention backward inside the graph
  HIPCHK(hipStreamBeginCapture(m->gstream, hipStreamCaptureModeThreadLocal));
  const int rc = body((vt_stream)m->gstream);
  const hipError_t e = hipStreamEndCapture(m->gstream, &m->graph);
  m->capturing = false;
  m->prof_filter = saved;
  if (rc != 0) return rc;
  if (e != hipSuccess) return (int)e;
  HIPCHK(hipGraphInstantiate(&m->graph_exec, m->graph, nullptr, nullptr, 0));
  return 0;
}

extern "C" int vt_model_graph_capture(vt_model *m, const float *x_dev, const int *y_dev, int n, float lr, float divisor) {
  if (!m || !x_dev || !y_dev || n < 1 || n > m->maxN || divisor == 0.f) return VT_EINVAL;
  return capture_graph(m, [&](vt_stream gs) {
    int rc = vt_model_forward(m, x_dev, n, nullptr, gs);
    if (rc == 0) rc = vt_model_backward(m, y_dev, gs);
    if (rc == 0) rc = vt_model_sgd_step(m, lr, divisor, gs);
    return rc;
  });
}

extern "C" int vt_model_graph_capture_forward(vt_model *m, const float *x_dev, int n, float *probs_dev) {
  if (!m || !x_dev || n < 1 || n > m->maxN) return VT_EINVAL;
  return capture_graph(m, [&](vt_stream gs) { return vt_model_forward(m, x_dev, n, probs_dev, gs); });
}

extern "C" int vt_model_graph_launch(vt_model *m, vt_stream s) {
  if (!m || !m->graph_exec) return VT_EINVAL;
  // an eager step's deferred attention backward (and the update behind it) may still be reading and writing the
  // buffers, gradients and weights the captured step is about to overwrite
  VTCHK(attention_join(m, (hipStream_t)s));
  HIPCHK(hipGraphLaunch(m->graph_exec, (hipStream_t)s));
  return 0;
}

extern "C" int vt_model_metrics(vt_model *m, const int *d_labels, float *d_loss_sum, int *d_argmax, vt_stream s) {
  if (!m || m->lastN < 1) return VT_EINVAL;
  return vt_metrics(m->last_probs ? m->last_probs : m->probs, d_labels, m->lastN, m->K, d_loss_sum, d_argmax, s);
}

extern "C" long vt_model_read(vt_model *m, const char *name, float *host, long capacity) {
  if (!m || !name) return VT_EINVAL;
  auto it = m->named.find(name);
  if (it == m->named.end()) return VT_EINVAL;
  const long per_max = it->second.second;
  // buffers sized by max_batch hold lastN samples' worth of data at the front
  long cnt = per_max;
  if (m->lastN > 0 && per_max % m->maxN == 0 && std::string(name).rfind("stat.", 0) != 0 &&
      std::string(name).rfind("coef.", 0) != 0)
    cnt = per_max / m->maxN * m->lastN;
  if (!host) return cnt;
  if (capacity < cnt) return VT_EINVAL;
  DeviceGuard dg(m->device);
  if (!dg.ok || quiesce(m) != 0) return VT_EINVAL;
  const float *src = it->second.first;
  if (src == m->probs && m->last_probs) src = m->last_probs;
  if (src == m->du && m->du_windows && m->lastN > 0) {  // the dense delta is only formed on request
    if (vt_head_expand_windows(m->dpair, m->du, m->lastN, m->d.input_dim, nullptr) != 0 ||
        hipDeviceSynchronize() != hipSuccess)
      return VT_EINVAL;
  }
  if (m->gen && src == m->gen->du && m->lastN > 0) {  // likewise for the general-shape step
    if (gen_expand_du(m->gen, m->arena, &m->lo, m->lastN, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) return VT_EINVAL;
  }
  if (m->gen && src == m->gen->z && m->lastN > 0) {  // the projector's pre-norm output may have been pooled without being stored
    if (gen_expand_z(m->gen, m->arena, &m->lo, m->lastN, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) return VT_EINVAL;
  }
  if (m->gen && src == m->gen->dz && m->lastN > 0) {  // the delta behind the projector's BatchNorm may exist as records only
    if (gen_expand_dz(m->gen, m->lastN, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) return VT_EINVAL;
  }
  if (m->gen && src == m->gen->tdx && m->lastN > 0) {  // and for the BatchNorm2D input delta it folds away
    if (gen_expand_tdx(m->gen, m->lastN, nullptr) != 0 || hipDeviceSynchronize() != hipSuccess) return VT_EINVAL;
  }
  if (hipMemcpy(host, src, cnt * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return VT_EINVAL;
  return cnt;
}
