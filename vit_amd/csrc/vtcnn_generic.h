// State of the general-shape VTCNN step (vtcnn_generic.hip): activation and delta buffers sized for
// max_batch samples.  Owned by the model handle, which also owns the arenas and the layout.
#pragma once
#include "vtcnn_common.h"

namespace vt {

#define HIPCHK_G(x)                         \
  do {                                      \
    hipError_t e__ = (x);                   \
    if (e__ != hipSuccess) return (int)e__; \
  } while (0)

struct GenModel {
  int Ci, H, P, C1, C2, L, D, heads, hd, K, Q, pk, maxN;
  int bf16;  // 1: products other than the gradient contractions run through bf16 operands
  // forward
  float *col, *conv, *x, *tokA, *T0, *tok, *a, *hdn, *rl, *s2, *o, *Mq, *k, *v, *pa, *z, *pooled, *logits, *probs;
  int *pool_idx;
  float *stat_bn, *stat_n1, *stat_n2, *stat_pn, *coef, *part, *ws, *spx;
  float *zmax;    // z at every pooling window's arg-max [N * Q] (what the backward needs of z when z itself is not stored)
  int z_valid;    // 1: `z` holds the projector's pre-norm output of the last forward (0: only its row 0, statistics, pooling)
  int cks;        // row stride of col: Ci * 9 rounded up to a multiple of 4 (16-byte rows; the extra columns are zero)
  float *convWp;  // conv.W with rows of cks floats (extra columns zero): the B operand that goes with col
  float *spt;  // tile records of the tokenizer's softmax over pixels ([N * P / 16][2][L]; null when P is not a multiple of 16)
  long part_floats, ws_floats;
  float *ws2, *part2, *coef2;  // scratch of the backward pass's side chain (same sizes as ws, part, coef)
  // backward
  float *dlogits, *dpool, *du, *dz, *pdx, *dsa, *dsa_tok, *U, *dv, *dk, *pdt, *d0, *ds2, *dh, *da, *dr, *dc;
  float *Qm, *Km, *Vm, *dQ, *dK, *dV, *attn, *dat, *dT0, *tdx, *csum, *dzr, *coef_pn;
  int heads_batched;  // 1: the attention backward's buffers hold every head at once (one launch per operation)
  float *dch;         // the concat delta head-major [heads][n*L][hd]
  int dz_valid;   // 1: `dz` holds the dense delta behind the projector's BatchNorm (0: only the records `dzr` were written)
  int tdx_valid;  // 1: `tdx` holds the BatchNorm2D input delta of the last backward (0: it was folded into the conv gradient)
};

// alloc(ctx, &ptr, floats, name or NULL) allocates device memory that the caller tracks
int gen_create(GenModel *g, const vt_dims *d, int maxN, int (*alloc)(void *, float **, size_t, const char *), void *ctx);
// arena: weights (read); arena_rw: the same arena, for the running statistics the forward updates
int gen_forward(GenModel *g, const float *arena, const vt_layout *lo, float *arena_rw, const float *X, int n, int training,
                float *probs_out, hipStream_t s);
// phase bit 0: everything but the attention block; bit 1: the attention block (mha.q*/k*/v* gradients)
// side, ev_fork, ev_join (all or none): the transformer half (and, with both bits, the attention block) runs on `side`
// between the two events, beside the tokenizer / convolution half on s; s has waited for it when the call returns
int gen_backward(GenModel *g, const float *arena, const vt_layout *lo, float *grads, const float *X, const int *labels, int n,
                 int phase, hipStream_t s, hipStream_t side = nullptr, hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr);
// fills g->du (the dense delta behind the pooling layer) from the last backward's window records
int gen_expand_du(GenModel *g, const float *arena, const vt_layout *lo, int n, hipStream_t s);
// fills g->dz (the dense delta behind the projector's BatchNorm) from the last backward's window records
int gen_expand_dz(GenModel *g, int n, hipStream_t s);
// fills g->z (the projector's pre-norm output) when the last forward pooled it without storing it
int gen_expand_z(GenModel *g, const float *arena, const vt_layout *lo, int n, hipStream_t s);
// fills g->tdx (the BatchNorm2D input delta) from the last backward's coefficients
int gen_expand_tdx(GenModel *g, int n, hipStream_t s);

}  // namespace vt
