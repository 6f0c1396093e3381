/*
 * Canonical flat layout of every VTCNN parameter / state tensor ("arena").
 *
 * One float array holds, in this order:
 *   [trainable tensors]  - updated by SGD, all-reduced in data-parallel training
 *   [frozen tensors]     - the filter tokenizer's two Dense layers, which the reference never
 *                          updates (layers/tokenizer.hpp:18-19 never gives them an optimizer)
 *   [state tensors]      - BatchNorm running statistics
 * Every tensor starts on a 16-byte boundary (offsets are multiples of 4 floats); padding floats
 * are zero.  A gradient arena has the same layout restricted to the trainable region.
 *
 * Tensor order inside the trainable region follows the reference's construction order
 * (model/vit_cnn.hpp:56-91, layers/visual_transformer.hpp:31-46,
 * layers/multi_head_attention.hpp:33-46, layers/transformer_layer.hpp:21-25,
 * layers/projector_layer.hpp:23-34) with the tokenizer moved to the frozen region.
 *
 * Shapes are the reference's: Dense weights are [in, out] row-major (layers/dense_layer.hpp:23),
 * conv weights [C1, Ci, 3, 3] (layers/conv2d_layer.hpp:29).
 */
#ifndef VTCNN_LAYOUT_H
#define VTCNN_LAYOUT_H

#ifdef __cplusplus
extern "C" {
#endif

#define VT_MAX_HEADS 8
#define VT_MAX_TENSORS 96
#define VT_NAME_LEN 40

/* Model hyper-parameters, named as in the reference's VTCNN constructor
 * (model/vit_cnn.hpp:38-49).  conv channels C1 = vt_channels / 2 (vit_cnn.hpp:49); pool kernel =
 * input_dim / 14 (vit_cnn.hpp:53); fc input = vt_channels * 14 * 14 (vit_cnn.hpp:91). */
typedef struct vt_dims {
  int image_channels; /* Ci */
  int input_dim;      /* H = W */
  int tokens;         /* L */
  int token_channels; /* D (= attention dim) */
  int vt_channels;    /* C2 */
  int heads;          /* h, head_dim d = D / h */
  int num_classes;    /* K */
} vt_dims;

typedef struct vt_tensor_info {
  char name[VT_NAME_LEN];
  int offset; /* in floats from the start of the arena */
  int size;   /* number of floats */
  int ndim;
  int shape[4];
  int kind; /* 0 trainable, 1 frozen, 2 state */
} vt_tensor_info;

typedef struct vt_layout {
  int n_tensors;
  int n_trainable; /* floats in the trainable region, padding included */
  int n_total;     /* floats in the whole arena */
  int n_trainable_exact; /* trainable floats without padding */
  vt_tensor_info t[VT_MAX_TENSORS];
} vt_layout;

static inline int vt__add(vt_layout *lo, int *off, const char *name, int kind, int ndim, int s0, int s1,
                          int s2, int s3) {
  vt_tensor_info *ti = &lo->t[lo->n_tensors++];
  int i = 0;
  for (; name[i] && i < VT_NAME_LEN - 1; ++i) ti->name[i] = name[i];
  for (; i < VT_NAME_LEN; ++i) ti->name[i] = 0;
  ti->ndim = ndim;
  ti->shape[0] = s0; ti->shape[1] = s1; ti->shape[2] = s2; ti->shape[3] = s3;
  ti->size = s0 * (ndim > 1 ? s1 : 1) * (ndim > 2 ? s2 : 1) * (ndim > 3 ? s3 : 1);
  ti->offset = *off;
  ti->kind = kind;
  *off += (ti->size + 3) & ~3;
  return ti->offset;
}

static inline void vt__name(char *dst, const char *a, int h, const char *b) {
  int i = 0;
  for (; *a; ++a) dst[i++] = *a;
  dst[i++] = (char)('0' + h);
  for (; *b; ++b) dst[i++] = *b;
  dst[i] = 0;
}

/* Returns 0 on success, -1 if the dims are not representable. */
static inline int vt_make_layout(const vt_dims *d, vt_layout *lo) {
  const int Ci = d->image_channels, C2 = d->vt_channels, C1 = C2 / 2, L = d->tokens, D = d->token_channels;
  const int h = d->heads, K = d->num_classes;
  if (Ci < 1 || C1 < 1 || L < 1 || D < 1 || h < 1 || h > VT_MAX_HEADS || D % h != 0 || K < 1 ||
      d->input_dim < 14 || d->input_dim % 14 != 0)
    return -1;
  const int hd = D / h, Q = C2 * 14 * 14;
  int off = 0;
  char nm[VT_NAME_LEN];
  lo->n_tensors = 0;
  vt__add(lo, &off, "conv.W", 0, 4, C1, Ci, 3, 3);
  vt__add(lo, &off, "conv.b", 0, 1, C1, 1, 1, 1);
  vt__add(lo, &off, "bn.gamma", 0, 1, C1, 1, 1, 1);
  vt__add(lo, &off, "bn.beta", 0, 1, C1, 1, 1, 1);
  for (int i = 0; i < h; ++i) {
    vt__name(nm, "mha.q", i, ".W"); vt__add(lo, &off, nm, 0, 2, D, hd, 1, 1);
    vt__name(nm, "mha.k", i, ".W"); vt__add(lo, &off, nm, 0, 2, D, hd, 1, 1);
    vt__name(nm, "mha.v", i, ".W"); vt__add(lo, &off, nm, 0, 2, D, hd, 1, 1);
  }
  vt__add(lo, &off, "mha.o.W", 0, 2, D, D, 1, 1);
  vt__add(lo, &off, "ffn.linear1.W", 0, 2, D, D, 1, 1);
  vt__add(lo, &off, "ffn.linear1.b", 0, 1, D, 1, 1, 1);
  vt__add(lo, &off, "ffn.linear2.W", 0, 2, D, D, 1, 1);
  vt__add(lo, &off, "ffn.linear2.b", 0, 1, D, 1, 1, 1);
  vt__add(lo, &off, "norm1.gamma", 0, 1, D, 1, 1, 1);
  vt__add(lo, &off, "norm1.beta", 0, 1, D, 1, 1, 1);
  vt__add(lo, &off, "norm2.gamma", 0, 1, D, 1, 1, 1);
  vt__add(lo, &off, "norm2.beta", 0, 1, D, 1, 1, 1);
  vt__add(lo, &off, "proj.linear1.W", 0, 2, C1, D, 1, 1);
  vt__add(lo, &off, "proj.linear2.W", 0, 2, D, D, 1, 1);
  vt__add(lo, &off, "proj.linear3.W", 0, 2, D, C2, 1, 1);
  vt__add(lo, &off, "proj.linear3.b", 0, 1, C2, 1, 1, 1);
  vt__add(lo, &off, "proj.norm.gamma", 0, 1, C2, 1, 1, 1);
  vt__add(lo, &off, "proj.norm.beta", 0, 1, C2, 1, 1, 1);
  vt__add(lo, &off, "proj.downsample.W", 0, 2, C1, C2, 1, 1);
  vt__add(lo, &off, "proj.downsample.b", 0, 1, C2, 1, 1, 1);
  vt__add(lo, &off, "fc.W", 0, 2, Q, K, 1, 1);
  vt__add(lo, &off, "fc.b", 0, 1, K, 1, 1, 1);
  lo->n_trainable = off;
  lo->n_trainable_exact = 0;
  for (int i = 0; i < lo->n_tensors; ++i) lo->n_trainable_exact += lo->t[i].size;
  vt__add(lo, &off, "tok.linear1.W", 1, 2, C1, L, 1, 1);
  vt__add(lo, &off, "tok.linear1.b", 1, 1, L, 1, 1, 1);
  vt__add(lo, &off, "tok.linear2.W", 1, 2, C1, D, 1, 1);
  vt__add(lo, &off, "tok.linear2.b", 1, 1, D, 1, 1, 1);
  vt__add(lo, &off, "bn.running_mean", 2, 1, C1, 1, 1, 1);
  vt__add(lo, &off, "bn.running_var", 2, 1, C1, 1, 1, 1);
  vt__add(lo, &off, "norm1.running_mean", 2, 1, D, 1, 1, 1);
  vt__add(lo, &off, "norm1.running_var", 2, 1, D, 1, 1, 1);
  vt__add(lo, &off, "norm2.running_mean", 2, 1, D, 1, 1, 1);
  vt__add(lo, &off, "norm2.running_var", 2, 1, D, 1, 1, 1);
  vt__add(lo, &off, "proj.norm.running_mean", 2, 1, C2, 1, 1, 1);
  vt__add(lo, &off, "proj.norm.running_var", 2, 1, C2, 1, 1, 1);
  lo->n_total = off;
  return 0;
}

static inline int vt_find(const vt_layout *lo, const char *name) {
  for (int i = 0; i < lo->n_tensors; ++i) {
    const char *a = lo->t[i].name, *b = name;
    while (*a && *a == *b) { ++a; ++b; }
    if (*a == 0 && *b == 0) return i;
  }
  return -1;
}

#ifdef __cplusplus
}
#endif
#endif /* VTCNN_LAYOUT_H */
