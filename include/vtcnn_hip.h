/*
 * C ABI of libvtcnn_hip.so: the MI355X (gfx950) backend of the VTCNN hot path.
 *
 * The reference has no FFI layer: its boundary is the C++ header API (Layer / Tensor / Model /
 * Trainer) consumed by main.cpp.  This header is the replacement boundary: every entry point
 * names the reference function it replaces.  All functions return 0 on success, a positive
 * hipError_t, or a negative VT_E* code; no C++ types, no torch types, plain pointers and sizes.
 *
 * Two levels:
 *   1. launchers (vt_*_fwd / vt_*_bwd ...): one reference layer operation over a whole mini-batch
 *      on caller-provided DEVICE buffers and a caller-provided stream.  They never allocate and
 *      never synchronise.
 *   2. model handle (vt_model_*): owns the parameter arena and the activation workspace and
 *      chains the launchers into VTCNN::forward / backward / update_weights.
 *
 * Device tensor layouts ("token-major"): feature maps are [N, P, C] with P = H*W pixels and C
 * channels innermost; token matrices are [N*L, D] row-major; parameters use the canonical arena
 * of vtcnn_layout.h.  The per-layer launchers are compiled for the reference's own
 * hyper-parameters (main.cpp:66-78): C1=8, C2=16, L=16, D=32, heads=2, P<=1024, K<=16.  A model
 * handle of any other shape (channel widths, token count and width and head count up to 1024
 * channels, any image size that is a multiple of 14) runs a general step built from vt_gemm_f32 and
 * row-matrix kernels behind the same vt_model_* calls.
 */
#ifndef VTCNN_HIP_H
#define VTCNN_HIP_H

#include "vtcnn_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VT_EINVAL (-1)
#define VT_EUNSUPPORTED (-2)
#define VT_ENOMEM (-3)
#define VT_EIO (-4)
#define VT_EFORMAT (-5)
#define VT_ELIMIT (-6) /* size beyond the fixed capacity of an allocation-free entry point */

typedef void *vt_stream; /* hipStream_t */
typedef struct vt_model vt_model;

const char *vt_error_string(int code);
int vt_version(void);

/* ---------------------------------------------------------------------------------------------
 * Level 2: model handle.
 * ------------------------------------------------------------------------------------------- */

/* VTCNN::VTCNN (model/vit_cnn.hpp:38-101).  Allocates the arena and a workspace for mini-batches
 * of up to max_batch images on HIP device `device`.  If ext_arena / ext_grads are non-NULL they
 * are device buffers of vt_layout.n_total / n_trainable floats supplied by the caller (e.g. so a
 * collective library can all-reduce the gradient arena in place); otherwise the model allocates
 * them.  Weights are NOT initialised: call vt_model_init_seed or vt_model_set_arena. */
int vt_model_create(const vt_dims *dims, int max_batch, int device, float *ext_arena, float *ext_grads,
                    vt_model **out);
int vt_model_destroy(vt_model *m);
int vt_model_layout(const vt_model *m, vt_layout *out);
/* 1: the handle has the reference's main.cpp shape and runs the fused kernels (the only ones behind
 * vt_model_forward_grouped, the two-stream backward, deferred attention and graph capture of a step);
 * 0: any other shape, run by the general step. */
int vt_model_is_fused(const vt_model *m);

/* Weight initialisation in the reference's construction and draw order from mt19937(seed)
 * (main.cpp:12; Tensor::kaiming_normal / rand_uniform, utils/tensor/tensor_cpu.hpp:69-106;
 * order of construction model/vit_cnn.hpp:56-91).  Host-side, then uploaded. */
int vt_model_init_seed(vt_model *m, unsigned seed);
/* Host <-> device copies of the whole arena (n_total floats) / gradient arena (n_trainable). */
int vt_model_set_arena(vt_model *m, const float *host);
int vt_model_get_arena(vt_model *m, float *host);
int vt_model_get_grads(vt_model *m, float *host);
float *vt_model_arena_ptr(vt_model *m); /* device pointers */
float *vt_model_grads_ptr(vt_model *m);

/* VTCNN::save / VTCNN::load, byte-compatible v1 .bin layout (model/vit_cnn.hpp:276-546). */
int vt_model_save(vt_model *m, const char *path);
int vt_model_load(vt_model *m, const char *path);

/* Test aid: when on, forward also stores the two attention maps that the backward pass otherwise
 * recomputes ("fwd.tok_A", "fwd.proj_a", each [N,P,L]: FilterTokenizer's softmax over pixels,
 * tokenizer.hpp:36, and the projector's softmax over tokens, projector_layer.hpp:84) so that
 * vt_model_read can return them.  Allocates the two buffers on first use. */
int vt_model_keep_internals(vt_model *m, int on);
/* VTCNN::set_training (model/vit_cnn.hpp:262-269): only BatchNorm2D honours eval mode; the three
 * BatchNorm1D layers always use batch statistics (layers/visual_transformer.hpp:221). */
int vt_model_set_training(vt_model *m, int training);
/* Optional reduced-precision mode of the general-shape step: bf16 != 0 sends the projector's value projection
 * (linear3), the classifier and the backward's recomputed query / key / value projections through vt_gemm_bf16_mfma;
 * the convolution, every product upstream of an exponential (feed-forward block, projector keys and scores), the
 * delta products of the backward pass, gradient contractions, statistics and softmax stay fp32.  Predictions, loss,
 * updated weights and every gradient tensor then agree with exact arithmetic to 1e-2 instead of 1e-4 (gradients under
 * the run's own pooling decisions).  The feed-forward block and the keys can be added through the environment
 * (VT_BF16_PRODUCTS=31); their rounding is multiplied by the score magnitudes in the projector's exponential and
 * leaves seven gradient tensors at 2e-2 .. 4e-2.  Models of the headline shape run fused fp32 kernels and accept
 * only 0.  No counterpart in the reference. */
int vt_model_set_matmul_precision(vt_model *m, int bf16);
/* Launch order of the headline-shape backward.  on != 0 (the default): once the projector's deltas
 * exist, the tokenizer / BatchNorm2D / convolution backward (and the fc weight gradient after it)
 * are issued on a stream of the model's own, beside the transformer and attention backward on the
 * caller's stream, and joined before the gradient reduction; the two chains share no buffer
 * (layers/visual_transformer.hpp:153-173: tokenizer.backward2 takes the projector's deltas, not the
 * transformer's), so the result is the same bytes as with on == 0, where everything is issued on the
 * caller's stream.  General-shape models ignore it.  No counterpart in the reference. */
int vt_model_set_concurrency(vt_model *m, int on);

/* Where the eight mini-batch-wide BatchNorm reductions of a headline-shape training step are closed (four
 * forward statistics, batch_normalization_layer.hpp:152-168, four backward reductions, :187-199 / 316-326).
 * on == 0 (the default): by vt_bn_stats_finalize / vt_bn_bwd_finalize launches.  on != 0: inside the kernel that
 * produces their partial sums, by its last workgroup to finish (write-through partial records, a sharded
 * arrival counter, one acquire; no workgroup waits for another), which removes eight dependent launches per
 * step.  Measured on MI355X the arrival protocol costs what the launches cost (batch 32: 150 vs 152 us per
 * step; batch 1024: 292 vs 270 us), hence the default.  The two orders of summation differ in the last bits
 * (both sum in double precision); each is deterministic.  Grouped evaluation sweeps always use the launches.
 * General-shape models accept only 0.  No counterpart in the reference. */
int vt_model_set_fused_reductions(vt_model *m, int on);

/* Deferred attention backward (headline shape; no counterpart in the reference).  on != 0: vt_model_backward
 * issues the attention block's backward (MultiHeadAttention::backward's q/k/v weight gradients,
 * layers/multi_head_attention.hpp:109-181) on a stream of the model's own instead of the caller's, and
 * vt_model_sgd_step updates those weights on that stream right behind it; both then run beside the rest of
 * the step and the NEXT step's forward, which never reads the attention weights (the layer's output is
 * discarded by the zero concat, multi_head_attention.hpp:100).  The token matrix is double-buffered for
 * this.  The next vt_model_backward, vt_model_adam_step, every host copy of the arenas (get / set / save /
 * load / read) and graph capture wait for the deferred work first, so results are the same bytes as
 * with on == 0; a caller that reads the gradient or weight arena on the device itself (a collective)
 * calls vt_model_attention_join first, which makes stream s wait for it. */
int vt_model_set_deferred_attention(vt_model *m, int on);
int vt_model_attention_join(vt_model *m, vt_stream s);

/* VTCNN::forward (model/vit_cnn.hpp:103-213).  d_images: device [n, Ci, H, W] fp32.
 * d_probs: device [n, K] softmax output, written in place by the head kernel and read again by the
 * backward pass and vt_model_metrics, so it must stay untouched until those have run (NULL: the
 * model's own buffer is used). */
int vt_model_forward(vt_model *m, const float *d_images, int n, float *d_probs, vt_stream s);
/* VTCNN::backward (model/vit_cnn.hpp:215-242) for the mini-batch of the last forward; d_labels:
 * device int32 [n].  Leaves the summed gradients in the gradient arena (overwrites it). */
int vt_model_backward(vt_model *m, const int *d_labels, vt_stream s);
/* The same backward in two calls, for data-parallel callers that overlap their gradient exchange
 * with compute.  vt_model_backward_main leaves every gradient final except those of the attention
 * weights (mha.q*, mha.k*, mha.v*: MultiHeadAttention::backward, layers/multi_head_attention.hpp:
 * 109-181), which vt_model_backward_attention then writes; nothing else reads or writes that block,
 * and the forward pass never reads the attention weights (its output is discarded by the zero
 * concat, layers/multi_head_attention.hpp:100), so their exchange and update may trail into the next step's
 * forward.  Both calls together produce exactly the bytes of vt_model_backward.  d_grads_out: where
 * the block's gradients go (device, vt_model_attention_range's count floats), NULL = their place in
 * the gradient arena; a buffer of the caller's own keeps the arena free for an exchange in flight.
 * vt_model_attention_range returns the block's first float and length inside the arenas. */
int vt_model_backward_main(vt_model *m, const int *d_labels, vt_stream s);
int vt_model_backward_attention(vt_model *m, float *d_grads_out, vt_stream s);
int vt_model_attention_range(const vt_model *m, int *first, int *count);
/* Several reference mini-batches in one launch set: the n images are consecutive mini-batches of `group` images (a
 * multiple of 4), each normalised with its own batch statistics.
 * Eval mode: the sweep of Trainer::evaluate / evaluate_test (utils/trainer.hpp:104-127, 172-197): BatchNorm2D on the
 * running statistics, the three BatchNorm1D layers on each batch's own; the last batch may be short; the running
 * statistics end as after that loop; no backward.  The probabilities equal those of calling vt_model_forward on each
 * batch up to the rounding of the statistics (1e-6 relative).
 * Training mode (headline shape, 28x28 inputs, n a multiple of group): the forward half of vt_model_train_step_grouped;
 * BatchNorm2D, too, takes each mini-batch's own statistics, the running statistics receive the groups' updates one after
 * another in order, and vt_model_backward (or the two-call form) afterwards leaves the SUM of the mini-batches'
 * gradients in the gradient arena. */
int vt_model_forward_grouped(vt_model *m, const float *d_images, int n, int group, float *d_probs, vt_stream s);
/* VTCNN::update_weights + SGD::update + zero_grad (model/vit_cnn.hpp:244-260,
 * utils/optimizer.hpp:22-28):  g <- g * (1/divisor);  w <- w - g * lr  over the trainable region.
 * The reference always passes divisor = Trainer::batch_size = 32 (utils/trainer.hpp:53). */
int vt_model_sgd_step(vt_model *m, float lr, float divisor, vt_stream s);
/* One training iteration of Trainer::train (utils/trainer.hpp:41-53) in one call: vt_model_forward,
 * vt_model_backward and vt_model_sgd_step with the same arguments, the same bytes in the arena and in
 * d_probs afterwards, gradients cleared.  Knowing that the update follows, the headline-shape step applies it
 * inside the launch that finishes the weight-gradient reductions (one launch fewer); a data-parallel caller,
 * who must exchange gradients between backward and update, keeps the separate calls. */
int vt_model_train_step(vt_model *m, const float *d_images, int n, const int *d_labels, float lr, float divisor,
                        float *d_probs, vt_stream s);
/* Data-parallel training on one GPU ("virtual ranks"; no counterpart in the reference, which trains one mini-batch at a
 * time, utils/trainer.hpp:39-54).  The n images are n / group independent reference mini-batches of `group` images (group
 * a multiple of 4, n a multiple of group; headline shape, 28x28 inputs): each is forwarded and back-propagated exactly as
 * vt_model_train_step would do it alone from the same weights -- its own statistics in all four BatchNorm layers, its
 * own attention groups of `group` consecutive token rows, its own row-0 term in the BatchNorm backward -- in ONE launch
 * set; the gradients of all mini-batches are summed and one update w <- w - lr * sum / divisor is applied, which with
 * divisor = (n / group) * batch_size is the update of n / group data-parallel ranks (utils/trainer.hpp:53 per rank).
 * At the reference's batch of 32 a step is bound by its chain of dependent launches, not by arithmetic: 32 mini-batches
 * side by side cost about what one costs.  Running statistics are updated in group order.  group >= n is
 * vt_model_train_step. */
int vt_model_train_step_grouped(vt_model *m, const float *d_images, int n, int group, const int *d_labels, float lr,
                                float divisor, float *d_probs, vt_stream s);
/* One fused Adam pass over the trainable region (g scaled by 1/divisor first, as update_weights
 * scales it; bias-corrected moments; gradients cleared), the moments kept by the model and the step
 * count advanced by every call.  The reference trains with SGD only (utils/optimizer.hpp:14-28):
 * an extension, parity unpinned (checked against the textbook recurrences). */
int vt_model_adam_step(vt_model *m, float lr, float beta1, float beta2, float eps, float divisor, vt_stream s);
/* The same update over floats [first, first + count) of the trainable region (first a multiple of 4). */
int vt_model_sgd_step_range(vt_model *m, float lr, float divisor, int first, int count, vt_stream s);
/* One training step (forward, backward, SGD update) captured once as a HIP graph and replayed:
 * at the reference's own batch size of 32 the step is ~30 launches of a few microseconds each and
 * launch overhead, not arithmetic, sets the step time.  x_dev / y_dev / n / lr / divisor are baked
 * into the graph: the caller refills the same two device buffers before every launch (for
 * instance with vt_loader_gather).  Capture runs on a stream owned by the model; launch enqueues
 * the whole step on stream s.  Re-capturing replaces the previous graph.  No counterpart in the
 * reference. */
int vt_model_graph_capture(vt_model *m, const float *x_dev, const int *y_dev, int n, float lr, float divisor);
/* The forward pass alone as the model's graph, in the mode set by vt_model_set_training: the
 * single-image latency path of Trainer::predict (utils/trainer.hpp:199-243) and of the prediction
 * service in front of it; d_probs as in vt_model_forward.  Replaces a captured training step. */
int vt_model_graph_capture_forward(vt_model *m, const float *x_dev, int n, float *d_probs);
int vt_model_graph_launch(vt_model *m, vt_stream s);

/* Tensor::compute_loss + argmax of the last forward (tensor_cpu.hpp:1137-1173, 1294-1307):
 * d_loss_sum device float[1] = sum over samples of -log(p[label] + 1e-8); d_argmax device int32[n].
 * Either may be NULL. */
int vt_model_metrics(vt_model *m, const int *d_labels, float *d_loss_sum, int *d_argmax, vt_stream s);

/* Per-launcher timing with HIP events recorded on the caller's stream around the launchers that
 * vt_model_forward / backward / sgd_step issue.  filter: NULL or "" = off, "*" = every launcher,
 * otherwise one launcher name without the vt_ prefix (e.g. "projector_bwd"), optionally followed by
 * "/n" to time only every n-th of its launches (an event pair costs the stream about 4 microseconds).
 * vt_model_profile_read synchronises the device, returns the number of distinct names seen and,
 * for 0 <= idx < that number, the name, the summed milliseconds and the launch count since the
 * last vt_model_profile call.  No counterpart in the reference (it has no profiler hooks). */
int vt_model_profile(vt_model *m, const char *filter);
int vt_model_profile_read(vt_model *m, int idx, char *name_out, int name_cap, float *total_ms, int *count);
/* Pair k of the launches timed since vt_model_profile: launcher name, start and end in microseconds
 * after the first timed launch began (launches on the model's own stream included).  Returns the
 * number of pairs; synchronises the device. */
int vt_model_profile_timeline(vt_model *m, int k, char *name_out, int name_cap, float *start_us, float *end_us);

/* Test / tooling access to the workspace: copies the named intermediate of the last
 * forward/backward to host.  Returns the number of floats, or a negative error. */
long vt_model_read(vt_model *m, const char *name, float *host, long capacity);

/* ---------------------------------------------------------------------------------------------
 * Host-side data path (bit-exact integer / byte work; no GPU involved).
 * ------------------------------------------------------------------------------------------- */

/* Host-only forms of weight initialisation and checkpoint I/O on a host arena of
 * vt_layout.n_total floats (same code the vt_model_* entry points use).  vt_checkpoint_read
 * leaves tensors that the v1 file does not contain (transformer block, BatchNorm1D running
 * statistics) untouched, as VTCNN::load does (model/vit_cnn.hpp:396-546). */
int vt_host_init_arena(const vt_dims *dims, unsigned seed, float *arena_host);
int vt_checkpoint_write(const vt_dims *dims, const float *arena_host, const char *path);
int vt_checkpoint_read(const vt_dims *dims, float *arena_host, const char *path);

/* Checkpoint v2 = the reference's v1 bytes (model/vit_cnn.hpp:276-394) followed by a trailer that
 * the reference's loader never reaches (VTCNN::load stops after the fc layer, vit_cnn.hpp:396-546,
 * so a v2 file still loads there): magic "VTCNNv2\0", i32 2, i32 tensor count, then every arena
 * tensor that v1 omits (attention, FFN, norm1 / norm2 parameters, the BatchNorm1D running
 * statistics) as  i32 name length, name, v1 tensor record;  then the training position.
 * v1 loses the transformer block: a reloaded model restarts it from the seed-42 initialisation.
 * Reading a v1 file through the v2 entry points succeeds with state->version = 1 and the omitted
 * tensors untouched. */
typedef struct vt_train_state {
  int version;            /* out: 1 = file had no trailer, 2 = full state restored */
  int epoch;              /* epochs completed */
  long long step;         /* optimizer steps taken */
  unsigned loader_seed;   /* DataLoader seed (utils/load_dataset.hpp:117) */
  int loader_resets;      /* reset() calls so far: fixes the shuffle the next epoch starts from */
} vt_train_state;
int vt_checkpoint_write_v2(const vt_dims *dims, const float *arena_host, const vt_train_state *state, const char *path);
int vt_checkpoint_read_v2(const vt_dims *dims, float *arena_host, vt_train_state *state, const char *path);
int vt_model_save_v2(vt_model *m, const vt_train_state *state, const char *path);
int vt_model_load_v2(vt_model *m, vt_train_state *state, const char *path);

/* loadImages2D / loadLabels (utils/load_dataset.hpp:20-85): big-endian IDX headers, pixel
 * u8 / 255.0f, label u8 -> int.  Query sizes with out == NULL.  The magic number is not checked (the
 * reference reads and ignores it); VT_EIO: the file cannot be opened, VT_EFORMAT: header shorter than its
 * fields, a negative count or a pixel count beyond int, VT_ENOMEM: no host memory for the staging copy. */
int vt_idx_read_images(const char *path, int max_images, float *out, int *n, int *rows, int *cols);
int vt_idx_read_labels(const char *path, int max_labels, int *out, int *n);
/* load_image_as_vector's decode step (utils/load_image_stb.hpp:9-26: stbi_load(path, &w, &h, &n, 1)):
 * one grey byte per pixel of a PNG or baseline JPEG file.  PNG: mapped from every colour type and depth exactly as
 * that decoder maps them ((77 r + 150 g + 29 b) >> 8 for colour, alpha dropped, 16-bit samples cut to their high
 * byte).  JPEG (sequential Huffman, 8-bit): the decoder's integer inverse DCT bit for bit; the luma plane of a YCbCr
 * file, the same grey formula on an RGB-coded one.  Host only.  out == NULL queries the size; capacity = bytes
 * available at out.  VT_EIO: cannot open; VT_EFORMAT: neither a decodable PNG nor JPEG; VT_EUNSUPPORTED: a JPEG
 * file in a mode that is not decoded here (progressive or arithmetic coding, four components, subsampled luma). */
int vt_image_read_gray(const char *path, unsigned char *out, int capacity, int *width, int *height);
/* DataLoader shuffling (utils/load_dataset.hpp:112-127, 155-159): indices = iota(n), shuffled
 * once by the constructor and once more by every reset(), all from one mt19937(seed) with
 * libstdc++'s std::shuffle.  Writes the permutation after `resets` resets (0 = constructor only). */
int vt_loader_indices(int n, unsigned seed, int resets, int *out);

/* ---------------------------------------------------------------------------------------------
 * Level 1: per-layer launchers.  All pointers are device pointers.
 * ------------------------------------------------------------------------------------------- */

/* Conv2DLayer::forward, 3x3 stride 1 pad 1 (layers/conv2d_layer.hpp:35-87) with the per-sample
 * partial BatchNorm statistics fused in.  x [N,Ci,H,W]; w [C1,Ci,3,3]; b [C1];
 * conv_out [N,P,C1]; part [N][2][C1] = (sum, centred second moment) per sample (may be NULL). */
int vt_conv3x3_fwd(const float *x, const float *w, const float *b, float *conv_out, float *part, int N, int Ci,
                   int H, int W, vt_stream s);
/* Statistics step of BatchNorm2DLayer / BatchNormLayer1D::forward in training mode
 * (layers/batch_normalization_layer.hpp:152-159, 274-283): combines per-block partials into
 * stat[2][C] = (mean, 1/sqrt(var+eps)) and updates the running statistics. */
int vt_bn_stats_finalize(const float *part, int nblocks, int C, long rows_per_block, long rows_total,
                         float *running_mean, float *running_var, float *stat, vt_stream s);
/* The same for consecutive groups of blocks_per_group partial blocks (the last group may be
 * short): stat[g][2][C] per group; batch_mv[groups][2][C] is scratch; the running statistics
 * receive the groups' updates one after another, in order. */
int vt_bn_stats_finalize_grouped(const float *part, int nblocks, int blocks_per_group, int C, long rows_per_block,
                                 long rows_total, float *running_mean, float *running_var, float *stat, float *batch_mv,
                                 vt_stream s);
/* Eval-mode statistics (batch_normalization_layer.hpp:287-291): stat from the running buffers. */
int vt_bn_stats_eval(const float *running_mean, const float *running_var, int C, float *stat, vt_stream s);
/* BatchNorm2D apply + flatten/transpose (model/vit_cnn.hpp:127-134) + FilterTokenizer::forward
 * (layers/tokenizer.hpp:22-48).  Writes tokens [N*L,D], the per-sample softmax statistics
 * tok_stat [N][2][L] = (max, sum over pixels per token) the backward pass rebuilds A from, the
 * per-sample partial statistics of the tokens for the transformer's first BatchNorm1D, and,
 * when tok_A is non-null, the attention map [N,P,L] itself. */
int vt_tokenizer_fwd(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                     const float *w1, const float *b1, const float *w2, const float *b2, float *tok_A,
                     float *tok_stat, float *tokens, float *part_n1, int N, int P, vt_stream s);
/* TransformerLayer::forward up to the second BatchNorm's input (layers/transformer_layer.hpp:43-120);
 * the attention branch contributes exactly 0 in the reference (Tensor::concat returns zeros,
 * tensor_cpu.hpp:251).  rows = N*L.  Writes a, hdn (pre-ReLU), s2 = a + ffn(a) and the partial
 * statistics of s2. */
int vt_transformer_ffn_fwd(const float *tokens, const float *n1_stat, const float *n1_gamma, const float *n1_beta,
                           const float *w1, const float *b1, const float *w2, const float *b2, float *a,
                           float *hdn, float *s2, float *part_n2, long rows, vt_stream s);
/* Second BatchNorm1D apply (transformer output o) + the token side of ProjectorLayer::forward
 * (layers/projector_layer.hpp:72-88): k = o.Wk, v = o.Wv + bv and the folded score matrix
 * Mq[c][l] = sum_d Wq[c][d] k[l][d] per sample.  pk [N][8*16 + 16*16 + 16*32] = (Mq, v, k). */
int vt_projector_tokens_fwd(const float *s2, const float *n2_stat, const float *n2_gamma, const float *n2_beta,
                            const float *wq, const float *wk, const float *wv, const float *bv, float *o,
                            float *pk, int N, vt_stream s);
/* Pixel side of ProjectorLayer::forward (projector_layer.hpp:65-104): attention over tokens,
 * downsample residual.  Writes proj_a [N,P,L], z [N,P,C2] (BatchNorm1D input) and its partial stats.
 * pivot [C2]: any per-channel value near the batch mean of z (the layer's running mean is used),
 * around which the partial second moments are accumulated. */
int vt_projector_fwd(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                     const float *pk, const float *wd, const float *bd, const float *pivot, float *proj_a, float *z,
                     float *part_pn, int N, int P, vt_stream s);
/* Projector BatchNorm1D apply + ReLU (projector_layer.hpp:106-113), the [N,P,C2]->[N,C2,H,W]
 * reinterpretation (model/vit_cnn.hpp:179), MaxPool2DLayer::forward (layers/pooling_layer.hpp:18-66),
 * Flatten, fc DenseLayer::forward and SoftmaxLayer::forward (vit_cnn.hpp:185-203).
 * pool_idx: uint8 [N,Q] window slot of the maximum. */
int vt_head_fwd(const float *z, const float *pn_stat, const float *pn_gamma, const float *pn_beta, const float *wfc,
                const float *bfc, float *pooled, unsigned char *pool_idx, float *logits, float *probs, int N,
                int H, int K, vt_stream s);
/* Grouped forms of three launchers above: statistics records are per group (n1_stat / n2_stat /
 * pn_stat = [groups][2][C]); group_rows must be a multiple of 64, group counts images; 0 = one group. */
int vt_transformer_ffn_fwd_grouped(const float *tokens, const float *n1_stat, const float *n1_gamma, const float *n1_beta,
                                   const float *w1, const float *b1, const float *w2, const float *b2, float *a,
                                   float *hdn, float *s2, float *part_n2, long rows, long group_rows, vt_stream s);
int vt_projector_tokens_fwd_grouped(const float *s2, const float *n2_stat, const float *n2_gamma, const float *n2_beta,
                                    const float *wq, const float *wk, const float *wv, const float *bv, float *o,
                                    float *pk, int N, int group, vt_stream s);
int vt_head_fwd_grouped(const float *z, const float *pn_stat, const float *pn_gamma, const float *pn_beta,
                        const float *wfc, const float *bfc, float *pooled, unsigned char *pool_idx, float *logits,
                        float *probs, int N, int H, int K, int group, vt_stream s);
/* Tensor::argmax_batch + cross-entropy terms (tensor_cpu.hpp:1155-1173, 1294-1307). */
int vt_metrics(const float *probs, const int *labels, int N, int K, float *loss_sum, int *argmax, vt_stream s);
/* DataLoader::next_batch on the device (utils/load_dataset.hpp:129-151) with loadImages2D's pixel
 * conversion (:47-51) fused in: images_u8 = the IDX pixel bytes resident on the device
 * [S][pixels], labels int [S], indices = n entries of the shuffled order (vt_loader_indices);
 * x[i] = images_u8[indices[i]] / 255.0f, y[i] = labels[indices[i]] (y may be NULL).
 * pixels must be a multiple of 4. */
int vt_loader_gather(const unsigned char *images_u8, const int *labels, const int *indices, int n, int pixels, float *x,
                     int *y, vt_stream s);
/* Per-batch bookkeeping of Trainer::evaluate / evaluate_test (utils/trainer.hpp:104-127) kept on
 * the device: loss_sums[slot] = -sum log(p[label] + 1e-8), correct[slot] = number of arg-max hits,
 * confusion[truth*K + pred] += 1 (int [K*K], may be NULL; zero it before the first batch),
 * argmax int [N] (may be NULL).  Read everything back once after the last batch. */
int vt_eval_accumulate(const float *probs, const int *labels, int N, int K, int slot, float *loss_sums, int *correct,
                       int *confusion, int *argmax, vt_stream s);
/* The same for N images that are consecutive mini-batches of `group` images (the last may be
 * short): batch g of the call fills slot slot0 + g. */
int vt_eval_accumulate_grouped(const float *probs, const int *labels, int N, int K, int group, int slot0,
                               float *loss_sums, int *correct, int *confusion, int *argmax, vt_stream s);

/* SoftmaxLayer::backward with targets (layers/softmax_layer.hpp:29), fc input delta
 * (layers/dense_layer.hpp:98-99), MaxPool2DLayer::backward (pooling_layer.hpp:68-97), ReLU mask
 * (layers/relu_layer.hpp:20-33) and the partial sums of the projector BatchNorm1D backward.
 * Writes d_logits [N,K], du [N,P,C2] (masked delta) and part [N][2][C2] = (sum du, sum du*xhat). */
int vt_head_bwd(const float *probs, const int *labels, const float *wfc, const unsigned char *pool_idx,
                const float *z, const float *pn_stat, const float *pn_gamma, const float *pn_beta, float *d_logits,
                float *du, float *part, int N, int H, int K, vt_stream s);
/* fc DenseLayer::backward weight/bias gradients (dense_layer.hpp:92-96): gW = pooled^T . d_logits. */
int vt_fc_wgrad(const float *pooled, const float *d_logits, float *gw, float *gb, float *scratch, int N, int Q, int K,
                vt_stream s);
/* The same backward for 28x28 inputs (2x2 pooling) without the dense delta.  dpair [N, P*C2/2] receives one
 * float per horizontally adjacent element pair (flat offsets F, F+1 with F even) of the [N,C2,H,W] view: 0, or,
 * if the pair holds the arg-max element of its pooling window and the window's maximum was positive (ReLU
 * active), the window's fc input delta with its lowest mantissa bit replaced by the arg-max column (0 = F,
 * 1 = F+1).  du is that delta (bit cleared) at the arg-max element and zero elsewhere; vt_head_expand_windows
 * forms it.  pooled = the window maxima kept by vt_head_fwd; z is read only for a channel whose gamma is exactly
 * 0.  part receives vt_head_bwd_windows_blocks(N) records of [2][C2] partial sums (sum du, sum du*xhat), not
 * one per image: pass that count to vt_bn_bwd_finalize. */
int vt_head_bwd_windows_blocks(int N);
int vt_head_bwd_windows(const float *probs, const int *labels, const float *wfc, const unsigned char *pool_idx,
                        const float *pooled, const float *z, const float *pn_stat, const float *pn_gamma,
                        const float *pn_beta, float *d_logits, float *dpair, float *part, int N, int H, int K, vt_stream s);
int vt_head_expand_windows(const float *dpair, float *du, int N, int H, vt_stream s);
/* vt_fc_wgrad without the scratch argument and without any allocation.  No counterpart in the
 * reference: it is for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  The partial slab lives in a fixed device
 * array inside the library (3.06 MiB, held by every GPU the library has run this call on until the
 * process ends), which bounds the size: ceil(N/128) * (Q*K + K) <= 16 * (3136*16 + 16) floats
 * (N <= 2048 at the reference's Q = 3136 for any K <= 16), else VT_ELIMIT.  Same kernels and the
 * same bytes in gw / gb as vt_fc_wgrad.  The first call on a device looks up that device's array
 * address (make it outside a stream capture); calls that overlap on one GPU share the array and
 * race: serialise them on one stream. */
int vt_fc_wgrad_static(const float *pooled, const float *d_logits, float *gw, float *gb, int N, int Q, int K,
                       vt_stream s);
/* BatchNorm backward reduction step (batch_normalization_layer.hpp:187-199, 316-326): from the
 * partial sums produces grad gamma/beta and coef[2][C] = (scale, shift) such that
 * delta_in = delta * scale + shift, including the reference's row-0 variance term. */
int vt_bn_bwd_finalize(const float *part, int nblocks, int C, long rows_total, const float *gamma, const float *stat,
                       const float *in_row0, float *ggamma, float *gbeta, float *coef, vt_stream s);
/* The same reduction for consecutive groups of blocks_per_group partial blocks (several independent mini-batches side
 * by side, each normalised with its own statistics; the last group may be short): group g covers rows
 * [g*rows_per_group, ...) of the layer input `in` [rows_total, C] (its first row supplies the row-0 term), takes
 * stat[g][2][C] and writes coef[g][2][C] and gpart[g][2][C] = (grad gamma | grad beta) of that mini-batch alone; the
 * caller adds the groups' gradients (vt_reduce_rows over gpart). */
int vt_bn_bwd_finalize_grouped(const float *part, int nblocks, int blocks_per_group, int C, long rows_per_group,
                               long rows_total, const float *gamma, const float *stat, const float *in, float *gpart,
                               float *coef, vt_stream s);
/* ProjectorLayer::backward, pixel side (projector_layer.hpp:133-184, 204-213).  The attention map
 * is recomputed from conv_out and the per-sample token block pk, not read back. */
int vt_projector_bwd(const float *du, const float *pn_coef, const float *conv_out, const float *bn_stat,
                     const float *bn_gamma, const float *bn_beta, const float *pk, const float *wq,
                     const float *wd, float *pdx, float *dv, float *dk, float *gslab, int N, int P, vt_stream s);
/* vt_projector_bwd fed by vt_head_bwd_windows: dpair instead of du; P must be 784. */
int vt_projector_bwd_windows(const float *dpair, const float *pn_coef, const float *conv_out, const float *bn_stat,
                             const float *bn_gamma, const float *bn_beta, const float *pk, const float *wq,
                             const float *wd, float *pdx, float *dv, float *dk, float *gslab, int N, int P, vt_stream s);
/* ProjectorLayer::backward, token side (projector_layer.hpp:186-202): Dense backward of linear3
 * (values) and linear2 (keys) over the N*L token rows.  g_wk_wv_bv receives the contiguous
 * gradient block [proj.linear2.W | proj.linear3.W | proj.linear3.b]; pdt the token delta.
 * scratch >= ceil(rows/64) * 1552 floats. */
int vt_projector_tokens_bwd(const float *o, const float *dv, const float *dk, const float *wv, const float *wk,
                            float *pdt, float *g_wk_wv_bv, float *scratch, long rows, vt_stream s);
/* TransformerLayer::backward (layers/transformer_layer.hpp:133-207) in three steps separated by
 * the two BatchNorm1D backward reductions.
 *  a: d0 = out - token_delta (:145), partial sums for norm2, and the tokenizer's linear2 input
 *     delta dT0 = token_delta . W2tok^T (layers/tokenizer.hpp:73-75).
 *  b: norm2 backward apply, ffn linear2 / linear1 backward (no ReLU derivative, :174-181),
 *     da = dh.W1^T + ds2, partial sums for norm1.  g_ffn = [gW1 | gb1 | gW2 | gb2].
 *  c: norm1 backward apply, delta of the (zero) concat dc = dr.Wo^T
 *     (layers/multi_head_attention.hpp:131-135) and the per-head q,k,v recomputation (:144-152). */
int vt_transformer_bwd_a(const float *o, const float *pdt, const float *s2, const float *n2_stat,
                         const float *tok_w2, float *part, float *dT0, long rows, vt_stream s);
/* vt_projector_tokens_bwd and vt_transformer_bwd_a as one launch (projector_layer.hpp:186-202, then
 * transformer_layer.hpp:142-161 and tokenizer.hpp:73-75): step a only reads the token delta of its own
 * rows, so nothing device-wide sits between the two.  Same outputs, bit for bit. */
int vt_projector_tokens_bwd_transformer_a(const float *o, const float *dv, const float *dk, const float *wv,
                                          const float *wk, float *pdt, float *g_wk_wv_bv, float *scratch,
                                          const float *s2, const float *n2_stat, const float *tok_w2, float *part,
                                          float *dT0, long rows, vt_stream s);
int vt_transformer_bwd_b(const float *o, const float *pdt, const float *n2_coef, const float *hdn, const float *a,
                         const float *tok, const float *n1_stat, const float *w1, const float *w2, float *ds2,
                         float *da, float *g_ffn, float *scratch, float *part, long rows, vt_stream s);
int vt_transformer_bwd_c(const float *da, const float *n1_coef, const float *wo, const float *tok, const float *wqkv,
                         float *dr, float *dc, float *qkv, long rows, vt_stream s);
/* MultiHeadAttention::backward per-head part (multi_head_attention.hpp:139-181): gradients of the
 * per-head query / key / value Dense layers, g_qkv = [q0 | k0 | v0 | q1 | k1 | v1].  Attention
 * groups are `groups` chunks of N consecutive token rows.  The output projection's gradient is
 * identically zero in the reference (its cached input is the zero concat, :133) and the layer's
 * input delta is never used (layers/visual_transformer.hpp:160); neither is computed.
 * scratch >= heads*4*(2*32 + 3)*rows + heads*ceil(rows/64)*1536 + groups*32 floats, rows = N*groups. */
int vt_mha_bwd(const float *qkv, const float *dc, const float *tok, const float *wqkv, float *g_qkv, float *scratch,
               int N, int groups, vt_stream s);
/* The two halves of vt_mha_bwd, separately launchable (and separately timed): the per-row
 * attention sums on the matrix cores, then the merge into the six gradient tensors. */
int vt_mha_bwd_rows(const float *qkv, const float *dc, const float *tok, float *scratch, int N, int groups, vt_stream s);
int vt_mha_bwd_wgrad(const float *qkv, const float *dc, const float *tok, const float *wqkv, float *g_qkv,
                     float *scratch, int N, int groups, vt_stream s);
/* DenseLayer::backward over a row matrix (dense_layer.hpp:79-100): gW = X^T dY, gb = sum dY,
 * dX = dY W^T (optionally accumulated into dX).  in,out <= 32. */
int vt_dense_bwd_rows(const float *X, const float *W, const float *dY, float *gW, float *gb, float *dX,
                      int accumulate_dx, float *scratch, long rows, int in, int out, int relu_x, vt_stream s);
/* FilterTokenizer::backward2 (layers/tokenizer.hpp:50-107) + partial sums of BatchNorm2D backward. */
int vt_tokenizer_bwd(const float *conv_out, const float *bn_stat, const float *bn_gamma, const float *bn_beta,
                     const float *tok_stat, const float *dT0, const float *w1, const float *b1, const float *pdx,
                     float *dy, float *part, int N, int P, vt_stream s);
/* BatchNorm2D backward apply + Conv2DLayer::backward weight/bias gradients
 * (conv2d_layer.hpp:115-147); the conv input delta is never consumed and is not computed. */
int vt_conv3x3_wgrad(const float *x, const float *dy, const float *bn_coef, float *gslab, int N, int Ci, int H, int W,
                     vt_stream s);
/* Column sums of a [rows][cols] slab into out[cols] (deterministic order). */
int vt_reduce_rows(const float *slab, long rows, int cols, int ld, float *out, vt_stream s);
/* Tensor::matmul and its batched form (tensor_cpu.hpp:754-814, 816-893) behind every DenseLayer
 * (layers/dense_layer.hpp:67-100) when the contraction is a real matrix product (the scaled and wide
 * shapes):  C[b] = alpha * op(A[b]) . op(B[b]) (+ bias[n]) (+ C[b] if accumulate), b < batch.
 * Row-major, leading dimensions lda/ldb/ldc, element strides between batch members; transX != 0
 * reads the operand transposed.  Exact fp32 on the matrix cores.  workspace (may be NULL) lets a
 * long contraction with a small result be cut along K; its partial sums are added in a fixed order. */
int vt_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, long strideA,
                const float *B, int ldb, long strideB, int accumulate, float *C, int ldc, long strideC,
                const float *bias, int batch, float *workspace, long workspace_floats, vt_stream s);
/* The same product with both operands rounded to bf16 (nearest even) on their way into the matrix
 * cores and fp32 accumulation (v_mfma_f32_16x16x16_bf16): operands and result stay fp32 in memory.
 * The optional reduced-precision mode for the wide shapes; not what the parity tests at 1e-4 run. */
int vt_gemm_bf16_mfma(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, long strideA,
                      const float *B, int ldb, long strideB, int accumulate, float *C, int ldc, long strideC,
                      const float *bias, int batch, float *workspace, long workspace_floats, vt_stream s);
/* The same product at fp32 accuracy on the bf16 pipe: each operand is carried as hi = bf16(x) and
 * lo = bf16(x - hi), a product is lo.hi + hi.lo + hi.hi (three v_mfma_f32_16x16x16_bf16 with fp32
 * accumulation; relative error about 2^-17 per product, random in sign) instead of four fp32 matrix
 * instructions of four times the duration.  What the general-shape step uses for its activation x weight and
 * delta x weight products and its gradient contractions of depth >= 32 (not for products that feed an
 * exponential); VT_GEMM_EXACT=1 in the environment keeps every product on vt_gemm_f32. */
int vt_gemm_f32x3(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, long strideA,
                  const float *B, int ldb, long strideB, int accumulate, float *C, int ldc, long strideC,
                  const float *bias, int batch, float *workspace, long workspace_floats, vt_stream s);
/* A sum of up to four such products into one result, the partial sums staying in registers:
 *   C[b] = alpha * sum_t A_t[b] . op(B_t[b]) (+ bias[n]) (+ C[b] if accumulate),  A_t [M, K_t], op(B_t) [K_t, N].
 * The reference forms these sums with one Tensor::matmul and one operator+ per term (the projector output
 * layers/projector_layer.hpp:93-100; the input delta that projector and tokenizer both contribute to,
 * layers/visual_transformer.hpp:153-176); here every term's operands are read once and C is written once.
 * form: 0 = exact fp32 (vt_gemm_f32's arithmetic), 1 = bf16 operands, 2 = hi/lo split (vt_gemm_f32x3's). */
typedef struct vt_gemm_term {
  const float *A, *B;
  long strideA, strideB; /* elements between batch members (0: shared by all) */
  int K, lda, ldb, transB;
} vt_gemm_term;
int vt_gemm_sum(int form, int nterms, const vt_gemm_term *terms, int M, int N, float alpha, int accumulate, float *C,
                int ldc, long strideC, const float *bias, int batch, vt_stream s);
/* vt_gemm_sum whose result is a BatchNorm's input (the projector's pre-norm output, layers/projector_layer.hpp:93-104,
 * layers/batch_normalization_layer.hpp:152-159): the launch also leaves the statistics' partial records of C's columns,
 * col_part [batch * blocks][2][N] = per block of *rows_per_block consecutive rows (sum, sum of squared distances from the
 * block's own mean), the input of vt_bn_stats_finalize(col_part, batch * blocks, N, *rows_per_block, batch * M, ...);
 * C is not read again for them.  The second moments are accumulated around col_pivot [N] (any value near the columns'
 * means keeps them of the order of the variance, e.g. the running mean; NULL: zero).  Row-panel kernel only (form 2
 * with 65 <= N <= 128 or form 0 with 33 <= N <= 64; tall M, A operands 16-byte loadable, batch members of a whole number
 * of blocks): VT_EUNSUPPORTED otherwise, and nothing is launched. */
/* ldc == 0: the statistics without the matrix: of C only row 0 of batch member 0 is stored (N floats; the row the
 * BatchNorm backward's variance term reads, layers/batch_normalization_layer.hpp:172-200), for callers that form C again
 * where they need it (vt_gemm_sum_bn_relu_pool). */
int vt_gemm_sum_colstats(int form, int nterms, const vt_gemm_term *terms, int M, int N, float alpha, float *C, int ldc,
                         long strideC, const float *bias, int batch, const float *col_pivot, float *col_part,
                         long col_part_floats, int *rows_per_block, vt_stream s);
/* Conv2DLayer::forward, 3x3 / stride 1 / pad 1 (layers/conv2d_layer.hpp:35-87), as ONE row-panel launch in training
 * mode: the im2col rows (row = image * H * H + h * H + w, column = (ci * 3 + kh) * 3 + kw, columns 9 Ci .. ld - 1 zero)
 * are gathered from X [images][Ci][H][H] on load, written to col [images * H * H][ld] for the weight gradient
 * (conv2d_layer.hpp:115-147) and multiplied by Wp^T (Wp [Co][ld]: conv.W with rows padded to ld floats) in exact fp32:
 * C [images * H * H][Co] = col . Wp^T + bias, plus the BatchNorm partial records of C's columns as in
 * vt_gemm_sum_colstats (BatchNorm2DLayer::forward, layers/batch_normalization_layer.hpp:262-298).  H a multiple of 16,
 * 9 Ci <= ld <= 32, ld a multiple of 4, 33 <= Co <= 64, 16-byte aligned col and Wp: VT_EUNSUPPORTED otherwise, and
 * nothing is launched (callers then form col themselves and use vt_gemm_sum_colstats). */
int vt_conv3x3_rows_colstats(const float *X, int images, int Ci, int H, const float *Wp, int ld, int Co, const float *bias,
                             float *col, float *C, const float *col_pivot, float *col_part, long col_part_floats,
                             int *rows_per_block, vt_stream s);
/* The sum of products of vt_gemm_sum (alpha 1, one H*H x N matrix per batch member) taken straight through the
 * BatchNorm apply, the ReLU, the reference's reshape [P, N] -> [N, H, H] (no data movement, model/vit_cnn.hpp:179) and
 * MaxPool2DLayer::forward with pk x pk windows to 14 x 14 (layers/pooling_layer.hpp:18-66; the first maximum of a window
 * in row-major order wins), without storing the matrix:
 *   pooled [batch][N][14][14], pool_idx (position of the maximum inside its H x H plane), cmax (the matrix's own value
 *   there, what the BatchNorm backward needs of it); stat = mean [N] | 1/sqrt(var + eps) [N].
 * Row-panel kernel only: form 2, 65 <= N <= 128 and a multiple of 16, pk = 16 (H = 224), H*H a multiple of 112 rows,
 * A operands 16-byte loadable; VT_EUNSUPPORTED otherwise, and nothing is launched. */
int vt_gemm_sum_bn_relu_pool(int form, int nterms, const vt_gemm_term *terms, int H, int N, const float *bias, int batch,
                             const float *stat, const float *gamma, const float *beta, int pk, float *pooled, int *pool_idx,
                             float *cmax, vt_stream s);
/* BatchNorm2DLayer::forward's apply (layers/batch_normalization_layer.hpp:292-297) inside the first product that reads
 * its result (the tokenizer's scores, layers/tokenizer.hpp:30-34):
 *   A_out[m][k] = (A[m][k] - stat[k]) * stat[K + k] * gamma[k] + beta[k],   C = A_out . op(B) (+ bias[n])
 * exact fp32, one pass over A (A_out laid out like A and written once; the normalised matrix is not read back for the
 * product).  Row-panel kernel only: N <= 16 or 129 <= N <= 208 (one or thirteen result tiles), M >= 2048, A and A_out
 * 16-byte loadable, else VT_EUNSUPPORTED and nothing is launched. */
int vt_gemm_bn_apply(int M, int N, int K, const float *A, int lda, const float *stat, const float *gamma, const float *beta,
                     float *A_out, const float *B, int ldb, int transB, const float *bias, float *C, int ldc, vt_stream s);
/* The same launch when C's columns then go through Tensor::softmax over the ROW axis (the tokenizer's softmax over
 * pixels, layers/tokenizer.hpp:36; tensor_cpu.hpp:894-944 over dim 1): besides C it leaves, per 16-row tile t and column
 * n, the tile's share of that softmax: tile_rec[t][0][n] = max over the tile's rows of C, tile_rec[t][1][n] = sum over
 * them of exp(C - that max); tile_rec [ceil(M / 16)][2][N] floats, 16-byte aligned.  A consumer merges the records of the
 * rows it normalises over instead of reading C for them.  N a multiple of 4 besides the limits above, C 16-byte rows. */
int vt_gemm_bn_apply_tile_records(int M, int N, int K, const float *A, int lda, const float *stat, const float *gamma,
                                  const float *beta, float *A_out, const float *B, int ldb, int transB, const float *bias,
                                  float *C, int ldc, float *tile_rec, vt_stream s);
/* Products that read the delta behind the projector's BatchNorm (MaxPool2DLayer::backward + ReLU mask + BatchNorm
 * backward, layers/pooling_layer.hpp:68-97, relu_layer.hpp:20-33, batch_normalization_layer.hpp:172-200) as RECORDS
 * instead of a dense [rows, N] matrix.  With pooling windows a multiple of 16 wide, 16 adjacent elements of a row hold
 * at most one non-zero pooling delta, so the matrix is  coef[N + n] + coef[n] * delta[row][n]  with delta given per
 * row and group of 16 columns as one pair (value, column inside the group): rec [rows][N / 16][2] floats (the column
 * as an int's bits), an eighth of the dense bytes.  N a multiple of 16; rec 8-byte aligned.
 * vt_gemm_tn_records:  C[b] = A[b]^T . expand(rec[b]),  A [K, M] row-major (the weight gradient x^T . dz, the value
 *   delta attention^T . dz; layers/projector_layer.hpp:150-195).  64 < N <= any; hi/lo split arithmetic; workspace
 *   as in vt_gemm_f32.
 * vt_gemm_sum_records:  vt_gemm_sum (epi 0) or vt_gemm_softmax_rows' softmax backward (epi 2, epi_a laid out like C)
 *   with terms[0].A = rec (terms[0].K = N of the delta, a multiple of 16; terms[0].lda = K / 8 floats per row of
 *   records; terms[0].strideA in floats).  Results up to 64 columns wide with at most eight groups (K <= 128, rec
 *   16-byte aligned) apply the records row by row in fp32 (per result row: shift . op(B) plus one scaled row of op(B)
 *   per record) instead of expanding them into the matrix steps.  Row-panel kernel only: results up to 16, 33..64 or 129..208 columns wide
 *   without the epilogue, up to 16 or 129..208 with it, and op(B) of all terms within 160 KB of LDS: VT_EUNSUPPORTED
 *   otherwise, and nothing is launched. */
int vt_gemm_tn_records(int M, int N, int K, const float *A, int lda, long strideA, const float *rec, long strideRec,
                       const float *coef, float *C, int ldc, long strideC, int batch, float *workspace, long workspace_floats,
                       vt_stream s);
int vt_gemm_sum_records(int nterms, const vt_gemm_term *terms, const float *coef, int epi, const float *epi_a, int M, int N,
                        float *C, int ldc, long strideC, const float *bias, int batch, vt_stream s);
/* A product whose rows go straight through Tensor::softmax over the last axis (epi 1; the projector's attention map,
 * layers/projector_layer.hpp:80-84) or through Tensor::softmax_backward against a [batch, M, N] (epi 2: C <- a * (C -
 * sum_row a * C); layers/projector_layer.hpp:163-171, layers/tokenizer.hpp:88-91):  C[b] = epi(alpha * A[b] . op(B[b])),
 * C dense [batch, M, N] with strideC = M * N.  One launch when M is tall, N <= 16 or 129 <= N <= 208, and A is 16-byte
 * loadable (the epilogue runs on the accumulators; rows of more than 128 values use the hardware exponential, about
 * 1e-6 relative); otherwise the product followed by vt_softmax_rows / vt_softmax_bwd_rows.  form: 0 exact
 * fp32, 2 hi/lo split. */
int vt_gemm_softmax_rows(int form, int epi, const float *a, int transB, int M, int N, int K, float alpha, const float *A,
                         int lda, long strideA, const float *B, int ldb, long strideB, float *C, long strideC, int batch,
                         vt_stream s);
/* Tensor::softmax over the last axis (tensor_cpu.hpp:894-944), in place on S [rows, width], and
 * Tensor::softmax_backward (tensor_cpu.hpp:946-950): da <- a * (da - sum_last(a * da)), in place on da. */
int vt_softmax_rows(float *S, long rows, int width, vt_stream s);
int vt_softmax_bwd_rows(const float *a, float *da, long rows, int width, vt_stream s);
/* BatchNormLayer1D::forward / backward (layers/batch_normalization_layer.hpp:142-200) over a row matrix
 * in [R, C]; BatchNorm2DLayer (:262-327) is the same arithmetic on the [N*H*W, C] view of its input.
 * training != 0: batch statistics (biased variance), running = 0.1 batch + 0.9 running; else the running
 * buffers.  stat [2C] receives (mean, 1/sqrt(var + 1e-5)) and is what the backward pass takes.  backward:
 * ggamma = sum delta*xhat, gbeta = sum delta, din = the reference's input delta including its row-0
 * variance term (:197-199).  scratch: vt_batchnorm_scratch_floats(C) floats. */
/* Process-wide arithmetic of the general-shape step's products (no counterpart in the reference, whose products are
 * plain fp32 loops: Tensor::matmul, tensor_cpu.hpp:754-814): on != 0 runs every product on exact fp32 matrix instructions
 * (v_mfma_f32_16x16x4_f32), no bf16 hi/lo split products and no delta records; 0 (the default unless the environment
 * variable VT_GEMM_EXACT is set) leaves the deep feature contractions on split products.  Returns the previous
 * setting.  Together with VT_GENERAL_PATH=1 in the environment of vt_model_create (the general launchers also for
 * the shape the tuned kernels are built for) it measures what exact products cost on the headline shape. */
int vt_set_exact_products(int on);
long vt_batchnorm_scratch_floats(int C);
int vt_batchnorm_fwd(const float *in, long R, int C, const float *gamma, const float *beta, float *running_mean,
                     float *running_var, int training, float *stat, float *out, float *scratch, long scratch_floats,
                     vt_stream s);
int vt_batchnorm_bwd(const float *delta, const float *in, const float *stat, const float *gamma, long R, int C,
                     float *ggamma, float *gbeta, float *din, float *scratch, long scratch_floats, vt_stream s);
/* SelfAttention (layers/self_attention.hpp; the single-head layer the reference ships beside
 * MultiHeadAttention; VTCNN itself never instantiates it).
 * forward (:31-55): key = x.Wk, query = x.Wq (DenseLayer without bias), attn = softmax_last(key.query^T)
 * (key times query-transposed, unscaled), out = attn.x.  x, out [B,T,D]; wk, wq [D,Dk]; key, query
 * [B,T,Dk]; attn [B,T,T]: all device fp32, written by the call and kept by the caller for backward.
 * backward (:57-136): g_wk, g_wq [D,Dk] (overwritten: DenseLayer::backward assigns, dense_layer.hpp:93),
 * dx [B,T,D] = attn^T.delta + d_key.Wk^T + d_query.Wq^T.  ds [B,T,T], d_key, d_query [B,T,Dk]: scratch
 * of the caller; workspace (may be NULL) as in vt_gemm_f32. */
int vt_self_attention_fwd(const float *x, const float *wk, const float *wq, float *key, float *query, float *attn,
                          float *out, int B, int T, int D, int Dk, vt_stream s);
int vt_self_attention_bwd(const float *x, const float *wk, const float *wq, const float *key, const float *query,
                          const float *attn, const float *delta, float *ds, float *d_key, float *d_query, float *g_wk,
                          float *g_wq, float *dx, float *workspace, long workspace_floats, int B, int T, int D, int Dk,
                          vt_stream s);
/* Measurement aid, no counterpart in the reference: copies blocks * floats_per_block floats (a multiple
 * of 4) from src to dst with `blocks` workgroups.  mode 0: workgroup b streams its own contiguous region,
 * the access pattern of the one-workgroup-per-image kernels; mode 1: the same bytes in 4 KB chunks dealt
 * round-robin, so that the chunks in flight are neighbours in memory; modes 2 and 3: mode 0 with
 * nontemporal loads and stores / nontemporal stores only.  bench.py reports the mode 0 rate beside the
 * kernels' achieved bandwidth. */
int vt_probe_copy(const float *src, float *dst, int blocks, long floats_per_block, int mode, vt_stream s);
/* MultiHeadAttention::forward (layers/multi_head_attention.hpp:64-107) as a stand-alone launcher.
 * x [B,T,D]; wqkv: for each head its [D,hd] query, key and value matrices in that order (the arena order
 * mha.q0 | mha.k0 | mha.v0 | mha.q1 ...), wo [D,D]; heads [B,T,D] receives the head outputs side by side
 * (head h in columns h*hd ..), out [B,T,D] the layer output: with reference_compat != 0 exactly zero, as
 * in the reference, whose Tensor::concat returns an unfilled tensor (utils/tensor/tensor_cpu.hpp:218-256)
 * in front of a bias-free projection; with 0 the projection of the concatenated heads.
 * scratch >= 3*B*T*hd + B*T*T floats.  (The training step itself never launches this: with a zero
 * output the forward attention is dead code there, and its backward recomputes what it needs.) */
int vt_mha_fwd(const float *x, const float *wqkv, const float *wo, float *heads, float *out, float *scratch,
               long scratch_floats, int B, int T, int D, int nheads, int reference_compat, vt_stream s);
/* update_weights + SGD::update (utils/optimizer.hpp:22-28) over n floats. */
int vt_sgd_update(float *w, float *g, int n, float lr, float divisor, int zero_grad, vt_stream s);
/* Textbook Adam over the arena.  NOT in the reference (utils/optimizer.hpp has SGD only):
 * provided as an extension, parity unpinned. */
int vt_adam_update(float *w, float *g, float *m1, float *m2, int n, float lr, float beta1, float beta2, float eps,
                   int step, float divisor, vt_stream s);

/* ---------------------------------------------------------------------------------------------
 * Data parallel over the GPUs of one node, driven from ONE process (the C++ CLI's form; a host with
 * one process per GPU exchanges the gradient arena with its own collective library instead).
 * No counterpart in the reference, which trains on one device (utils/trainer.hpp:28-101): every
 * rank runs Trainer::train's forward + backward on its own mini-batch, the gradient arenas are
 * summed, and every rank calls vt_model_sgd_step with divisor ranks * batch_size, which is the
 * reference's update for one mini-batch of ranks * batch_size images except that each rank's
 * BatchNorm layers see the statistics of its own share only.
 * ------------------------------------------------------------------------------------------- */
typedef struct vt_dp vt_dp;
/* One RCCL communicator per entry of device_ids (distinct HIP devices, 1 <= ranks <= 64).  RCCL is
 * opened at run time (librccl.so.1, or the path in the environment variable VT_RCCL_LIBRARY):
 * VT_EUNSUPPORTED when it cannot be found, VT_EIO when it reports an error. */
int vt_dp_init(int ranks, const int *device_ids, vt_dp **out);
int vt_dp_destroy(vt_dp *dp);
int vt_dp_ranks(const vt_dp *dp);
/* In-place sum over the ranks of the `count` floats starting at d_bufs[r] + first (d_bufs[r]: a buffer
 * on rank r's device, e.g. vt_model_grads_ptr), issued on streams[r] behind whatever that stream
 * already holds; returns without waiting.  Splitting the arena into several calls (buckets) lets the
 * exchange of one part start while another part's backward kernels are still being issued. */
int vt_dp_allreduce_async(vt_dp *dp, float *const *d_bufs, long first, long count, const vt_stream *streams);
/* The share of ONE rank in the same exchange, for a host that drives every GPU from a thread of its own (one
 * thread issuing for all ranks is bound by its launch rate long before the GPUs are busy): every rank's
 * thread makes this call with the same first / count, in the same order; the exchange runs once all of
 * them have.  Calls for different ranks may run concurrently; calls for one rank may not. */
int vt_dp_allreduce_rank_async(vt_dp *dp, int rank, float *d_buf, long first, long count, vt_stream stream);
/* The same share as an OVERLAPPED exchange, in buckets.  vt_dp_exchange_begin orders the sum of floats
 * [first, first + count) of d_buf behind what `compute` (rank's compute stream) holds at the time of the call and runs
 * it on a stream of the exchange's own, so that kernels issued on `compute` afterwards run beside it: a caller issues
 * one bucket per group of gradients as soon as they are final (vt_model_backward_main leaves everything but the
 * attention block final: fc | projector + feed-forward | BatchNorm2D + convolution, then vt_model_backward_attention,
 * then the attention block), and vt_dp_exchange_join makes `compute` wait for every bucket begun so far before the
 * update reads the sums.  Every rank's thread issues the same buckets in the same order.  Buckets of one rank run one
 * after another on its exchange stream. */
int vt_dp_exchange_begin(vt_dp *dp, int rank, float *d_buf, long first, long count, vt_stream compute);
int vt_dp_exchange_join(vt_dp *dp, int rank, vt_stream compute);
/* Blocks the host until the exchanges issued so far have finished on every rank.  Work issued on
 * streams[r] after vt_dp_allreduce_async is ordered behind the exchange without this call. */
int vt_dp_wait(vt_dp *dp);

#ifdef __cplusplus
}
#endif
#endif /* VTCNN_HIP_H */
