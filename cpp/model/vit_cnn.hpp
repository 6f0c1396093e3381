// VTCNN on MI355X behind the reference's class surface (model/vit_cnn.hpp:14-557): same
// constructor, forward / backward / update_weights / zero_grad / set_training / get_output /
// save / load, same exceptions.  All compute is delegated to libvtcnn_hip.so (include/vtcnn_hip.h);
// this class stages mini-batches through pinned host memory and one HIP stream.
#pragma once
#include <cmath>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "../../include/vtcnn_hip.h"
#include "../utils/optimizer.hpp"
#include "model.hpp"

class VTCNN : public Model {
  vt_model *h_ = nullptr;
  vt_dims dims_{};
  int max_batch_, device_;
  hipStream_t stream_ = nullptr;
  float *d_in_ = nullptr, *d_probs_ = nullptr, *pin_in_ = nullptr, *pin_out_ = nullptr;
  int *d_labels_ = nullptr;
  int last_n_ = 0;
  float lr_ = 0.001f;  // every layer of the reference is wired to SGD(0.001f) (vit_cnn.hpp:94-100)
  std::vector<int> labels_host_;

  static void check(int code, const char *what) {
    if (code != 0) throw std::runtime_error(std::string(what) + ": " + vt_error_string(code));
  }

  // Weights are drawn from Tensor::global_gen AS IT STANDS (the reference's layers call Tensor::kaiming_normal /
  // rand_uniform in construction order, model/vit_cnn.hpp:56-91), so a program that seeds or advances the
  // generator before constructing the model gets what it would get from the reference; with the generator
  // freshly seeded with 42 (main.cpp:12) this is the arena of vt_model_init_seed(42), bit for bit.
  void init_from_global_gen() {
    vt_layout lo;
    check(vt_model_layout(h_, &lo), "VTCNN: vt_model_layout");
    std::vector<float> arena(lo.n_total, 0.0f);
    auto put = [&](const std::string &name, const Tensor &t) {
      const int k = vt_find(&lo, name.c_str());
      if (k < 0 || lo.t[k].size != (int)t.data.size()) throw std::runtime_error("VTCNN: layout has no tensor " + name);
      std::copy(t.data.begin(), t.data.end(), arena.begin() + lo.t[k].offset);
    };
    auto dense = [&](const std::string &name, int in, int out) {
      const float limit = std::sqrt(6.0f / (in + out));
      put(name, Tensor::rand_uniform({in, out}, -limit, limit));
    };
    const int Ci = dims_.image_channels, c2 = dims_.vt_channels, c1 = c2 / 2, L = dims_.tokens, D = dims_.token_channels;
    const int hd = D / dims_.heads;
    put("conv.W", Tensor::kaiming_normal({c1, Ci, 3, 3}, Ci * 9));
    dense("tok.linear1.W", c1, L);
    dense("tok.linear2.W", c1, D);
    for (int h = 0; h < dims_.heads; ++h)
      for (const char *w : {"q", "k", "v"}) dense("mha." + std::string(w) + std::to_string(h) + ".W", D, hd);
    dense("mha.o.W", D, D);
    dense("ffn.linear1.W", D, D);
    dense("ffn.linear2.W", D, D);
    dense("proj.linear1.W", c1, D);
    dense("proj.linear2.W", D, D);
    dense("proj.linear3.W", D, c2);
    dense("proj.downsample.W", c1, c2);
    dense("fc.W", c2 * 14 * 14, dims_.num_classes);
    for (const char *n : {"bn.gamma", "norm1.gamma", "norm2.gamma", "proj.norm.gamma", "bn.running_var", "norm1.running_var",
                          "norm2.running_var", "proj.norm.running_var"}) {
      const int k = vt_find(&lo, n);
      std::fill(arena.begin() + lo.t[k].offset, arena.begin() + lo.t[k].offset + lo.t[k].size, 1.0f);
    }
    check(vt_model_set_arena(h_, arena.data()), "VTCNN: vt_model_set_arena");
  }

 public:
  // transformer_enc_layers, transformer_fc_dim and transformer_dropout are accepted and ignored,
  // exactly as in the reference (visual_transformer.hpp:42, transformer_layer.hpp:18-23);
  // vt_layers_num must be 1 (the recurrent tokenizer of further layers is commented out there).
  VTCNN(int image_channels, int input_dim, int vt_layers_num, int tokens, int token_channels, int vt_channels,
        int /*transformer_enc_layers*/, int transformer_heads, int /*transformer_fc_dim*/ = 1024,
        float /*transformer_dropout*/ = 0.5f, int num_classes = 1000, int max_batch = 32, int device = 0)
      : max_batch_(max_batch), device_(device) {
    if (vt_layers_num != 1) throw std::runtime_error("VTCNN: only vt_layers_num == 1 is supported");
    dims_.image_channels = image_channels;
    dims_.input_dim = input_dim;
    dims_.tokens = tokens;
    dims_.token_channels = token_channels;
    dims_.vt_channels = vt_channels;
    dims_.heads = transformer_heads;
    dims_.num_classes = num_classes;
    vt_host::DeviceScope on_device(device_);  // stream and staging buffers live on the model's device
    check(vt_model_create(&dims_, max_batch_, device, nullptr, nullptr, &h_), "VTCNN: vt_model_create");
    init_from_global_gen();
    const size_t in_n = (size_t)max_batch_ * image_channels * input_dim * input_dim;
    vt_host::hip_check(hipStreamCreate(&stream_), "hipStreamCreate");
    vt_host::hip_check(hipMalloc((void **)&d_in_, in_n * sizeof(float)), "hipMalloc");
    vt_host::hip_check(hipMalloc((void **)&d_probs_, (size_t)max_batch_ * num_classes * sizeof(float)), "hipMalloc");
    vt_host::hip_check(hipMalloc((void **)&d_labels_, (size_t)max_batch_ * sizeof(int)), "hipMalloc");
    vt_host::hip_check(hipHostMalloc((void **)&pin_in_, in_n * sizeof(float)), "hipHostMalloc");
    vt_host::hip_check(hipHostMalloc((void **)&pin_out_, (size_t)max_batch_ * num_classes * sizeof(float)), "hipHostMalloc");
  }
  VTCNN(const VTCNN &) = delete;
  VTCNN &operator=(const VTCNN &) = delete;
  ~VTCNN() override {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    vt_model_destroy(h_);
    (void)hipFree(d_in_);
    (void)hipFree(d_probs_);
    (void)hipFree(d_labels_);
    (void)hipHostFree(pin_in_);
    (void)hipHostFree(pin_out_);
    if (stream_) (void)hipStreamDestroy(stream_);
  }

  Tensor forward(const Tensor &input) override {
    const int n = input.shape.empty() ? 0 : input.shape[0];
    const size_t per = (size_t)dims_.image_channels * dims_.input_dim * dims_.input_dim;
    if (n < 1 || n > max_batch_ || input.data.size() != per * n)
      throw std::runtime_error("VTCNN::forward: bad input shape");
    std::copy(input.data.begin(), input.data.end(), pin_in_);
    vt_host::hip_check(hipMemcpyAsync(d_in_, pin_in_, per * n * sizeof(float), hipMemcpyHostToDevice, stream_), "H2D");
    check(vt_model_forward(h_, d_in_, n, d_probs_, stream_), "VTCNN::forward");
    const size_t out_n = (size_t)n * dims_.num_classes;
    vt_host::hip_check(hipMemcpyAsync(pin_out_, d_probs_, out_n * sizeof(float), hipMemcpyDeviceToHost, stream_), "D2H");
    vt_host::hip_check(hipStreamSynchronize(stream_), "sync");
    last_n_ = n;
    output = Tensor({n, dims_.num_classes}, std::vector<float>(pin_out_, pin_out_ + out_n));
    return output;
  }

  // targets: one-hot [N,K] as the reference's Trainer passes it (utils/trainer.hpp:47-52)
  void backward(const Tensor &targets) override {
    const int K = dims_.num_classes;
    if (targets.shape.size() != 2 || targets.shape[0] != last_n_ || targets.shape[1] != K)
      throw std::runtime_error("VTCNN::backward: targets must be one-hot [N, num_classes] of the last forward");
    labels_host_.resize(last_n_);
    for (int i = 0; i < last_n_; ++i) {
      const float *row = targets.data.data() + (size_t)i * K;
      labels_host_[i] = (int)(std::max_element(row, row + K) - row);
    }
    vt_host::hip_check(hipMemcpyAsync(d_labels_, labels_host_.data(), last_n_ * sizeof(int), hipMemcpyHostToDevice, stream_),
                       "H2D labels");
    check(vt_model_backward(h_, d_labels_, stream_), "VTCNN::backward");
  }

  // g <- g/batch_size; w <- w - g*lr for every trainable tensor (vit_cnn.hpp:244-251); the fused
  // kernel also clears the gradients, so zero_grad() has nothing left to do.
  void update_weights(float batch_size) override { check(vt_model_sgd_step(h_, lr_, batch_size, stream_), "update_weights"); }
  void zero_grad() override {}
  void set_training(bool is_training) override { check(vt_model_set_training(h_, is_training ? 1 : 0), "set_training"); }
  const Tensor &get_output() const { return output; }

  void save(const std::string &filename) const {
    vt_host::hip_check(hipStreamSynchronize(stream_), "sync");
    const int rc = vt_model_save(h_, filename.c_str());
    if (rc == VT_EIO) throw std::runtime_error("No se pudo abrir el archivo para guardar el modelo: " + filename);
    check(rc, "VTCNN::save");
  }
  void load(const std::string &filename) {
    vt_host::hip_check(hipStreamSynchronize(stream_), "sync");
    const int rc = vt_model_load(h_, filename.c_str());
    if (rc == VT_EIO) throw std::runtime_error("No se pudo abrir el archivo para cargar el modelo: " + filename);
    if (rc == VT_EFORMAT) throw std::runtime_error("El checkpoint no coincide con el modelo actual (capas, tipos o formas)");
    check(rc, "VTCNN::load");
  }
  // v2 = the same v1 bytes plus a trailer with the transformer block, the BatchNorm1D statistics
  // and the training position (the reference's loader stops before the trailer).
  void save_v2(const std::string &filename, const vt_train_state &state) const {
    vt_host::hip_check(hipStreamSynchronize(stream_), "sync");
    const int rc = vt_model_save_v2(h_, &state, filename.c_str());
    if (rc == VT_EIO) throw std::runtime_error("No se pudo abrir el archivo para guardar el modelo: " + filename);
    check(rc, "VTCNN::save_v2");
  }
  vt_train_state load_v2(const std::string &filename) {
    vt_host::hip_check(hipStreamSynchronize(stream_), "sync");
    vt_train_state st{};
    const int rc = vt_model_load_v2(h_, &st, filename.c_str());
    if (rc == VT_EIO) throw std::runtime_error("No se pudo abrir el archivo para cargar el modelo: " + filename);
    if (rc == VT_EFORMAT) throw std::runtime_error("El checkpoint no coincide con el modelo actual (capas, tipos o formas)");
    check(rc, "VTCNN::load_v2");
    return st;
  }
  vt_model *handle() { return h_; }

  // ---- device-resident stepping (used by Trainer when the loader keeps its dataset on the device; the
  // reference has no counterpart: its loop copies every batch through host tensors) ------------------------
  int max_batch() const { return max_batch_; }
  int device() const { return device_; }
  const vt_dims &dims() const { return dims_; }
  float learning_rate() const { return lr_; }
  // a second model of the same shape on another device with this model's weights and statistics (data-parallel rank);
  // Tensor::global_gen is left as it was
  std::unique_ptr<VTCNN> replica_on(int device) {
    const std::mt19937 keep = Tensor::global_gen;
    std::unique_ptr<VTCNN> r(new VTCNN(dims_.image_channels, dims_.input_dim, 1, dims_.tokens, dims_.token_channels, dims_.vt_channels,
                                       0, dims_.heads, 0, 0.0f, dims_.num_classes, max_batch_, device));
    Tensor::global_gen = keep;
    r->copy_weights_from(*this);
    return r;
  }
  void copy_weights_from(VTCNN &other) {
    vt_layout lo;
    check(vt_model_layout(h_, &lo), "VTCNN: vt_model_layout");
    std::vector<float> arena(lo.n_total);
    vt_host::hip_check(hipStreamSynchronize(other.stream_), "sync");
    check(vt_model_get_arena(other.h_, arena.data()), "VTCNN: vt_model_get_arena");
    vt_host::hip_check(hipStreamSynchronize(stream_), "sync");
    check(vt_model_set_arena(h_, arena.data()), "VTCNN: vt_model_set_arena");
  }
  int num_classes() const { return dims_.num_classes; }
  hipStream_t stream() const { return stream_; }
  // one training iteration on images gathered on the device from the resident dataset: loader batch (rows
  // d_indices[0..n) of the u8 image matrix, utils/load_dataset.hpp:129-151), forward, backward,
  // update_weights(batch_size), and the batch's loss sum / correct count into slot `slot` of the two device
  // arrays (utils/trainer.hpp:39-58).  Nothing is copied to the host and nothing waits.
  void train_batch_resident(const unsigned char *d_images_u8, const int *d_all_labels, const int *d_indices, int n,
                            float batch_size, int slot, float *d_loss_sums, int *d_correct) {
    const int pixels = dims_.image_channels * dims_.input_dim * dims_.input_dim;
    check(vt_loader_gather(d_images_u8, d_all_labels, d_indices, n, pixels, d_in_, d_labels_, stream_), "vt_loader_gather");
    // forward + backward + update in one call (the update rides in the step's last launch); the probabilities stay
    // in d_probs_ for the bookkeeping below
    check(vt_model_train_step(h_, d_in_, n, d_labels_, lr_, batch_size, d_probs_, stream_), "VTCNN::train_step");
    check(vt_eval_accumulate(d_probs_, d_labels_, n, dims_.num_classes, slot, d_loss_sums, d_correct, nullptr, nullptr, stream_),
          "vt_eval_accumulate");
    last_n_ = n;
  }
  // `n / group` consecutive loader batches of `group` images as independent mini-batches in ONE launch set
  // (vt_model_train_step_grouped): each with its own BatchNorm statistics and attention groups, the gradients summed,
  // one update with divisor (n / group) * batch_size -- one step of n / group data-parallel ranks on this one GPU.
  // The batches' loss sums / correct counts go to slots slot0 .. of the two device arrays.
  void train_group_resident(const unsigned char *d_images_u8, const int *d_all_labels, const int *d_indices, int n, int group,
                            float batch_size, int slot0, float *d_loss_sums, int *d_correct) {
    const int pixels = dims_.image_channels * dims_.input_dim * dims_.input_dim;
    check(vt_loader_gather(d_images_u8, d_all_labels, d_indices, n, pixels, d_in_, d_labels_, stream_), "vt_loader_gather");
    check(vt_model_train_step_grouped(h_, d_in_, n, group, d_labels_, lr_, batch_size * (float)(n / group), d_probs_, stream_),
          "VTCNN::train_step_grouped");
    check(vt_eval_accumulate_grouped(d_probs_, d_labels_, n, dims_.num_classes, group, slot0, d_loss_sums, d_correct, nullptr,
                                     nullptr, stream_),
          "vt_eval_accumulate_grouped");
    last_n_ = n;
  }
  // the same iteration up to and including backward: the summed gradients of the batch stay in the gradient arena
  // (a data-parallel caller exchanges them before update_weights)
  void gradients_resident(const unsigned char *d_images_u8, const int *d_all_labels, const int *d_indices, int n, int slot,
                          float *d_loss_sums, int *d_correct) {
    const int pixels = dims_.image_channels * dims_.input_dim * dims_.input_dim;
    check(vt_loader_gather(d_images_u8, d_all_labels, d_indices, n, pixels, d_in_, d_labels_, stream_), "vt_loader_gather");
    check(vt_model_forward(h_, d_in_, n, d_probs_, stream_), "VTCNN::forward");
    check(vt_eval_accumulate(d_probs_, d_labels_, n, dims_.num_classes, slot, d_loss_sums, d_correct, nullptr, nullptr, stream_),
          "vt_eval_accumulate");
    check(vt_model_backward(h_, d_labels_, stream_), "VTCNN::backward");
    last_n_ = n;
  }
  // the same in two calls, for a data-parallel caller that exchanges gradients while the backward still runs:
  // after gradients_main_resident every gradient outside attention_range() is final; gradients_attention() then
  // writes the attention block's (vt_model_backward_main / vt_model_backward_attention: together the bytes of
  // vt_model_backward)
  void gradients_main_resident(const unsigned char *d_images_u8, const int *d_all_labels, const int *d_indices, int n, int slot,
                               float *d_loss_sums, int *d_correct) {
    const int pixels = dims_.image_channels * dims_.input_dim * dims_.input_dim;
    check(vt_loader_gather(d_images_u8, d_all_labels, d_indices, n, pixels, d_in_, d_labels_, stream_), "vt_loader_gather");
    check(vt_model_forward(h_, d_in_, n, d_probs_, stream_), "VTCNN::forward");
    check(vt_eval_accumulate(d_probs_, d_labels_, n, dims_.num_classes, slot, d_loss_sums, d_correct, nullptr, nullptr, stream_),
          "vt_eval_accumulate");
    check(vt_model_backward_main(h_, d_labels_, stream_), "VTCNN::backward_main");
    last_n_ = n;
  }
  void gradients_attention() { check(vt_model_backward_attention(h_, nullptr, stream_), "VTCNN::backward_attention"); }
  // forward-only over n gathered images that are consecutive loader batches of `group` images each (the last
  // may be short), every batch with its own BatchNorm1D statistics as Trainer::evaluate sees them; per-batch
  // loss sums / correct counts into slots slot0.., confusion matrix and arg-max labels (may be null) on the device
  void eval_chunk_resident(const unsigned char *d_images_u8, const int *d_all_labels, const int *d_indices, int n, int group,
                           int slot0, float *d_loss_sums, int *d_correct, int *d_confusion, int *d_argmax) {
    const int pixels = dims_.image_channels * dims_.input_dim * dims_.input_dim;
    check(vt_loader_gather(d_images_u8, d_all_labels, d_indices, n, pixels, d_in_, d_labels_, stream_), "vt_loader_gather");
    if (n > group) check(vt_model_forward_grouped(h_, d_in_, n, group, d_probs_, stream_), "VTCNN::forward_grouped");
    else check(vt_model_forward(h_, d_in_, n, d_probs_, stream_), "VTCNN::forward");
    check(vt_eval_accumulate_grouped(d_probs_, d_labels_, n, dims_.num_classes, group, slot0, d_loss_sums, d_correct, d_confusion,
                                     d_argmax, stream_),
          "vt_eval_accumulate_grouped");
    last_n_ = n;
  }
};
