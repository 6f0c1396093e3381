// VTCNN on MI355X behind the reference's class surface (model/vit_cnn.hpp:14-557): same
// constructor, forward / backward / update_weights / zero_grad / set_training / get_output /
// save / load, same exceptions.  All compute is delegated to libvtcnn_hip.so (include/vtcnn_hip.h);
// this class stages mini-batches through pinned host memory and one HIP stream.
#pragma once
#include <string>
#include <vector>

#include "../../include/vtcnn_hip.h"
#include "../utils/optimizer.hpp"
#include "model.hpp"

class VTCNN : public Model {
  vt_model *h_ = nullptr;
  vt_dims dims_{};
  int max_batch_;
  hipStream_t stream_ = nullptr;
  float *d_in_ = nullptr, *d_probs_ = nullptr, *pin_in_ = nullptr, *pin_out_ = nullptr;
  int *d_labels_ = nullptr;
  int last_n_ = 0;
  float lr_ = 0.001f;  // every layer of the reference is wired to SGD(0.001f) (vit_cnn.hpp:94-100)
  std::vector<int> labels_host_;

  static void check(int code, const char *what) {
    if (code != 0) throw std::runtime_error(std::string(what) + ": " + vt_error_string(code));
  }

 public:
  // transformer_enc_layers, transformer_fc_dim and transformer_dropout are accepted and ignored,
  // exactly as in the reference (visual_transformer.hpp:42, transformer_layer.hpp:18-23);
  // vt_layers_num must be 1 (the recurrent tokenizer of further layers is commented out there).
  VTCNN(int image_channels, int input_dim, int vt_layers_num, int tokens, int token_channels, int vt_channels,
        int /*transformer_enc_layers*/, int transformer_heads, int /*transformer_fc_dim*/ = 1024,
        float /*transformer_dropout*/ = 0.5f, int num_classes = 1000, int max_batch = 32, int device = 0)
      : max_batch_(max_batch) {
    if (vt_layers_num != 1) throw std::runtime_error("VTCNN: only vt_layers_num == 1 is supported");
    dims_.image_channels = image_channels;
    dims_.input_dim = input_dim;
    dims_.tokens = tokens;
    dims_.token_channels = token_channels;
    dims_.vt_channels = vt_channels;
    dims_.heads = transformer_heads;
    dims_.num_classes = num_classes;
    check(vt_model_create(&dims_, max_batch_, device, nullptr, nullptr, &h_), "VTCNN: vt_model_create");
    // weights come from Tensor::global_gen's seed (42 in main.cpp:12) in construction order
    check(vt_model_init_seed(h_, 42u), "VTCNN: vt_model_init_seed");
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
};
