// SelfAttention behind the reference's Layer surface (layers/self_attention.hpp:4-155): same
// constructor, forward / backward / update_weights / zero_grad / get_input_deltas, same exception
// text.  The arithmetic runs in libvtcnn_hip.so (vt_self_attention_fwd / _bwd); this class keeps the
// two bias-free DenseLayers' weights as host Tensors, drawn from Tensor::global_gen in the
// reference's order (query first, then key: self_attention.hpp:18-19), and stages data per call.
#pragma once
#include <stdexcept>
#include <vector>

#include "../../include/vtcnn_hip.h"
#include "layer.hpp"

class SelfAttention : public Layer {
  int channels_, attn_dim_;
  Tensor last_input, input_deltas;
  Tensor d_key_flat, d_query_flat;  // kept like the reference's delta_key_flat / delta_query_flat
  Tensor key_, query_, attn_;       // forward results the backward pass reads on the device
  SGD key_opt_{0.001f}, query_opt_{0.001f};  // self_attention.hpp:21-22
  hipStream_t stream_ = nullptr;

  static void check(int code, const char *what) {
    if (code != 0) throw std::runtime_error(std::string(what) + ": " + vt_error_string(code));
  }
  static Tensor xavier(int in, int out) {  // DenseLayer's initialisation (dense_layer.hpp:22-23)
    Tensor t({in, out});
    const float limit = std::sqrt(6.0f / (in + out));
    std::uniform_real_distribution<float> dist(-limit, limit);
    for (float &x : t.data) x = dist(Tensor::global_gen);
    return t;
  }

 public:
  Tensor query_weights, key_weights, grad_query_weights, grad_key_weights;

  SelfAttention(int channels, int attn_dim) : channels_(channels), attn_dim_(attn_dim) {
    query_weights = xavier(channels, attn_dim);
    key_weights = xavier(channels, attn_dim);
    grad_query_weights = Tensor::zeros({channels, attn_dim});
    grad_key_weights = Tensor::zeros({channels, attn_dim});
    vt_host::hip_check(hipStreamCreate(&stream_), "hipStreamCreate");
  }
  SelfAttention(const SelfAttention &) = delete;
  SelfAttention &operator=(const SelfAttention &) = delete;
  ~SelfAttention() override {
    if (stream_) {
      (void)hipStreamSynchronize(stream_);
      (void)hipStreamDestroy(stream_);
    }
  }

  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    const Tensor &x = inputs.at(0);  // [B, T, D]
    if (x.shape.size() != 3 || x.shape[2] != channels_) throw std::runtime_error("SelfAttention::forward: bad input shape");
    last_input = x;
    const int B = x.shape[0], T = x.shape[1], D = x.shape[2];
    key_ = Tensor({B, T, attn_dim_});
    query_ = Tensor({B, T, attn_dim_});
    attn_ = Tensor({B, T, T});
    Tensor out({B, T, D});
    check(vt_self_attention_fwd(last_input.to_device(stream_), key_weights.to_device(stream_), query_weights.to_device(stream_),
                                key_.device_buffer(), query_.device_buffer(), attn_.device_buffer(), out.device_buffer(), B, T,
                                D, attn_dim_, stream_),
          "SelfAttention::forward");
    out.from_device(stream_);
    return {out};
  }

  void backward(const Tensor *targets = nullptr, const Layer *next_layer = nullptr) override {
    Tensor delta;
    if (targets != nullptr)
      delta = *targets;
    else if (next_layer != nullptr)
      delta = next_layer->get_input_deltas();
    else
      throw std::runtime_error("SelfAttention: No hay fuente de delta para backward.");
    const Tensor &x = last_input;
    if (x.shape.size() != 3 || delta.data.size() != x.data.size()) throw std::runtime_error("SelfAttention::backward: bad delta shape");
    const int B = x.shape[0], T = x.shape[1], D = x.shape[2];
    Tensor ds({B, T, T});
    d_key_flat = Tensor({B * T, attn_dim_});
    d_query_flat = Tensor({B * T, attn_dim_});
    input_deltas = Tensor({B, T, D});
    vt_host::DeviceBuffer ws(1 << 20);
    // the forward's key / query / attn are still on the device (the reference recomputes them, :81-89)
    check(vt_self_attention_bwd(last_input.to_device(stream_), key_weights.to_device(stream_), query_weights.to_device(stream_),
                                key_.device_buffer(), query_.device_buffer(), attn_.device_buffer(), delta.to_device(stream_),
                                ds.device_buffer(), d_key_flat.device_buffer(), d_query_flat.device_buffer(),
                                grad_key_weights.device_buffer(), grad_query_weights.device_buffer(),
                                input_deltas.device_buffer(), ws.ptr, (long)ws.count, B, T, D, attn_dim_, stream_),
          "SelfAttention::backward");
    d_key_flat.from_device(stream_);
    d_query_flat.from_device(stream_);
    grad_key_weights.from_device(stream_);
    grad_query_weights.from_device(stream_);
    input_deltas.from_device(stream_);
  }

  // Like the reference (:138-145): nothing happens unless an optimizer was attached to this layer
  // itself; then each DenseLayer rescales its gradient in place and applies its own SGD(0.001)
  // (dense_layer.hpp:102-112).
  void update_weights(float batch_size) override {
    if (!optimizer) return;
    Tensor none;
    for (float &g : grad_query_weights.data) g *= 1.0f / batch_size;
    for (float &g : grad_key_weights.data) g *= 1.0f / batch_size;
    query_opt_.update(query_weights, grad_query_weights, none, none);
    key_opt_.update(key_weights, grad_key_weights, none, none);
  }
  void zero_grad() override {
    d_query_flat.fill(0.0f);
    d_key_flat.fill(0.0f);
    grad_query_weights.fill(0.0f);
    grad_key_weights.fill(0.0f);
  }
  void set_training(bool) override {}
  const Tensor &get_input_deltas() const override { return input_deltas; }
};
