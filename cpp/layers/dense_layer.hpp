// DenseLayer behind the reference's surface (layers/dense_layer.hpp:4-150): same public fields, constructor,
// get/set_parameters, forward / backward / update_weights, "Dense" type string.  The three matrix products
// of a step (x.W, x^T.delta, delta.W^T) are vt_gemm_f32 launches; nothing is transposed on the host.
#pragma once
#include <cmath>

#include "layer.hpp"

namespace vt_host {
// update_weights of every reference layer: g <- g * (1/batch_size) in place (Tensor::operator/ multiplies by
// the reciprocal, tensor_cpu.hpp:1086-1097), then the layer's optimizer (dense_layer.hpp:102-112)
inline void scaled_update(Optimizer *opt, Tensor &w, Tensor &gw, Tensor &b, Tensor &gb, float batch_size) {
  if (!opt) return;
  const float r = 1.0f / batch_size;
  for (float &g : gw.data) g *= r;
  for (float &g : gb.data) g *= r;
  opt->update(w, gw, b, gb);
}
}  // namespace vt_host

class DenseLayer : public Layer {
 public:
  Tensor weights, bias, input, output, grad_weights, grad_bias, input_deltas, output_deltas;
  bool use_bias, training;

  // Xavier-uniform weights from Tensor::global_gen, zero bias (dense_layer.hpp:18-31)
  DenseLayer(int in_features, int out_features, bool use_bias_ = true) : use_bias(use_bias_), training(true) {
    const float limit = std::sqrt(6.0f / (in_features + out_features));
    weights = Tensor::rand_uniform({in_features, out_features}, -limit, limit);
    grad_weights = Tensor::zeros({in_features, out_features});
    if (use_bias) {
      bias = Tensor::zeros({out_features});
      grad_bias = Tensor::zeros({out_features});
    }
  }

  void get_parameters(Tensor &out_weights, Tensor &out_bias) const {
    out_weights = weights;
    out_bias = use_bias ? bias : Tensor();
  }
  void set_parameters(const Tensor &new_weights, const Tensor &new_bias) {
    weights = new_weights;
    bias = (use_bias && !new_bias.data.empty()) ? new_bias : Tensor();
  }

  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    input = inputs.at(0);  // [rows, in]
    output = input.matmul(weights, use_bias ? &bias : nullptr);
    output_deltas = output;
    return {output};
  }

  void backward(const Tensor *targets = nullptr, const Layer *next_layer = nullptr) override {
    if (!targets && !next_layer) throw std::runtime_error("DenseLayer::backward: no delta source");
    const Tensor &delta = targets ? *targets : next_layer->get_input_deltas();  // [rows, out]
    grad_weights = input.matmul_tn(delta);                                       // x^T . delta
    if (use_bias) grad_bias = delta.sum(0);
    input_deltas = delta.matmul_nt(weights);                                     // delta . W^T
  }

  void update_weights(float batch_size) override { vt_host::scaled_update(optimizer, weights, grad_weights, bias, grad_bias, batch_size); }
  void zero_grad() override {
    grad_weights.fill(0.0f);
    if (use_bias) grad_bias.fill(0.0f);
  }
  void set_training(bool is_training_) override { training = is_training_; }
  const Tensor &get_output() const { return output; }
  const Tensor &get_input_deltas() const override { return input_deltas; }
  const Tensor &get_last_input() const { return input; }
  std::string get_type() const override { return "Dense"; }
  bool has_weights() const override { return true; }
};
