// DropoutLayer (reference layers/dropout_layer.hpp:4-52): inverted dropout; the mask is drawn element by
// element from Tensor::global_gen (Tensor::bernoulli), so the draw order is the reference's.  Used by MLP only.
#pragma once
#include "layer.hpp"

class DropoutLayer : public Layer {
  float dropout_prob;
  Tensor mask, input_deltas;

 public:
  explicit DropoutLayer(float prob = 0.5f) : dropout_prob(prob) {}
  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    const Tensor &x = inputs.at(0);
    if (!is_training) return {x};
    mask = Tensor::bernoulli(x.shape, 1.0f - dropout_prob);  // 1 = kept
    return {(x * mask) / (1.0f - dropout_prob)};
  }
  void backward(const Tensor * = nullptr, const Layer *next_layer = nullptr) override {
    if (!next_layer) throw std::runtime_error("DropoutLayer::backward: next_layer required");
    input_deltas = next_layer->get_input_deltas() * mask;
  }
  void update_weights(float) override {}
  void zero_grad() override {}
  void set_training(bool training) override { is_training = training; }
  const Tensor &get_input_deltas() const override { return input_deltas; }
  std::string get_type() const override { return "Dropout"; }
};
