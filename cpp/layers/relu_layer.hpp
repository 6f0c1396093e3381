// ReLULayer (reference layers/relu_layer.hpp:4-43).  backward multiplies the incoming delta with the
// derivative RESHAPED to the delta's shape: a flat, element-by-element product whatever the two layouts mean
// (this is how the projector's [N,P,C] activations meet the pooling layer's [N,C,H,W] delta).
#pragma once
#include "layer.hpp"

class ReLULayer : public Layer {
  Tensor input, input_deltas, output;
  bool training = true;

 public:
  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    input = inputs.at(0);
    output = input.relu();
    return {output};
  }
  void backward(const Tensor * = nullptr, const Layer *next_layer = nullptr) override {
    if (!next_layer) throw std::runtime_error("ReLULayer::backward: next_layer required");
    const Tensor &delta = next_layer->get_input_deltas();
    if (delta.data.size() != input.data.size()) throw std::runtime_error("ReLULayer::backward: delta size mismatch");
    input_deltas = delta;
    for (size_t i = 0; i < input.data.size(); ++i) input_deltas.data[i] = delta.data[i] * (input.data[i] > 0 ? 1.0f : 0.0f);
  }
  void update_weights(float) override {}
  void zero_grad() override {}
  std::string get_type() const override { return "ReLU"; }
  void set_training(bool is_training_) override { training = is_training_; }
  const Tensor &get_input_deltas() const override { return input_deltas; }
};
