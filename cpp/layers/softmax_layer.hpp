// SoftmaxLayer (reference layers/softmax_layer.hpp:5-60): softmax over the class axis; with targets the
// delta is softmax - targets (no 1/N, :29), otherwise Tensor::softmax_backward of the next layer's delta.
#pragma once
#include "layer.hpp"

class SoftmaxLayer : public Layer {
  Tensor input, output, input_deltas;
  bool training = true;

 public:
  SoftmaxLayer() {}
  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    input = inputs.at(0);  // [batch, classes]
    output = input.softmax(1);
    return {output};
  }
  void backward(const Tensor *targets = nullptr, const Layer *next_layer = nullptr) override {
    if (targets) input_deltas = output - *targets;
    else if (next_layer) input_deltas = Tensor::softmax_backward(output, next_layer->get_input_deltas());
    else throw std::runtime_error("SoftmaxLayer::backward: targets or next_layer must be provided.");
  }
  void update_weights(float) override {}
  void zero_grad() override {}
  void set_training(bool is_training_) override { training = is_training_; }
  const Tensor &get_input_deltas() const override { return input_deltas; }
  std::string get_type() const override { return "Softmax"; }
  bool has_weights() const override { return false; }
};
