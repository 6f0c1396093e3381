// FlattenLayer (reference layers/flatten_layer.hpp:4-39): [N, ...] -> [N, rest]; the bytes stay in place.
#pragma once
#include "layer.hpp"

class FlattenLayer : public Layer {
  Tensor input, output, inputDeltas;

 public:
  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    input = inputs.at(0);
    output = input.reshape({input.shape.at(0), -1});
    return {output};
  }
  void backward(const Tensor * = nullptr, const Layer *next_layer = nullptr) override {
    if (!next_layer) throw std::runtime_error("FlattenLayer::backward: next_layer required");
    inputDeltas = next_layer->get_input_deltas().reshape(input.shape);
  }
  void update_weights(float) override {}
  void zero_grad() override {}
  void set_training(bool isTraining) override { is_training = isTraining; }
  const Tensor &get_input_deltas() const override { return inputDeltas; }
  bool has_weights() const override { return false; }
};
