// MaxPool2DLayer (reference layers/pooling_layer.hpp:4-104): windows of kernel_size x kernel_size at the
// given stride over [N,C,H,W]; the first strictly greater element of a window wins (:49-54); backward
// writes each output delta to its window's winner (:88-92).  Index work on the host, as in the reference
// (the fused step pools inside vt_head_fwd / vt_head_bwd).
#pragma once
#include <limits>

#include "layer.hpp"

class MaxPool2DLayer : public Layer {
  int kernel_size, stride;
  Tensor input, output, input_deltas;
  std::vector<int> max_indices;  // winner of every output element as h * W + w inside its plane

 public:
  MaxPool2DLayer(int kernel_size_, int stride_) : kernel_size(kernel_size_), stride(stride_) {}

  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    input = inputs.at(0);
    if (input.shape.size() != 4) throw std::runtime_error("MaxPool2DLayer::forward: expects [N,C,H,W]");
    const int planes = input.shape[0] * input.shape[1], H = input.shape[2], W = input.shape[3];
    const int Ho = (H - kernel_size) / stride + 1, Wo = (W - kernel_size) / stride + 1;
    output = Tensor({input.shape[0], input.shape[1], Ho, Wo});
    max_indices.assign((size_t)planes * Ho * Wo, -1);
    size_t o = 0;
    for (int pl = 0; pl < planes; ++pl) {
      const float *src = input.data.data() + (size_t)pl * H * W;
      for (int i = 0; i < Ho; ++i)
        for (int j = 0; j < Wo; ++j, ++o) {
          float best = -std::numeric_limits<float>::infinity();
          int where = -1;
          for (int ki = 0; ki < kernel_size; ++ki)
            for (int kj = 0; kj < kernel_size; ++kj) {
              const int at = (i * stride + ki) * W + j * stride + kj;
              if (src[at] > best) {
                best = src[at];
                where = at;
              }
            }
          output.data[o] = best;
          max_indices[o] = where;
        }
    }
    return {output};
  }

  void backward(const Tensor * = nullptr, const Layer *next_layer = nullptr) override {
    if (!next_layer) throw std::runtime_error("MaxPool2DLayer::backward: next_layer required");
    const Tensor &delta = next_layer->get_input_deltas();
    if (delta.data.size() != max_indices.size()) throw std::runtime_error("MaxPool2DLayer::backward: delta size mismatch");
    input_deltas = Tensor::zeros(input.shape);
    const size_t plane_in = (size_t)input.shape[2] * input.shape[3], per_plane = delta.data.size() / ((size_t)input.shape[0] * input.shape[1]);
    for (size_t o = 0; o < delta.data.size(); ++o)
      if (max_indices[o] >= 0) input_deltas.data[(o / per_plane) * plane_in + max_indices[o]] = delta.data[o];
  }
  void update_weights(float) override {}
  void zero_grad() override {}
  const Tensor &get_input_deltas() const override { return input_deltas; }
  void set_training(bool is_train) override { is_training = is_train; }
  bool has_weights() const override { return false; }
};
