// Optimizer interface of the reference (utils/optimizer.hpp:1-30).  On the device path the update
// itself is one fused kernel over the whole parameter arena (vt_sgd_update); these classes carry
// the hyper-parameters and keep the per-tensor host form for callers that hold host tensors.
#pragma once
#include "tensor.hpp"

class Optimizer {
 public:
  virtual ~Optimizer() = default;
  virtual void update(Tensor &weights, const Tensor &grad_weights, Tensor &biases, const Tensor &grad_biases) = 0;
  virtual float learning_rate_value() const = 0;
};

class SGD : public Optimizer {
  float learning_rate;

 public:
  explicit SGD(float lr = 0.01f) : learning_rate(lr) {}
  float learning_rate_value() const override { return learning_rate; }
  // w <- w - g*lr, b <- b - gb*lr on host tensors
  void update(Tensor &weights, const Tensor &grad_weights, Tensor &biases, const Tensor &grad_biases) override {
    for (size_t i = 0; i < weights.data.size() && i < grad_weights.data.size(); ++i)
      weights.data[i] = weights.data[i] - grad_weights.data[i] * learning_rate;
    for (size_t i = 0; i < biases.data.size() && i < grad_biases.data.size(); ++i)
      biases.data[i] = biases.data[i] - grad_biases.data[i] * learning_rate;
  }
};
