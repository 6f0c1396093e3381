// BatchNormLayer1D and BatchNorm2DLayer behind the reference's surface
// (layers/batch_normalization_layer.hpp:3-234, 236-506).  Both are the same device arithmetic
// (vt_batchnorm_fwd / vt_batchnorm_bwd) on a row matrix [rows, channels]: the 1-D layer's input already is
// one; the 2-D layer's [N,C,H,W] input is viewed as [N*H*W, C] (one row per pixel), so that "row 0" of the
// backward pass's variance term (:197-199, :324-326) is the element [0, c, 0, 0] the reference reads.
// Training mode: batch statistics, biased variance, running = 0.1 batch + 0.9 running; eval: running buffers.
#pragma once
#include <iostream>

#include "dense_layer.hpp"

namespace vt_host {
class BatchNormCore {
 protected:
  Tensor gamma, beta, running_mean, running_var, grad_gamma, grad_beta;
  Tensor rows_in_, stat_;  // the row-matrix input of the last forward and its (mean, 1/std) record
  float momentum, eps;
  int channels_;

  BatchNormCore(int channels, float eps_, float momentum_) : momentum(momentum_), eps(eps_), channels_(channels) {
    gamma = Tensor({channels});
    gamma.fill(1.0f);
    beta = Tensor::zeros({channels});
    running_mean = Tensor::zeros({channels});
    running_var = Tensor({channels});
    running_var.fill(1.0f);
    grad_gamma = Tensor::zeros({channels});
    grad_beta = Tensor::zeros({channels});
    stat_ = Tensor({2 * channels});
    if (eps_ != 1e-5f || momentum_ != 0.1f) throw std::runtime_error("BatchNorm: the device kernels are built for eps 1e-5, momentum 0.1");
  }
  static void ok(int rc, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + vt_error_string(rc));
  }
  Tensor forward_rows(const Tensor &rows, bool training) {  // [R, C] -> [R, C]
    rows_in_ = rows;
    const long R = rows.shape.at(0);
    Tensor out(rows.shape);
    DeviceBuffer scratch((size_t)vt_batchnorm_scratch_floats(channels_));
    ok(vt_batchnorm_fwd(rows_in_.to_device(), R, channels_, gamma.to_device(), beta.to_device(), running_mean.to_device(),
                        running_var.to_device(), training ? 1 : 0, stat_.device_buffer(), out.device_buffer(), scratch.ptr,
                        (long)scratch.count, nullptr),
       "BatchNorm forward");
    out.from_device();
    if (training) {
      running_mean.from_device();
      running_var.from_device();
    }
    stat_.from_device();
    return out;
  }
  Tensor backward_rows(const Tensor &delta_rows) {  // [R, C] -> input delta [R, C]; fills grad_gamma / grad_beta
    if (delta_rows.data.size() != rows_in_.data.size()) throw std::runtime_error("BatchNorm backward: delta size mismatch");
    Tensor din(rows_in_.shape);
    DeviceBuffer scratch((size_t)vt_batchnorm_scratch_floats(channels_));
    ok(vt_batchnorm_bwd(delta_rows.to_device(), rows_in_.to_device(), stat_.to_device(), gamma.to_device(), rows_in_.shape[0], channels_,
                        grad_gamma.device_buffer(), grad_beta.device_buffer(), din.device_buffer(), scratch.ptr, (long)scratch.count,
                        nullptr),
       "BatchNorm backward");
    grad_gamma.from_device();
    grad_beta.from_device();
    din.from_device();
    return din;
  }
  // update_weights: both gradients scaled, then two optimizer calls with empty biases (:202-212, :329-339)
  void update(Optimizer *opt, float batch_size) {
    if (!opt) return;
    Tensor none;
    scaled_update(opt, gamma, grad_gamma, none, none, batch_size);
    scaled_update(opt, beta, grad_beta, none, none, batch_size);
  }
  // four tensor records: gamma, beta, running mean, running variance (:52-84, :384-432)
  void save_records(std::ostream &out) const {
    gamma.serialize(out);
    beta.serialize(out);
    running_mean.serialize(out);
    running_var.serialize(out);
  }
  void load_records(std::istream &in, const char *who) {
    Tensor t[4];
    for (Tensor &x : t) {
      x.deserialize(in);
      if (x.shape.empty() || x.shape[0] != channels_) throw std::runtime_error(std::string(who) + ": Shape mismatch during load");
    }
    gamma = t[0];
    beta = t[1];
    running_mean = t[2];
    running_var = t[3];
    grad_gamma = Tensor::zeros({channels_});
    grad_beta = Tensor::zeros({channels_});
  }
};
}  // namespace vt_host

class BatchNormLayer1D : public Layer, protected vt_host::BatchNormCore {
  Tensor input_deltas;

 public:
  explicit BatchNormLayer1D(int features, float eps_ = 1e-5f, float momentum_ = 0.1f) : BatchNormCore(features, eps_, momentum_) {}
  std::string get_type() const override { return "BatchNorm1D"; }
  void get_parameters(Tensor &out_gamma, Tensor &out_beta) const {
    out_gamma = gamma;
    out_beta = beta;
  }
  void set_parameters(const Tensor &new_gamma, const Tensor &new_beta) {  // running statistics are left alone (:41-50)
    gamma = new_gamma;
    beta = new_beta;
    grad_gamma = Tensor::zeros({channels_});
    grad_beta = Tensor::zeros({channels_});
  }
  void save(std::ostream &out) const { save_records(out); }
  void load(std::istream &in) { load_records(in, "BatchNormLayer1D"); }
  const Tensor &get_running_mean() const { return running_mean; }
  const Tensor &get_running_var() const { return running_var; }
  const Tensor &get_grad_gamma() const { return grad_gamma; }
  const Tensor &get_grad_beta() const { return grad_beta; }

  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    const Tensor &x = inputs.at(0);  // [rows, C]
    if (x.shape.size() != 2 || x.shape[1] != channels_) throw std::runtime_error("BatchNormLayer1D::forward: expects [rows, features]");
    if (!is_training) std::cout << "BatchNorm1D: Usando estadísticas acumuladas" << std::endl;  // as the reference prints (:163)
    return {forward_rows(x, is_training)};
  }
  void backward(const Tensor *targets, const Layer *next_layer) override {
    if (!targets && !next_layer) throw std::runtime_error("BatchNormLayer1D::backward: no delta source");
    input_deltas = backward_rows(targets ? *targets : next_layer->get_input_deltas());
  }
  void update_weights(float batch_size) override { update(optimizer, batch_size); }
  void zero_grad() override {
    grad_gamma.fill(0.0f);
    grad_beta.fill(0.0f);
  }
  void set_training(bool train) override { is_training = train; }
  const Tensor &get_input_deltas() const override { return input_deltas; }
  bool has_weights() const override { return true; }
};

class BatchNorm2DLayer : public Layer, protected vt_host::BatchNormCore {
  Tensor input_deltas;
  std::vector<int> in_shape_;

  // [N,C,H,W] <-> one row per pixel
  static Tensor to_rows(const Tensor &x) {
    const int N = x.shape[0], C = x.shape[1], HW = x.shape[2] * x.shape[3];
    return x.reshape({N, C, HW}).transpose({0, 2, 1}).reshape({N * HW, C});
  }
  Tensor from_rows(const Tensor &rows) const {
    const int N = in_shape_[0], C = in_shape_[1], HW = in_shape_[2] * in_shape_[3];
    return rows.reshape({N, HW, C}).transpose({0, 2, 1}).reshape(in_shape_);
  }

 public:
  explicit BatchNorm2DLayer(int channels, float eps_ = 1e-5f, float momentum_ = 0.1f) : BatchNormCore(channels, eps_, momentum_) {}
  std::string get_type() const override { return "BatchNorm2D"; }
  void get_parameters(Tensor &out_gamma, Tensor &out_beta) const {
    out_gamma = gamma;
    out_beta = beta;
  }
  void set_parameters(const Tensor &new_gamma, const Tensor &new_beta) {
    gamma = new_gamma;
    beta = new_beta;
    grad_gamma = Tensor::zeros({channels_});
    grad_beta = Tensor::zeros({channels_});
  }
  void save(std::ostream &out) const { save_records(out); }
  void load(std::istream &in) { load_records(in, "BatchNorm2DLayer"); }
  const Tensor &get_running_mean() const { return running_mean; }
  const Tensor &get_running_var() const { return running_var; }
  const Tensor &get_grad_gamma() const { return grad_gamma; }
  const Tensor &get_grad_beta() const { return grad_beta; }

  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    const Tensor &x = inputs.at(0);
    if (x.shape.size() != 4 || x.shape[1] != channels_) throw std::runtime_error("BatchNorm2DLayer::forward: expects [N,C,H,W]");
    in_shape_ = x.shape;
    return {from_rows(forward_rows(to_rows(x), is_training))};
  }
  void backward(const Tensor *targets, const Layer *next_layer) override {
    if (!targets && !next_layer) throw std::runtime_error("BatchNorm2DLayer::backward: no delta source");
    const Tensor &delta = targets ? *targets : next_layer->get_input_deltas();
    if (delta.data.size() != rows_in_.data.size()) throw std::runtime_error("BatchNorm2DLayer::backward: delta size mismatch");
    input_deltas = from_rows(backward_rows(to_rows(delta.reshape(in_shape_))));
  }
  void update_weights(float batch_size) override { update(optimizer, batch_size); }
  void zero_grad() override {
    grad_gamma.fill(0.0f);
    grad_beta.fill(0.0f);
  }
  void set_training(bool train) override { is_training = train; }
  const Tensor &get_input_deltas() const override { return input_deltas; }
  bool has_weights() const override { return true; }
};
