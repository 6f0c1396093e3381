// ProjectorLayer behind the reference's surface (layers/projector_layer.hpp:4-392): tokens are projected
// back onto the feature map.  x [N,P,Cin], t [N,L,D]:
//   q = x.W1 (no bias) [N,P,D];  k = t.W2 (no bias) [N,L,D];  a = softmax over tokens of q.k^T [N,P,L]
//   v = t.W3 + b3 [N,L,Cout];  z = (x.Wd + bd) + a.v;  out = relu(BatchNorm1D(z as [N*P, Cout]))   (:43-116)
// backward (:118-214) takes the next layer's delta through the ReLU as a flat product (the pooling layer
// hands over [N,Cout,H,W]; the activations are [N,P,Cout]: same bytes, no transpose), recomputes q, k, v
// from the cached inputs, and produces the input delta and the token delta (get_token_deltas).
// As in the reference the layer needs in_channels != out_channels (the downsample layer is used unconditionally).
#pragma once
#include "batch_normalization_layer.hpp"
#include "dense_layer.hpp"
#include "relu_layer.hpp"

class ProjectorLayer : public Layer {
  int in_channels, out_channels, token_channels;
  DenseLayer *linear1, *linear2, *linear3, *downsample;
  BatchNormLayer1D *norm;
  ReLULayer *relu;
  Tensor cache, input_deltas, x_cache, token_cache, token_deltas;

 public:
  ProjectorLayer(int in_channels_, int out_channels_, int token_channels_)
      : in_channels(in_channels_), out_channels(out_channels_), token_channels(token_channels_) {
    if (in_channels == out_channels) throw std::runtime_error("ProjectorLayer: in_channels must differ from out_channels");
    linear1 = new DenseLayer(in_channels, token_channels, false);
    linear2 = new DenseLayer(token_channels, token_channels, false);
    linear3 = new DenseLayer(token_channels, out_channels);
    norm = new BatchNormLayer1D(out_channels);
    relu = new ReLULayer();
    downsample = new DenseLayer(in_channels, out_channels);
    for (Layer *l : std::vector<Layer *>{linear1, linear2, linear3, norm, downsample}) l->set_optimizer(new SGD(0.001f));
  }
  ProjectorLayer(const ProjectorLayer &) = delete;
  ProjectorLayer &operator=(const ProjectorLayer &) = delete;
  ~ProjectorLayer() override {
    delete linear1;
    delete linear2;
    delete linear3;
    delete norm;
    delete relu;
    delete downsample;
  }

  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    x_cache = inputs.at(0);      // [N, P, Cin]
    token_cache = inputs.at(1);  // [N, L, D]
    const int N = x_cache.shape.at(0), P = x_cache.shape.at(1), L = token_cache.shape.at(1);
    const Tensor x_rows = x_cache.reshape({N * P, in_channels}), t_rows = token_cache.reshape({N * L, token_channels});
    const Tensor q = linear1->forward({x_rows})[0].reshape({N, P, token_channels});
    const Tensor k = linear2->forward({t_rows})[0].reshape({N, L, token_channels});
    cache = q.matmul_nt(k).softmax(2);  // [N, P, L]
    const Tensor v = linear3->forward({t_rows})[0].reshape({N, L, out_channels});
    const Tensor z = downsample->forward({x_rows})[0].reshape({N, P, out_channels}) + cache.matmul(v);
    const Tensor y = norm->forward({z.reshape({N * P, out_channels})})[0].reshape({N, P, out_channels});
    return {relu->forward({y})[0]};
  }

  void backward(const Tensor * = nullptr, const Layer *next_layer = nullptr) override {
    relu->backward(nullptr, next_layer);  // flat product with the derivative, whatever the delta's shape
    const int N = x_cache.shape[0], P = x_cache.shape[1], L = token_cache.shape[1];
    const Tensor du = relu->get_input_deltas().reshape({N * P, out_channels});
    norm->backward(&du, nullptr);
    const Tensor dz_rows = norm->get_input_deltas();  // [N*P, Cout]
    const Tensor dz = dz_rows.reshape({N, P, out_channels});
    downsample->backward(&dz_rows);
    input_deltas = downsample->get_input_deltas().reshape({N, P, in_channels});
    // the three dense layers' outputs are recomputed from the cached inputs (:161, :174-177)
    const Tensor x_rows = x_cache.reshape({N * P, in_channels}), t_rows = token_cache.reshape({N * L, token_channels});
    const Tensor v = linear3->forward({t_rows})[0].reshape({N, L, out_channels});
    const Tensor d_a = dz.matmul_nt(v);          // [N, P, L]
    const Tensor d_v = cache.matmul_tn(dz);      // [N, L, Cout]
    const Tensor d_s = Tensor::softmax_backward(cache, d_a);
    const Tensor q = linear1->forward({x_rows})[0].reshape({N, P, token_channels});
    const Tensor k = linear2->forward({t_rows})[0].reshape({N, L, token_channels});
    const Tensor d_q = d_s.matmul(k);            // [N, P, D]
    const Tensor d_k = d_s.matmul_tn(q);         // (q^T . ds)^T  [N, L, D]
    const Tensor d_v_rows = d_v.reshape({N * L, out_channels}), d_k_rows = d_k.reshape({N * L, token_channels});
    linear3->backward(&d_v_rows);
    linear2->backward(&d_k_rows);
    token_deltas = (linear3->get_input_deltas() + linear2->get_input_deltas()).reshape({N, L, token_channels});
    const Tensor d_q_rows = d_q.reshape({N * P, token_channels});
    linear1->backward(&d_q_rows);
    input_deltas = input_deltas + linear1->get_input_deltas().reshape({N, P, in_channels});
  }

  void update_weights(float batch_size) override {
    for (Layer *l : std::vector<Layer *>{linear1, linear2, linear3, norm, downsample}) l->update_weights(batch_size);
  }
  void zero_grad() override {
    for (Layer *l : std::vector<Layer *>{linear1, linear2, linear3, norm, downsample}) l->zero_grad();
  }
  void set_training(bool training) override {
    is_training = training;
    for (Layer *l : std::vector<Layer *>{linear1, linear2, linear3, norm, downsample}) l->set_training(training);
  }
  const Tensor &get_input_deltas() const override { return input_deltas; }
  Tensor get_token_deltas() const { return token_deltas; }
  bool has_weights() const override { return true; }
  std::string get_type() const override { return "ProjectorLayer"; }

  // linear1.W, linear2.W, linear3.W, linear3.b, norm.gamma, norm.beta, downsample.W, downsample.b (:263-296)
  std::vector<Tensor> get_parameters() const {
    Tensor g, b;
    norm->get_parameters(g, b);
    return {linear1->weights, linear2->weights, linear3->weights, linear3->bias, g, b, downsample->weights, downsample->bias};
  }
  void set_parameters(const std::vector<Tensor> &p) {
    if (p.size() != 8) throw std::runtime_error("ProjectorLayer: Expected 8 parameter tensors, got " + std::to_string(p.size()));
    const Tensor none;
    linear1->set_parameters(p[0], none);
    linear2->set_parameters(p[1], none);
    linear3->set_parameters(p[2], p[3]);
    norm->set_parameters(p[4], p[5]);
    downsample->set_parameters(p[6], p[7]);
  }
  DenseLayer *dense(int which) { return which == 0 ? linear1 : which == 1 ? linear2 : which == 2 ? linear3 : downsample; }
  BatchNormLayer1D *batch_norm() { return norm; }
};
