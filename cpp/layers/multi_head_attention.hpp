// MultiHeadAttention behind the reference's surface (layers/multi_head_attention.hpp:4-314).
// Per head: query / key / value DenseLayers without bias (channels -> head_dim), scores = key.query^T / sqrt(d)
// (key times query-transposed, :89-91), softmax over the last axis, head = attn.value; the heads are
// concatenated and projected by a bias-free output DenseLayer.  The reference's Tensor::concat returns its
// result unfilled (utils/tensor/tensor_cpu.hpp:218-256), so the layer's forward output is exactly zero and
// the output projection never receives a gradient; reference_compat (default true) reproduces that, false
// concatenates for real.  backward (:109-193) recomputes the per-head forward and omits the 1/sqrt(d)
// factor in the score delta (:165-170).  Every product is a device GEMM (Tensor::gemm -> vt_gemm_f32).
#pragma once
#include "dense_layer.hpp"

class MultiHeadAttention : public Layer {
  int num_heads, channels, head_dim;
  std::vector<DenseLayer *> query_linears, key_linears, value_linears;
  DenseLayer *output_linear;
  Tensor last_input, input_deltas, delta_output_flat;
  std::vector<Tensor> delta_query_flat, delta_key_flat, delta_value_flat;
  bool reference_compat_ = true;

  struct HeadForward {
    Tensor query, key, value, attn;
  };
  HeadForward run_head(int h, const Tensor &x_rows, int B, int T) {
    HeadForward f;
    f.query = query_linears[h]->forward({x_rows})[0].reshape({B, T, head_dim});
    f.key = key_linears[h]->forward({x_rows})[0].reshape({B, T, head_dim});
    f.value = value_linears[h]->forward({x_rows})[0].reshape({B, T, head_dim});
    f.attn = (f.key.matmul_nt(f.query) * (1.0f / std::sqrt(static_cast<float>(head_dim)))).softmax(2);  // [B, T, T]
    return f;
  }

 public:
  MultiHeadAttention(int channels_, int attn_dim, int num_heads_)
      : num_heads(num_heads_), channels(channels_), head_dim(num_heads_ > 0 ? attn_dim / num_heads_ : 0) {
    if (num_heads < 1 || attn_dim % num_heads != 0) throw std::runtime_error("attn_dim debe ser divisible por num_heads");
    for (int h = 0; h < num_heads; ++h) {  // draw order: per head query, key, value; then the output layer (:33-46)
      query_linears.push_back(new DenseLayer(channels, head_dim, false));
      key_linears.push_back(new DenseLayer(channels, head_dim, false));
      value_linears.push_back(new DenseLayer(channels, head_dim, false));
      for (DenseLayer *l : {query_linears[h], key_linears[h], value_linears[h]}) l->set_optimizer(new SGD(0.001f));
    }
    output_linear = new DenseLayer(attn_dim, channels, false);
    output_linear->set_optimizer(new SGD(0.001f));
    delta_query_flat.resize(num_heads);
    delta_key_flat.resize(num_heads);
    delta_value_flat.resize(num_heads);
  }
  MultiHeadAttention(const MultiHeadAttention &) = delete;
  MultiHeadAttention &operator=(const MultiHeadAttention &) = delete;
  ~MultiHeadAttention() override {
    for (int h = 0; h < num_heads; ++h) {
      delete query_linears[h];
      delete key_linears[h];
      delete value_linears[h];
    }
    delete output_linear;
  }
  void set_reference_compat(bool on) { reference_compat_ = on; }

  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    last_input = inputs.at(0);  // [B, T, D]
    const int B = last_input.shape.at(0), T = last_input.shape.at(1), D = last_input.shape.at(2);
    const Tensor x_rows = last_input.reshape({B * T, D});
    std::vector<Tensor> heads(num_heads);
    for (int h = 0; h < num_heads; ++h) {
      const HeadForward f = run_head(h, x_rows, B, T);
      heads[h] = f.attn.matmul(f.value);  // [B, T, head_dim]
    }
    const Tensor joined = Tensor::concat(heads, 2, reference_compat_);  // zeros in compat mode
    return {output_linear->forward({joined.reshape({B * T, head_dim * num_heads})})[0].reshape({B, T, D})};
  }

  void backward(const Tensor *targets = nullptr, const Layer *next_layer = nullptr) override {
    if (!targets && !next_layer) throw std::runtime_error("MultiHeadAttention: No hay fuente de delta para backward.");
    const Tensor &delta = targets ? *targets : next_layer->get_input_deltas();  // [B, T, D]
    const int B = delta.shape.at(0), T = delta.shape.at(1), D = delta.shape.at(2);
    delta_output_flat = delta.reshape({B * T, D});
    output_linear->backward(&delta_output_flat);  // its cached input is the (zero) concat: zero weight gradient in compat mode
    const Tensor d_concat = output_linear->get_input_deltas().reshape({B, T, head_dim * num_heads});
    const Tensor x_rows = last_input.reshape({B * T, D});
    input_deltas = Tensor::zeros({B, T, D});
    for (int h = 0; h < num_heads; ++h) {
      const Tensor d_head = d_concat.slice(2, h * head_dim, (h + 1) * head_dim);  // [B, T, head_dim]
      const HeadForward f = run_head(h, x_rows, B, T);
      const Tensor d_attn = d_head.matmul_nt(f.value);                  // [B, T, T]
      const Tensor d_value = f.attn.matmul_tn(d_head);                  // attn^T . dhead
      const Tensor d_scores = Tensor::softmax_backward(f.attn, d_attn);  // no 1/sqrt(d) here, as in the reference
      delta_key_flat[h] = d_scores.matmul(f.query).reshape({B * T, head_dim});
      delta_query_flat[h] = d_scores.matmul_tn(f.key).reshape({B * T, head_dim});  // (key^T . dscores)^T
      delta_value_flat[h] = d_value.reshape({B * T, head_dim});
      key_linears[h]->backward(&delta_key_flat[h]);
      query_linears[h]->backward(&delta_query_flat[h]);
      value_linears[h]->backward(&delta_value_flat[h]);
      for (DenseLayer *l : {query_linears[h], key_linears[h], value_linears[h]})
        input_deltas = input_deltas + l->get_input_deltas().reshape({B, T, D});
    }
  }

  void update_weights(float batch_size) override {
    for (int h = 0; h < num_heads; ++h)
      for (DenseLayer *l : {query_linears[h], key_linears[h], value_linears[h]}) l->update_weights(batch_size);
    output_linear->update_weights(batch_size);
  }
  void zero_grad() override {
    for (int h = 0; h < num_heads; ++h) {
      delta_query_flat[h].fill(0.0f);
      delta_key_flat[h].fill(0.0f);
      delta_value_flat[h].fill(0.0f);
      for (DenseLayer *l : {query_linears[h], key_linears[h], value_linears[h]}) l->zero_grad();
    }
    output_linear->zero_grad();
  }
  void set_training(bool training) override {
    for (int h = 0; h < num_heads; ++h)
      for (DenseLayer *l : {query_linears[h], key_linears[h], value_linears[h]}) l->set_training(training);
    output_linear->set_training(training);
  }
  const Tensor &get_input_deltas() const override { return input_deltas; }
  std::string get_type() const override { return "MultiHeadAttention"; }

  // per head query, key, value weights, then the output projection (:238-272)
  std::vector<Tensor> get_parameters() const {
    std::vector<Tensor> p;
    for (int h = 0; h < num_heads; ++h)
      for (const DenseLayer *l : {query_linears[h], key_linears[h], value_linears[h]}) p.push_back(l->weights);
    p.push_back(output_linear->weights);
    return p;
  }
  void set_parameters(const std::vector<Tensor> &p) {
    if ((int)p.size() != 3 * num_heads + 1) throw std::runtime_error("MultiHeadAttention: wrong number of parameter tensors");
    const Tensor none;
    for (int h = 0; h < num_heads; ++h) {
      query_linears[h]->set_parameters(p[3 * h], none);
      key_linears[h]->set_parameters(p[3 * h + 1], none);
      value_linears[h]->set_parameters(p[3 * h + 2], none);
    }
    output_linear->set_parameters(p[3 * num_heads], none);
  }
  // gradient access for tests / tools: which = 0 query, 1 key, 2 value of head h; h = -1: output projection
  DenseLayer *dense(int h, int which) { return h < 0 ? output_linear : which == 0 ? query_linears[h] : which == 1 ? key_linears[h] : value_linears[h]; }
};
