// TransformerLayer behind the reference's surface (layers/transformer_layer.hpp:4-343): one encoder block on
// a token tensor labelled [L, N, C]:  r = x + attention(x);  a = BatchNorm1D(r as [L*N, C]);
// b = linear2(relu(linear1(a)));  out = BatchNorm1D(a + b)   (:43-131).  The feed-forward block is C -> C -> C
// whatever fc dimension the caller asked for, and dropout is ignored, as in the reference (:18-23).
// backward (:133-207): with targets the seed is  out - targets  (:142-145); no ReLU derivative between the two
// dense layers (:174-181); the resulting input delta is kept in last_output, which is what get_input_deltas
// returns.  set_training does nothing (:224): both norms always use batch statistics.
#pragma once
#include "batch_normalization_layer.hpp"
#include "multi_head_attention.hpp"

class TransformerLayer : public Layer {
  int token_channels, attn_dim, num_heads;
  MultiHeadAttention *attention;
  DenseLayer *linear1, *linear2;
  BatchNormLayer1D *norm1, *norm2;
  Tensor last_output;

 public:
  TransformerLayer(int token_channels_, int attn_dim_, int num_heads_, float /*dropout*/ = 0.5f)
      : token_channels(token_channels_), attn_dim(attn_dim_), num_heads(num_heads_) {
    attention = new MultiHeadAttention(token_channels, attn_dim, num_heads);
    linear1 = new DenseLayer(token_channels, token_channels);
    linear2 = new DenseLayer(token_channels, token_channels);
    norm1 = new BatchNormLayer1D(token_channels);
    norm2 = new BatchNormLayer1D(token_channels);
    for (Layer *l : std::vector<Layer *>{attention, linear1, linear2, norm1, norm2}) l->set_optimizer(new SGD(0.001f));
  }
  TransformerLayer(const TransformerLayer &) = delete;
  TransformerLayer &operator=(const TransformerLayer &) = delete;
  ~TransformerLayer() override {
    delete attention;
    delete linear1;
    delete linear2;
    delete norm1;
    delete norm2;
  }

  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    const Tensor &x = inputs.at(0);  // [L, N, C]
    const int L = x.shape.at(0), N = x.shape.at(1), C = x.shape.at(2);
    const Tensor residual = x + attention->forward({x})[0];
    const Tensor a = norm1->forward({residual.reshape({L * N, C})})[0];  // [L*N, C]
    const Tensor hidden = linear1->forward({a})[0].relu();
    const Tensor b = linear2->forward({hidden})[0];
    last_output = norm2->forward({a + b})[0].reshape({L, N, C});
    return {last_output};
  }

  void backward(const Tensor *targets = nullptr, const Layer *next_layer = nullptr) override {
    Tensor delta;
    if (next_layer) delta = next_layer->get_input_deltas();
    else if (targets) delta = last_output - *targets;
    else throw std::runtime_error("TransformerLayer backward: either next_layer or targets must be provided");
    const int L = last_output.shape[0], N = last_output.shape[1], C = last_output.shape[2];
    delta = delta.reshape({L * N, C});
    norm2->backward(&delta, nullptr);
    const Tensor d_sum = norm2->get_input_deltas();  // delta of a + b: goes to both branches
    linear2->backward(&d_sum, nullptr);
    const Tensor d_hidden = linear2->get_input_deltas();
    linear1->backward(&d_hidden, nullptr);  // straight through: the ReLU between the two is not differentiated
    const Tensor d_a = linear1->get_input_deltas() + d_sum;
    norm1->backward(&d_a, nullptr);
    const Tensor d_residual = norm1->get_input_deltas().reshape({L, N, C});
    attention->backward(&d_residual, nullptr);
    last_output = d_residual + attention->get_input_deltas();
  }

  void update_weights(float batch_size) override {
    for (Layer *l : std::vector<Layer *>{attention, linear1, linear2, norm1, norm2}) l->update_weights(batch_size);
  }
  void zero_grad() override {
    for (Layer *l : std::vector<Layer *>{attention, linear1, linear2, norm1, norm2}) l->zero_grad();
  }
  void set_training(bool) override {}
  const Tensor &get_input_deltas() const override { return last_output; }
  std::string get_type() const override { return "TransformerLayer"; }

  // attention parameters, linear1 W b, linear2 W b, norm1 gamma beta, norm2 gamma beta (:241-264).  Note
  // has_weights() is NOT overridden (false): a VisualTransformer therefore leaves this block out of its own
  // parameter list and out of checkpoints (layers/layer.hpp:21, visual_transformer.hpp:256-260).
  std::vector<Tensor> get_parameters() const {
    std::vector<Tensor> p = attention->get_parameters();
    Tensor g1, b1, g2, b2;
    norm1->get_parameters(g1, b1);
    norm2->get_parameters(g2, b2);
    for (const Tensor &t : {linear1->weights, linear1->bias, linear2->weights, linear2->bias, g1, b1, g2, b2}) p.push_back(t);
    return p;
  }
  void set_parameters(const std::vector<Tensor> &p) {
    const size_t na = attention->get_parameters().size();
    if (p.size() != na + 8) throw std::runtime_error("TransformerLayer: wrong number of parameter tensors");
    attention->set_parameters(std::vector<Tensor>(p.begin(), p.begin() + na));
    linear1->set_parameters(p[na], p[na + 1]);
    linear2->set_parameters(p[na + 2], p[na + 3]);
    norm1->set_parameters(p[na + 4], p[na + 5]);
    norm2->set_parameters(p[na + 6], p[na + 7]);
  }
  MultiHeadAttention *attention_layer() { return attention; }
  DenseLayer *dense(int which) { return which == 0 ? linear1 : linear2; }
  BatchNormLayer1D *batch_norm(int which) { return which == 0 ? norm1 : norm2; }
};
