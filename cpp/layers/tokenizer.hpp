// FilterTokenizer behind the reference's surface (layers/tokenizer.hpp:5-282): tokens are softmax-weighted
// pools of the pixels.  x [N,P,C]:  S = x.W1 + b1 [N,P,L];  A = softmax over the PIXEL axis;  T0 = A^T.x
// [N,L,C];  tokens = T0.W2 + b2 [N,L,D] (:22-48).  backward2(delta_tokens, delta_x) is the form the visual
// transformer calls (:50-107): through linear2, through the pooling product, through the softmax with the
// reference's last-axis softmax_backward (it reduces over the TOKEN axis although A was normalised over
// pixels, :91), through linear1, plus the delta arriving directly on x.  The two DenseLayers get their
// gradients but, as in the reference, no optimizer of their own: update_weights leaves them unchanged.
// RecurrentTokenizer is not provided (the reference cannot construct it either, visual_transformer.hpp:35).
#pragma once
#include "dense_layer.hpp"
#include "softmax_layer.hpp"

class FilterTokenizer : public Layer {
  int tokens, in_channels, token_channels;
  DenseLayer *linear1, *linear2;
  Tensor cache1, cache2, token_cache, input_deltas, input_cache;

 public:
  FilterTokenizer(int in_channels_, int token_channels_, int tokens_)
      : tokens(tokens_), in_channels(in_channels_), token_channels(token_channels_) {
    linear1 = new DenseLayer(in_channels, tokens);
    linear2 = new DenseLayer(in_channels, token_channels);
  }
  FilterTokenizer(const FilterTokenizer &) = delete;
  FilterTokenizer &operator=(const FilterTokenizer &) = delete;
  ~FilterTokenizer() override {
    delete linear1;
    delete linear2;
  }

  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    input_cache = inputs.at(0);  // [N, P, C]
    const int N = input_cache.shape.at(0), P = input_cache.shape.at(1), C = input_cache.shape.at(2);
    cache1 = linear1->forward({input_cache.reshape({N * P, C})})[0].reshape({N, P, tokens});
    cache2 = cache1.softmax(1);                                   // over the pixels of each (image, token)
    const Tensor pooled = cache2.matmul_tn(input_cache);          // A^T . x  [N, L, C]
    token_cache = linear2->forward({pooled.reshape({N * tokens, C})})[0].reshape({N, tokens, token_channels});
    return {token_cache};
  }

  void backward2(const Tensor &delta_tokens, const Tensor &delta_x) {
    const int N = input_cache.shape[0], P = input_cache.shape[1], C = input_cache.shape[2];
    input_deltas = Tensor::zeros({N, P, C});
    if (!delta_tokens.empty()) {
      const Tensor dt_flat = delta_tokens.reshape({N * tokens, token_channels});
      linear2->backward(&dt_flat);
      const Tensor d_pooled = linear2->get_input_deltas().reshape({N, tokens, C});  // [N, L, C]
      const Tensor d_A = input_cache.matmul_nt(d_pooled);                           // (dT0 . x^T)^T  [N, P, L]
      input_deltas = input_deltas + cache2.matmul(d_pooled);                        // A . dT0        [N, P, C]
      const Tensor d_S = Tensor::softmax_backward(cache2, d_A).reshape({N * P, tokens});
      linear1->backward(&d_S);
      input_deltas = input_deltas + linear1->get_input_deltas().reshape({N, P, C});
    }
    if (!delta_x.empty()) input_deltas = input_deltas + delta_x.reshape({N, P, C});
  }

  // the single-delta form (:109-183): the delta is taken as a delta on x
  void backward(const Tensor *targets, const Layer *next_layer) override {
    if (!targets && !next_layer) throw std::runtime_error("FilterTokenizer: No gradient source");
    const Tensor &dx = targets ? *targets : next_layer->get_input_deltas();  // [N, P, C]
    const int N = input_cache.shape[0], P = input_cache.shape[1], C = input_cache.shape[2];
    const Tensor d_pooled = cache2.matmul_tn(dx).reshape({N * tokens, C});
    linear2->backward(&d_pooled);
    const Tensor d_A = dx.matmul_nt(token_cache);      // [N, P, L]; needs token_channels == C, as in the reference
    const Tensor dx_att = cache2.matmul(token_cache);  // [N, P, C]
    // softmax backward over the pixel axis here: A * (dA - sum_P(A * dA))
    Tensor prod = cache2 * d_A;
    Tensor d_S = d_A;
    for (int n = 0; n < N; ++n)
      for (int l = 0; l < tokens; ++l) {
        float s = 0.0f;
        for (int p = 0; p < P; ++p) s += prod.data[((size_t)n * P + p) * tokens + l];
        for (int p = 0; p < P; ++p) {
          const size_t i = ((size_t)n * P + p) * tokens + l;
          d_S.data[i] = cache2.data[i] * (d_A.data[i] - s);
        }
      }
    const Tensor d_S_flat = d_S.reshape({N * P, tokens});
    linear1->backward(&d_S_flat);
    input_deltas = dx_att + linear1->get_input_deltas().reshape({N, P, C});
  }

  const Tensor &get_input_deltas() const override { return input_deltas; }
  void update_weights(float batch_size) override {
    linear1->update_weights(batch_size);
    linear2->update_weights(batch_size);
  }
  void zero_grad() override {
    linear1->zero_grad();
    linear2->zero_grad();
  }
  bool has_weights() const override { return true; }
  void set_training(bool train) override {
    is_training = train;
    linear1->set_training(train);
    linear2->set_training(train);
  }
  std::string get_type() const override { return "FilterTokenizer"; }
  // linear1 weights, bias, linear2 weights, bias (:216-236)
  std::vector<Tensor> get_parameters() const { return {linear1->weights, linear1->bias, linear2->weights, linear2->bias}; }
  void set_parameters(const std::vector<Tensor> &p) {
    if (p.size() != 4) throw std::runtime_error("FilterTokenizer: Expected 4 parameter tensors");
    linear1->set_parameters(p[0], p[1]);
    linear2->set_parameters(p[2], p[3]);
  }
  DenseLayer *dense(int which) { return which == 0 ? linear1 : linear2; }  // gradient access for tests / tools
};
