// VisualTransformer behind the reference's surface (layers/visual_transformer.hpp:4-378): FilterTokenizer ->
// permute -> TransformerLayer -> permute -> ProjectorLayer.  The permutes go through Tensor::permute, which in
// the reference relabels the shape without moving data: the transformer sees the [N*L, C] token rows in (n, l)
// order labelled [L, N, C].  forward returns {feature map [N,P,Cout], tokens [N,L,D]} (:66-125).  backward
// (:127-199): projector first; the transformer is seeded with the projector's token delta as TARGETS; the
// tokenizer receives the projector's token delta and input delta (not the transformer's).  Only "filter"
// tokenizers exist (the recurrent one is commented out in the reference, :35, and dereferenced unset at :53).
#pragma once
#include "projector_layer.hpp"
#include "tokenizer.hpp"
#include "transformer_layer.hpp"

class VisualTransformer : public Layer {
  int in_channels, out_channels, token_channels, tokens, attn_dim;
  bool is_projected;
  std::string tokenizer_type;
  FilterTokenizer *tokenizer = nullptr;
  TransformerLayer *transformer = nullptr;
  ProjectorLayer *projector = nullptr;
  Tensor input_cache, token_cache, input_deltas;

 public:
  VisualTransformer(int in_channels_, int out_channels_, int token_channels_, int tokens_, const std::string &tokenizer_type_,
                    int attn_dim_, int /*transformer_enc_layers*/, int transformer_heads, int /*transformer_fc_dim*/,
                    float /*transformer_dropout*/, bool is_projected_ = true)
      : in_channels(in_channels_), out_channels(out_channels_), token_channels(token_channels_), tokens(tokens_),
        attn_dim(attn_dim_), is_projected(is_projected_), tokenizer_type(tokenizer_type_) {
    if (tokenizer_type == "recurrent") throw std::runtime_error("VisualTransformer: the recurrent tokenizer does not exist (the reference never constructs it)");
    if (tokenizer_type != "filter") throw std::runtime_error("Unknown tokenizer type: " + tokenizer_type);
    if (!is_projected) throw std::runtime_error("VisualTransformer: is_projected = false is not supported (the reference dereferences the missing projector)");
    tokenizer = new FilterTokenizer(in_channels, token_channels, tokens);
    transformer = new TransformerLayer(token_channels, attn_dim, transformer_heads);
    projector = new ProjectorLayer(in_channels, out_channels, token_channels);
    for (Layer *l : std::vector<Layer *>{tokenizer, transformer, projector}) l->set_optimizer(new SGD(0.001f));
  }
  VisualTransformer(const VisualTransformer &) = delete;
  VisualTransformer &operator=(const VisualTransformer &) = delete;
  ~VisualTransformer() override {
    delete tokenizer;
    delete transformer;
    delete projector;
  }

  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    input_cache = inputs.at(0);  // [N, P, Cin]
    Tensor t = tokenizer->forward({input_cache})[0];               // [N, L, D]
    t = transformer->forward({t.permute({1, 0, 2})})[0];           // labelled [L, N, D]
    token_cache = t.permute({1, 0, 2});                            // labelled [N, L, D] again
    return {projector->forward({input_cache, token_cache})[0], token_cache};
  }

  void backward(const Tensor * = nullptr, const Layer *next_layer = nullptr) override {
    projector->backward(nullptr, next_layer);
    const Tensor &delta_x = projector->get_input_deltas();
    const Tensor delta_t = projector->get_token_deltas();
    const Tensor seed = delta_t.permute({1, 0, 2});
    transformer->backward(&seed);  // targets: the layer starts from out - targets
    tokenizer->backward2(delta_t, delta_x);
    input_deltas = tokenizer->get_input_deltas();
  }

  void update_weights(float batch_size) override {
    for (Layer *l : std::vector<Layer *>{tokenizer, transformer, projector}) l->update_weights(batch_size);
  }
  void zero_grad() override {
    for (Layer *l : std::vector<Layer *>{tokenizer, transformer, projector}) l->zero_grad();
  }
  void set_training(bool is_training_) override { is_training = is_training_; }  // not passed on (:221)
  const Tensor &get_input_deltas() const override { return input_deltas; }
  bool has_weights() const override { return true; }
  std::string get_type() const override { return "VisualTransformer"; }

  // tokenizer (4) + projector (8): the transformer reports has_weights() == false and is left out (:244-270)
  std::vector<Tensor> get_parameters() const {
    std::vector<Tensor> p = tokenizer->get_parameters();
    if (transformer->has_weights()) {
      const std::vector<Tensor> tp = transformer->get_parameters();
      p.insert(p.end(), tp.begin(), tp.end());
    }
    const std::vector<Tensor> pp = projector->get_parameters();
    p.insert(p.end(), pp.begin(), pp.end());
    return p;
  }
  void set_parameters(const std::vector<Tensor> &p) {
    size_t at = 0;
    auto take = [&](size_t n) {
      if (at + n > p.size()) throw std::runtime_error("VisualTransformer: too few parameter tensors");
      std::vector<Tensor> v(p.begin() + at, p.begin() + at + n);
      at += n;
      return v;
    };
    tokenizer->set_parameters(take(4));
    if (transformer->has_weights()) transformer->set_parameters(take(transformer->get_parameters().size()));
    projector->set_parameters(take(8));
  }
  FilterTokenizer *tokenizer_layer() { return tokenizer; }
  TransformerLayer *transformer_layer() { return transformer; }
  ProjectorLayer *projector_layer() { return projector; }
};
