// Builds VTCNN out of the individual Layer mirrors (cpp/layers/*.hpp) the way the reference's model wires
// them (model/vit_cnn.hpp:56-100 construction, :103-213 forward, :215-242 backward, :244-251 update), runs one
// training step on the batch in <in.bin> and writes every tensor the Python test compares with the reference's
// own dump:  weights before the step, probabilities, every gradient, weights after update_weights(32).
//   layer_chain_check <N> <K> <in.bin: N*784 image floats, N label floats> <out.bin: records>
// record = i32 name length, name, i32 count, float32 data
#include <cstdio>
#include <cstdlib>
#include <functional>

#include "layers/all_layers.hpp"
#include "model/cnn.hpp"
#include "model/mlp.hpp"

std::mt19937 Tensor::global_gen(42);

namespace {
struct Chain {
  Conv2DLayer conv{1, 8, 3, 3, 1, 1};
  BatchNorm2DLayer bn{8};
  VisualTransformer vt{8, 16, 32, 16, "filter", 32, 2, 2, 64, 0.1f, true};
  MaxPool2DLayer pool{2, 2};
  FlattenLayer flatten;
  DenseLayer fc;
  SoftmaxLayer softmax;
  std::vector<int> nchw;

  explicit Chain(int K) : fc(16 * 14 * 14, K) {
    for (Layer *l : std::vector<Layer *>{&conv, &bn, &vt, &fc, &softmax}) l->set_optimizer(new SGD(0.001f));
  }
  Tensor forward(const Tensor &images) {
    Tensor x = bn.forward({conv.forward({images})[0]})[0];  // [N, 8, 28, 28]
    nchw = x.shape;
    x = x.flatten(2).transpose({0, 2, 1});                   // [N, 784, 8]
    x = vt.forward({x})[0];                                  // [N, 784, 16]
    x = x.reshape({nchw[0], 16, nchw[2], nchw[3]});          // relabelled, not transposed
    return softmax.forward({fc.forward({flatten.forward({pool.forward({x})[0]})[0]})[0]})[0];
  }
  void backward(const Tensor &onehot) {
    softmax.backward(&onehot, nullptr);
    fc.backward(nullptr, &softmax);
    flatten.backward(nullptr, &fc);
    pool.backward(nullptr, &flatten);
    vt.backward(nullptr, &pool);
    const Tensor d = vt.get_input_deltas().transpose({0, 2, 1}).reshape(nchw);
    bn.backward(&d, nullptr);
    conv.backward(nullptr, &bn);
  }
  void update(float batch_size) {
    conv.update_weights(batch_size);
    bn.update_weights(batch_size);
    fc.update_weights(batch_size);
    vt.update_weights(batch_size);
  }
  // name -> tensor, weights (grads == false) or gradients, in the names of the reference dump
  void each(bool grads, const std::function<void(const std::string &, const Tensor &)> &f) {
    auto dense = [&](const std::string &n, DenseLayer *d, bool with_bias) {
      f(n + ".W", grads ? d->grad_weights : d->weights);
      if (with_bias) f(n + ".b", grads ? d->grad_bias : d->bias);
    };
    auto norm = [&](const std::string &n, BatchNormLayer1D *b) {
      Tensor g, be;
      b->get_parameters(g, be);
      f(n + ".gamma", grads ? b->get_grad_gamma() : g);
      f(n + ".beta", grads ? b->get_grad_beta() : be);
    };
    Tensor cw, cb, bg, bb;
    conv.get_parameters(cw, cb);
    bn.get_parameters(bg, bb);
    f("conv.W", grads ? conv.get_grad_weights() : cw);
    f("conv.b", grads ? conv.get_grad_biases() : cb);
    f("bn.gamma", grads ? bn.get_grad_gamma() : bg);
    f("bn.beta", grads ? bn.get_grad_beta() : bb);
    TransformerLayer *tr = vt.transformer_layer();
    for (int h = 0; h < 2; ++h)
      for (int w = 0; w < 3; ++w) dense(std::string("mha.") + "qkv"[w] + std::to_string(h), tr->attention_layer()->dense(h, w), false);
    dense("mha.o", tr->attention_layer()->dense(-1, 0), false);
    dense("ffn.linear1", tr->dense(0), true);
    dense("ffn.linear2", tr->dense(1), true);
    norm("norm1", tr->batch_norm(0));
    norm("norm2", tr->batch_norm(1));
    ProjectorLayer *pr = vt.projector_layer();
    dense("proj.linear1", pr->dense(0), false);
    dense("proj.linear2", pr->dense(1), false);
    dense("proj.linear3", pr->dense(2), true);
    norm("proj.norm", pr->batch_norm());
    dense("proj.downsample", pr->dense(3), true);
    dense("fc", &fc, true);
    if (!grads) {
      dense("tok.linear1", vt.tokenizer_layer()->dense(0), true);
      dense("tok.linear2", vt.tokenizer_layer()->dense(1), true);
    }
  }
};
}  // namespace

int main(int argc, char **argv) {
  if (argc != 5) {
    std::fprintf(stderr, "usage: layer_chain_check N K in.bin out.bin\n");
    return 2;
  }
  const int N = std::atoi(argv[1]), K = std::atoi(argv[2]);
  try {
    Tensor images({N, 1, 28, 28}), labels({N, 1});
    FILE *fi = std::fopen(argv[3], "rb");
    if (!fi || std::fread(images.data.data(), sizeof(float), images.data.size(), fi) != images.data.size() ||
        std::fread(labels.data.data(), sizeof(float), labels.data.size(), fi) != labels.data.size())
      return 4;
    std::fclose(fi);
    FILE *fo = std::fopen(argv[4], "wb");
    if (!fo) return 4;
    auto put = [&](const std::string &name, const Tensor &t) {
      const int len = (int)name.size(), cnt = (int)t.data.size();
      std::fwrite(&len, sizeof(int), 1, fo);
      std::fwrite(name.data(), 1, len, fo);
      std::fwrite(&cnt, sizeof(int), 1, fo);
      std::fwrite(t.data.data(), sizeof(float), cnt, fo);
    };
    Tensor::set_seed(42);
    Chain net(K);
    net.each(false, [&](const std::string &n, const Tensor &t) { put("p0." + n, t); });
    const Tensor preds = net.forward(images);
    put("fwd.preds", preds);
    net.backward(Tensor::one_hot(labels, K));
    net.each(true, [&](const std::string &n, const Tensor &t) { put("g." + n, t); });
    net.update(32.0f);
    net.each(false, [&](const std::string &n, const Tensor &t) { put("p1." + n, t); });
    // the reference's own main.cpp compiles against these headers: the classes it names exist
    static_assert(sizeof(CNN *) && sizeof(MLP *), "model headers");
    std::fclose(fo);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "layer_chain_check: %s\n", e.what());
    return 1;
  }
  return 0;
}
