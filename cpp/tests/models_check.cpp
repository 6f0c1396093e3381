// Runs the two other models of the reference, CNN (model/cnn.hpp:13-119) and MLP (model/mlp.hpp:13-178, with its
// DropoutLayer, layers/dropout_layer.hpp:4-52), through the Layer mirrors for one training step each, the way the
// reference's Trainer steps a model (utils/trainer.hpp:41-56: forward, backward(one-hot), update_weights(32), zero_grad),
// and writes every tensor the Python test compares with the real reference's dump of the same step:
//   models_check <N> <in.bin: N*784 CNN image floats, N label floats, N*784 MLP image floats, N label floats> <out.bin>
// record = i32 name length, name, i32 count, float32 data.  Each model is constructed right after Tensor::set_seed(42),
// so its weights and the MLP's dropout masks are the reference's draws from Tensor::global_gen.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "layers/all_layers.hpp"
#include "model/cnn.hpp"
#include "model/mlp.hpp"

std::mt19937 Tensor::global_gen(42);

namespace {
template <class M>
struct Open : M {  // the layer list is a protected member of Model, as in the reference
  std::vector<Layer *> &list() { return this->layers; }
};

FILE *fo = nullptr;
void put(const std::string &name, const Tensor &t) {
  const int len = (int)name.size(), cnt = (int)t.data.size();
  std::fwrite(&len, sizeof(int), 1, fo);
  std::fwrite(name.data(), 1, len, fo);
  std::fwrite(&cnt, sizeof(int), 1, fo);
  std::fwrite(t.data.data(), sizeof(float), cnt, fo);
}

template <class M>
void dump(const std::string &pfx, const std::string &tag, Open<M> &m, bool grads) {
  int li = 0;
  for (Layer *l : m.list()) {
    const std::string n = pfx + "." + tag + ".l" + std::to_string(li++);
    if (auto *d = dynamic_cast<DenseLayer *>(l)) {
      put(n + ".W", grads ? d->grad_weights : d->weights);
      put(n + ".b", grads ? d->grad_bias : d->bias);
    } else if (auto *c = dynamic_cast<Conv2DLayer *>(l)) {
      Tensor w, b;
      c->get_parameters(w, b);
      put(n + ".W", grads ? c->get_grad_weights() : w);
      put(n + ".b", grads ? c->get_grad_biases() : b);
    } else if (auto *bn = dynamic_cast<BatchNorm2DLayer *>(l)) {
      Tensor g, be;
      bn->get_parameters(g, be);
      put(n + ".gamma", grads ? bn->get_grad_gamma() : g);
      put(n + ".beta", grads ? bn->get_grad_beta() : be);
      if (!grads) {
        put(n + ".running_mean", bn->get_running_mean());
        put(n + ".running_var", bn->get_running_var());
      }
    }
  }
}

template <class M>
void step(const std::string &pfx, int K, const Tensor &X, const Tensor &labels) {
  Tensor::set_seed(42);
  Open<M> m;
  const Tensor onehot = Tensor::one_hot(labels, K);
  dump(pfx, "p0", m, false);
  put(pfx + ".preds", m.forward(X));
  m.backward(onehot);
  dump(pfx, "g", m, true);
  m.update_weights(32.0f);
  m.zero_grad();
  dump(pfx, "p1", m, false);
  put(pfx + ".preds2", m.forward(X));
  m.set_training(false);
  put(pfx + ".preds_eval", m.forward(X));
  m.set_training(true);
}
}  // namespace

int main(int argc, char **argv) {
  if (argc != 4) {
    std::fprintf(stderr, "usage: models_check N in.bin out.bin\n");
    return 2;
  }
  const int N = std::atoi(argv[1]);
  try {
    Tensor xc({N, 1, 28, 28}), yc({N, 1}), xm({N, 1, 28, 28}), ym({N, 1});
    FILE *fi = std::fopen(argv[2], "rb");
    if (!fi) return 4;
    for (Tensor *t : {&xc, &yc, &xm, &ym})
      if (std::fread(t->data.data(), sizeof(float), t->data.size(), fi) != t->data.size()) return 4;
    std::fclose(fi);
    fo = std::fopen(argv[3], "wb");
    if (!fo) return 4;
    step<CNN>("cnn", 10, xc, yc);
    step<MLP>("mlp", 8, xm, ym);
    std::fclose(fo);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "models_check: %s\n", e.what());
    return 1;
  }
  return 0;
}
