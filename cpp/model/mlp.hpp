// MLP (reference model/mlp.hpp:13-178): dense 784->128, ReLU, dropout 0.25, dense 128->64, ReLU, dense 64->8,
// softmax; SGD(0.01) on the dense layers.  save / load use the reference's layout: i32 layer count, then per
// layer i32 type length, type string and, for weighted layers, the weight and bias tensor records.
// main.cpp includes this header but never constructs the class.
#pragma once
#include <fstream>

#include "../layers/all_layers.hpp"
#include "model.hpp"

class MLP : public Model {
 public:
  MLP() {
    layers = {new DenseLayer(784, 128), new ReLULayer(), new DropoutLayer(0.25f), new DenseLayer(128, 64),
              new ReLULayer(),          new DenseLayer(64, 8), new SoftmaxLayer()};
    for (Layer *l : layers)
      if (l->has_weights()) l->set_optimizer(new SGD(0.01f));
  }
  // the input is taken as [batch, 784] whatever its shape (:130-138)
  Tensor forward(const Tensor &input) override { return Model::forward(input.reshape({input.shape.at(0), 784})); }
  void backward(const Tensor &target) override {
    layers.back()->backward(&target, nullptr);
    for (int i = (int)layers.size() - 2; i >= 0; --i) layers[i]->backward(nullptr, layers[i + 1]);
  }
  void save(const std::string &filename) {
    std::ofstream out(filename, std::ios::binary);
    if (!out) throw std::runtime_error("No se pudo abrir el archivo para guardar el modelo");
    const int n = (int)layers.size();
    out.write(reinterpret_cast<const char *>(&n), sizeof(int));
    for (Layer *l : layers) {
      const std::string type = l->get_type();
      const int len = (int)type.size();
      out.write(reinterpret_cast<const char *>(&len), sizeof(int));
      out.write(type.data(), len);
      if (l->has_weights()) {
        Tensor w, b;
        static_cast<DenseLayer *>(l)->get_parameters(w, b);
        w.serialize(out);
        b.serialize(out);
      }
    }
  }
  void load(const std::string &filename) {
    std::ifstream in(filename, std::ios::binary);
    if (!in) throw std::runtime_error("No se pudo abrir el archivo para cargar el modelo");
    int n = 0;
    in.read(reinterpret_cast<char *>(&n), sizeof(int));
    if (n != (int)layers.size()) throw std::runtime_error("Cantidad de capas no coincide");
    for (Layer *l : layers) {
      int len = 0;
      in.read(reinterpret_cast<char *>(&len), sizeof(int));
      std::string type(len > 0 && len < 64 ? len : 0, ' ');
      in.read(&type[0], type.size());
      if (type != l->get_type()) throw std::runtime_error("Tipo de capa no coincide");
      if (l->has_weights()) {
        Tensor w, b;
        w.deserialize(in);
        b.deserialize(in);
        static_cast<DenseLayer *>(l)->set_parameters(w, b);
      }
    }
  }
};
