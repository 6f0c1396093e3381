// CNN (reference model/cnn.hpp:13-119): conv 1->16, BatchNorm2D, ReLU, pool, conv 16->4, BatchNorm2D, ReLU,
// pool, flatten, dense 196->16, ReLU, dense 16->10, softmax; every weighted layer gets SGD(0.002).  The
// stepping (forward chain, backward from the last layer with the targets, update of weighted layers) is the
// base Model's.  main.cpp includes this header but never constructs the class.
#pragma once
#include "../layers/all_layers.hpp"
#include "model.hpp"

class CNN : public Model {
 public:
  CNN() {
    layers = {new Conv2DLayer(1, 16, 3, 3, 1, 1), new BatchNorm2DLayer(16), new ReLULayer(), new MaxPool2DLayer(2, 2),
              new Conv2DLayer(16, 4, 3, 3, 1, 1), new BatchNorm2DLayer(4),  new ReLULayer(), new MaxPool2DLayer(2, 2),
              new FlattenLayer(),                 new DenseLayer(196, 16),  new ReLULayer(), new DenseLayer(16, 10),
              new SoftmaxLayer()};
    for (Layer *l : layers)
      if (l->has_weights()) l->set_optimizer(new SGD(0.002f));
  }
  void backward(const Tensor &target) override {  // the last layer takes the targets, every other its successor (:72-80)
    layers.back()->backward(&target, nullptr);
    for (int i = (int)layers.size() - 2; i >= 0; --i) layers[i]->backward(nullptr, layers[i + 1]);
  }
  void set_optimizer(Optimizer *opt) {
    for (Layer *l : layers) l->set_optimizer(opt);
  }
};
