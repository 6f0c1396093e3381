// The reference's Model base (model/model.hpp:6-59): a list of layers run in sequence.
#pragma once
#include <vector>

#include "../layers/layer.hpp"

class Model {
 protected:
  std::vector<Layer *> layers;
  Tensor output;

 public:
  virtual ~Model() {
    for (Layer *l : layers) delete l;
  }
  virtual Tensor forward(const Tensor &input) {
    std::vector<Tensor> x{input};
    for (Layer *l : layers) x[0] = l->forward(x)[0];
    output = x[0];
    return output;
  }
  virtual void backward(const Tensor &targets) {
    for (int i = (int)layers.size() - 1; i >= 0; --i)
      layers[i]->backward(&targets, i + 1 < (int)layers.size() ? layers[i + 1] : nullptr);
  }
  virtual void update_weights(float batch_size) {
    for (Layer *l : layers)
      if (l->has_weights()) l->update_weights(batch_size);
  }
  virtual void zero_grad() {
    for (Layer *l : layers) l->zero_grad();
  }
  virtual void set_training(bool is_training) {
    for (Layer *l : layers) l->set_training(is_training);
  }
  const Tensor &get_output() const { return output; }
};
