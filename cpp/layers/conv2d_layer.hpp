// Conv2DLayer behind the reference's surface (layers/conv2d_layer.hpp:4-292): any kernel size, stride and
// zero padding over [N,C,H,W].  The layer keeps the reference's tensors (weights [Co,Ci,kh,kw], biases,
// their gradients, input deltas); the arithmetic is one device product per pass over the unfolded input
// ("im2col": one row per output pixel holding its Ci*kh*kw window, zeros outside the image):
//   forward   out_rows = col . W^T + b                     (:35-87)
//   backward  gW = delta_rows^T . col, gb = column sums,   (:115-147)
//             input deltas = fold(delta_rows . W_flipped)  (:150-190; the reference flips the kernel here)
// get_type() is not overridden: a checkpoint names this layer "Layer" (layers/layer.hpp:29).
#pragma once
#include "dense_layer.hpp"

class Conv2DLayer : public Layer {
  int in_channels, out_channels, kernel_h, kernel_w, stride, padding;
  Tensor weights, biases, grad_weights, grad_biases, input, input_deltas, output;
  Tensor col_;  // unfolded input of the last forward, reused by backward
  int Ho_ = 0, Wo_ = 0;

  // calls f(row, column inside the row, flat input offset) for every in-image element of every window
  template <typename F>
  void for_each_window_element(F f) const {
    const int N = input.shape[0], H = input.shape[2], W = input.shape[3];
    long row = 0;
    for (int n = 0; n < N; ++n)
      for (int ho = 0; ho < Ho_; ++ho)
        for (int wo = 0; wo < Wo_; ++wo, ++row)
          for (int ci = 0; ci < in_channels; ++ci)
            for (int kh = 0; kh < kernel_h; ++kh) {
              const int h = ho * stride + kh - padding;
              if (h < 0 || h >= H) continue;
              for (int kw = 0; kw < kernel_w; ++kw) {
                const int w = wo * stride + kw - padding;
                if (w < 0 || w >= W) continue;
                f(row, (ci * kernel_h + kh) * kernel_w + kw, (((long)n * in_channels + ci) * H + h) * W + w);
              }
            }
  }

 public:
  // Kaiming-normal weights from Tensor::global_gen, zero biases (conv2d_layer.hpp:20-33)
  Conv2DLayer(int in_channels_, int out_channels_, int kernel_h_, int kernel_w_, int stride_ = 1, int padding_ = 0)
      : in_channels(in_channels_), out_channels(out_channels_), kernel_h(kernel_h_), kernel_w(kernel_w_), stride(stride_), padding(padding_) {
    weights = Tensor::kaiming_normal({out_channels, in_channels, kernel_h, kernel_w}, in_channels * kernel_h * kernel_w);
    biases = Tensor::zeros({out_channels});
    grad_weights = Tensor::zeros(weights.shape);
    grad_biases = Tensor::zeros(biases.shape);
  }

  std::vector<Tensor> forward(const std::vector<Tensor> &inputs) override {
    input = inputs.at(0);
    if (input.shape.size() != 4 || input.shape[1] != in_channels) throw std::runtime_error("Conv2DLayer::forward: expects [N,C_in,H,W]");
    const int N = input.shape[0], CK = in_channels * kernel_h * kernel_w;
    Ho_ = (input.shape[2] + 2 * padding - kernel_h) / stride + 1;
    Wo_ = (input.shape[3] + 2 * padding - kernel_w) / stride + 1;
    col_ = Tensor({N * Ho_ * Wo_, CK});
    for_each_window_element([&](long row, int c, long src) { col_.data[row * CK + c] = input.data[src]; });
    const Tensor rows = col_.matmul_nt(weights.reshape({out_channels, CK}), &biases);  // [N*Ho*Wo, Co]
    output = rows.reshape({N, Ho_ * Wo_, out_channels}).transpose({0, 2, 1}).reshape({N, out_channels, Ho_, Wo_});
    return {output};
  }

  void backward(const Tensor * = nullptr, const Layer *next_layer = nullptr) override {
    if (!next_layer) throw std::runtime_error("Conv2DLayer::backward: next_layer required");
    const Tensor &delta = next_layer->get_input_deltas();  // [N, Co, Ho, Wo]
    const int N = input.shape[0], CK = in_channels * kernel_h * kernel_w;
    if ((long)delta.data.size() != (long)N * out_channels * Ho_ * Wo_) throw std::runtime_error("Conv2DLayer::backward: delta size mismatch");
    const Tensor rows = delta.reshape({N, out_channels, Ho_ * Wo_}).transpose({0, 2, 1}).reshape({N * Ho_ * Wo_, out_channels});
    grad_weights = rows.matmul_tn(col_).reshape(weights.shape);
    grad_biases = rows.sum(0);
    // The reference scatters delta * W[co][ci][K-1-kh][K-1-kw] to padded position (ho*stride + kh, wo*stride + kw)
    // (:150-176): the kernel is flipped where the forward pass (:62-81) used it unflipped, so the input delta is the
    // gradient of a convolution with the flipped kernel.  Reproduced as is (a model's first convolution never uses it;
    // the CNN model's second one does).
    Tensor flipped(weights.shape);
    for (int co = 0; co < out_channels; ++co)
      for (int ci = 0; ci < in_channels; ++ci)
        for (int kh = 0; kh < kernel_h; ++kh)
          for (int kw = 0; kw < kernel_w; ++kw)
            flipped.data[((size_t)(co * in_channels + ci) * kernel_h + kh) * kernel_w + kw] =
                weights.data[((size_t)(co * in_channels + ci) * kernel_h + (kernel_h - 1 - kh)) * kernel_w + (kernel_w - 1 - kw)];
    const Tensor dcol = rows.matmul(flipped.reshape({out_channels, CK}));  // [rows, CK]
    input_deltas = Tensor::zeros(input.shape);
    for_each_window_element([&](long row, int c, long src) { input_deltas.data[src] += dcol.data[row * CK + c]; });
  }

  void update_weights(float batch_size) override { vt_host::scaled_update(optimizer, weights, grad_weights, biases, grad_biases, batch_size); }
  void zero_grad() override {
    grad_weights.fill(0.0f);
    grad_biases.fill(0.0f);
  }
  void set_training(bool training) override { is_training = training; }
  const Tensor &get_input_deltas() const override { return input_deltas; }
  bool has_weights() const override { return true; }
  void get_parameters(Tensor &out_weights, Tensor &out_biases) const {
    out_weights = weights;
    out_biases = biases;
  }
  void set_parameters(const Tensor &new_weights, const Tensor &new_biases) {
    if (new_weights.shape != weights.shape || new_biases.shape != biases.shape) throw std::runtime_error("Conv2DLayer::set_parameters: shape mismatch");
    weights = new_weights;
    biases = new_biases;
  }
  // two tensor records: weights, biases (conv2d_layer.hpp:245-290)
  void save(std::ostream &out) const {
    weights.serialize(out);
    biases.serialize(out);
  }
  void load(std::istream &in) {
    Tensor w, b;
    w.deserialize(in);
    b.deserialize(in);
    set_parameters(w, b);
  }
  const Tensor &get_grad_weights() const { return grad_weights; }
  const Tensor &get_grad_biases() const { return grad_biases; }
};
