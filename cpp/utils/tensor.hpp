// Host-side Tensor of the MI355X VTCNN engine.
//
// Keeps the reference's public surface that callers poke directly -- the `shape` / `data` fields,
// the global generator `Tensor::global_gen`, one_hot, the loss / accuracy / F1 / confusion helpers
// (reference utils/tensor/tensor_cpu.hpp:13-1380) -- and adds what the device backend needs:
// an optional device mirror (`to_device` / `from_device`) living on the engine's HIP stream.
// All arithmetic of the hot path runs in libvtcnn_hip.so; this class only carries data.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <functional>
#include <memory>
#include <numeric>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

namespace vt_host {
inline void hip_check(hipError_t e, const char *what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
// Owns one device allocation.
struct DeviceBuffer {
  float *ptr = nullptr;
  size_t count = 0;
  explicit DeviceBuffer(size_t n) : count(n) { hip_check(hipMalloc((void **)&ptr, std::max<size_t>(n, 1) * sizeof(float)), "hipMalloc"); }
  ~DeviceBuffer() { if (ptr) (void)hipFree(ptr); }
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
};
}  // namespace vt_host

class Tensor {
 public:
  std::vector<int> shape;
  std::vector<float> data;
  static std::mt19937 global_gen;  // defined by the program, seeded with 42 (reference main.cpp:12)
  static void set_seed(unsigned int seed) { global_gen.seed(seed); }

  Tensor() {}
  explicit Tensor(const std::vector<int> &shp) : shape(shp), data(count_of(shp), 0.0f) {}
  Tensor(const std::vector<int> &shp, std::vector<float> values) : shape(shp), data(std::move(values)) {
    data.resize(count_of(shp), 0.0f);
  }
  static Tensor zeros(const std::vector<int> &shp) { return Tensor(shp); }

  bool empty() const { return data.empty(); }
  void fill(float v) { std::fill(data.begin(), data.end(), v); }

  // reshape with one optional -1 wildcard; the bytes stay where they are
  Tensor reshape(std::vector<int> new_shape) const {
    long known = 1;
    int wild = -1;
    for (size_t i = 0; i < new_shape.size(); ++i) {
      if (new_shape[i] == -1) wild = (int)i; else known *= new_shape[i];
    }
    if (wild >= 0) new_shape[wild] = known ? (int)(data.size() / known) : 0;
    if ((size_t)count_of(new_shape) != data.size()) throw std::runtime_error("reshape: element count mismatch");
    Tensor t;
    t.shape = new_shape;
    t.data = data;
    return t;
  }

  std::vector<int> to_vector() const {
    std::vector<int> out(data.size());
    std::transform(data.begin(), data.end(), out.begin(), [](float f) { return static_cast<int>(f); });
    return out;
  }

  std::string printsummary(std::string name = "") const {
    std::ostringstream os;
    os << (name.empty() ? std::string("Tensor:\n") : "Tensor: " + name + "\n");
    os << "Tensor data: " << data.size() << ", shape size: " << shape.size() << ", shape: [";
    for (size_t i = 0; i < shape.size(); ++i) os << shape[i] << (i + 1 < shape.size() ? ", " : "");
    os << "]";
    return os.str();
  }

  // labels [B,1] (class index stored as float) -> one-hot [B,K]  (tensor_cpu.hpp:118-136)
  static Tensor one_hot(const Tensor &labels, int num_classes) {
    if (labels.shape.size() != 2 || labels.shape[1] != 1)
      throw std::invalid_argument("Labels tensor debe tener shape [batch_size, 1]");
    const int B = labels.shape[0];
    Tensor out({B, num_classes});
    for (int i = 0; i < B; ++i) {
      const int k = static_cast<int>(labels.data[i]);
      if (k < 0 || k >= num_classes) throw std::out_of_range("Label fuera de rango en one_hot");
      out.data[(size_t)i * num_classes + k] = 1.0f;
    }
    return out;
  }

  // mean over the batch of -sum_k y*log(p + 1e-8)  (tensor_cpu.hpp:1155-1173)
  static float compute_loss(const Tensor &preds, const Tensor &targets_onehot) {
    const int B = preds.shape[0], K = preds.shape[1];
    float loss = 0.0f;
    for (int i = 0; i < B * K; ++i) loss += -targets_onehot.data[i] * std::log(preds.data[i] + 1e-8f);
    return loss / B;
  }

  // first maximum wins on ties (std::max_element semantics, tensor_cpu.hpp:1144-1146, 1294-1307)
  static std::vector<int> argmax_batch(const Tensor &preds) {
    const int B = preds.shape[0], K = preds.shape[1];
    std::vector<int> out(B);
    for (int i = 0; i < B; ++i) {
      const float *row = preds.data.data() + (size_t)i * K;
      out[i] = (int)(std::max_element(row, row + K) - row);
    }
    return out;
  }

  static float compute_accuracy(const Tensor &preds, const Tensor &targets, int /*num_classes*/) {
    const std::vector<int> am = argmax_batch(preds);
    int hits = 0;
    for (size_t i = 0; i < am.size(); ++i) hits += (am[i] == static_cast<int>(targets.data[i]));
    return static_cast<float>(hits) / (float)am.size();
  }

  static std::vector<std::vector<int>> confusion_matrix(const std::vector<int> &truth, const std::vector<int> &pred,
                                                        int K) {
    std::vector<std::vector<int>> cm(K, std::vector<int>(K, 0));
    for (size_t i = 0; i < truth.size(); ++i)
      if (truth[i] >= 0 && truth[i] < K && pred[i] >= 0 && pred[i] < K) ++cm[truth[i]][pred[i]];
    return cm;
  }

  // per-class precision / recall / F1 and their macro (classes with instances) and weighted means
  // (tensor_cpu.hpp:1194-1290)
  struct ClassStats {
    std::vector<float> precision, recall, f1;
    float macro_precision = 0, macro_recall = 0, macro_f1 = 0, weighted_f1 = 0;
  };
  static ClassStats class_stats(const std::vector<int> &truth, const std::vector<int> &pred, int K) {
    const auto cm = confusion_matrix(truth, pred, K);
    ClassStats s;
    s.precision.assign(K, 0.f);
    s.recall.assign(K, 0.f);
    s.f1.assign(K, 0.f);
    int present = 0, total = 0;
    for (int k = 0; k < K; ++k) {
      int col = 0, row = 0;
      for (int j = 0; j < K; ++j) { col += cm[j][k]; row += cm[k][j]; }
      if (col > 0) s.precision[k] = static_cast<float>(cm[k][k]) / col;
      if (row > 0) s.recall[k] = static_cast<float>(cm[k][k]) / row;
      if (s.precision[k] + s.recall[k] > 0) s.f1[k] = 2.0f * (s.precision[k] * s.recall[k]) / (s.precision[k] + s.recall[k]);
      if (row > 0) {
        s.macro_precision += s.precision[k];
        s.macro_recall += s.recall[k];
        s.macro_f1 += s.f1[k];
        s.weighted_f1 += s.f1[k] * row;
        total += row;
        ++present;
      }
    }
    if (present > 0) { s.macro_precision /= present; s.macro_recall /= present; s.macro_f1 /= present; }
    if (total > 0) s.weighted_f1 /= total;
    return s;
  }
  static std::tuple<float, float, std::vector<float>, std::vector<float>> precision_recall(
      const std::vector<int> &truth, const std::vector<int> &pred, int K) {
    const ClassStats s = class_stats(truth, pred, K);
    return {s.macro_precision, s.macro_recall, s.precision, s.recall};
  }
  static std::tuple<float, float, std::vector<float>> f1_score(const std::vector<int> &truth, const std::vector<int> &pred,
                                                               int K) {
    const ClassStats s = class_stats(truth, pred, K);
    return {s.macro_f1, s.weighted_f1, s.f1};
  }

  // ---- device mirror ---------------------------------------------------------------------
  // Uploads `data` to a device buffer owned by this tensor (allocated on first use) on `stream`.
  float *to_device(hipStream_t stream = nullptr) const {
    if (!dev_ || dev_->count != data.size()) dev_ = std::make_shared<vt_host::DeviceBuffer>(data.size());
    vt_host::hip_check(hipMemcpyAsync(dev_->ptr, data.data(), data.size() * sizeof(float), hipMemcpyHostToDevice, stream),
                       "Tensor::to_device");
    return dev_->ptr;
  }
  // The device mirror itself (allocated on first use), without copying `data` into it: for buffers
  // that a kernel fills and a later kernel reads.
  float *device_buffer() const {
    if (!dev_ || dev_->count != data.size()) dev_ = std::make_shared<vt_host::DeviceBuffer>(data.size());
    return dev_->ptr;
  }
  // Downloads the device mirror back into `data` (synchronises `stream`).
  void from_device(hipStream_t stream = nullptr) {
    if (!dev_) return;
    vt_host::hip_check(hipMemcpyAsync(data.data(), dev_->ptr, data.size() * sizeof(float), hipMemcpyDeviceToHost, stream),
                       "Tensor::from_device");
    vt_host::hip_check(hipStreamSynchronize(stream), "Tensor::from_device");
  }
  Tensor to_device(bool /*use_cuda*/) const { return *this; }  // reference signature (tensor_cpu.hpp:1131)

 private:
  static int count_of(const std::vector<int> &shp) {
    return std::accumulate(shp.begin(), shp.end(), 1, std::multiplies<int>());
  }
  mutable std::shared_ptr<vt_host::DeviceBuffer> dev_;
};
