// Host-side Tensor of the MI355X VTCNN engine.
//
// Keeps the reference's public surface that callers poke directly -- the `shape` / `data` fields,
// the global generator `Tensor::global_gen`, one_hot, the loss / accuracy / F1 / confusion helpers
// (reference utils/tensor/tensor_cpu.hpp:13-1380) -- and adds what the device backend needs:
// an optional device mirror (`to_device` / `from_device`) living on the engine's HIP stream.
// The fused hot path runs in libvtcnn_hip.so behind the model handle.  The reference's Tensor operations
// (tensor_cpu.hpp:138-1121) are offered too, with its semantics (quirks included: permute relabels without
// moving data, concat returns an unfilled tensor, operator* keeps the left operand's shape), for hosts that
// keep their own Layer objects: matrix products and last-axis softmax run on the device (vt_gemm_f32,
// vt_softmax_rows), index shuffles and elementwise broadcasts on the host.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <initializer_list>
#include <istream>
#include <ostream>
#include <limits>
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

#include "../../include/vtcnn_hip.h"

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
// makes `device` the calling thread's HIP device for the scope's lifetime
struct DeviceScope {
  int previous = 0;
  explicit DeviceScope(int device) {
    hip_check(hipGetDevice(&previous), "hipGetDevice");
    if (device != previous) hip_check(hipSetDevice(device), "hipSetDevice");
  }
  ~DeviceScope() { (void)hipSetDevice(previous); }
  DeviceScope(const DeviceScope &) = delete;
  DeviceScope &operator=(const DeviceScope &) = delete;
};
}  // namespace vt_host

class Tensor {
 public:
  std::vector<int> shape;
  std::vector<float> data;
  static std::mt19937 global_gen;  // defined by the program, seeded with 42 (reference main.cpp:12)
  static void set_seed(unsigned int seed) { global_gen.seed(seed); }

  Tensor() {}
  // copies are deep copies of shape and data (tensor_cpu.hpp:1124-1129); a copy never shares the
  // original's device mirror
  Tensor(const Tensor &o) : shape(o.shape), data(o.data) {}
  Tensor &operator=(const Tensor &o) {
    if (this != &o) {
      shape = o.shape;
      data = o.data;
      dev_.reset();
    }
    return *this;
  }
  Tensor(Tensor &&) = default;
  Tensor &operator=(Tensor &&) = default;
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

  // the checkpoint's tensor record: i32 ndim, i32 dims[ndim], i32 count, f32 data[count], little-endian
  // (tensor_cpu.hpp:39-61)
  void serialize(std::ostream &out) const {
    const int nd = (int)shape.size(), cnt = (int)data.size();
    out.write(reinterpret_cast<const char *>(&nd), sizeof(int));
    out.write(reinterpret_cast<const char *>(shape.data()), sizeof(int) * nd);
    out.write(reinterpret_cast<const char *>(&cnt), sizeof(int));
    out.write(reinterpret_cast<const char *>(data.data()), sizeof(float) * cnt);
  }
  void deserialize(std::istream &in) {
    int nd = 0, cnt = 0;
    in.read(reinterpret_cast<char *>(&nd), sizeof(int));
    if (!in || nd < 0 || nd > 8) throw std::runtime_error("Tensor::deserialize: bad record");
    shape.assign(nd, 0);
    in.read(reinterpret_cast<char *>(shape.data()), sizeof(int) * nd);
    in.read(reinterpret_cast<char *>(&cnt), sizeof(int));
    if (!in || cnt < 0) throw std::runtime_error("Tensor::deserialize: bad record");
    data.assign(cnt, 0.0f);
    in.read(reinterpret_cast<char *>(data.data()), sizeof(float) * cnt);
    dev_.reset();
  }

  // ---- initialisers (tensor_cpu.hpp:64-116): every draw comes from global_gen, element by element ---------
  static Tensor rand_uniform(const std::vector<int> &shp, float low, float high) {
    Tensor t(shp);
    std::uniform_real_distribution<float> dist(low, high);
    for (float &x : t.data) x = dist(global_gen);
    return t;
  }
  static Tensor normal_(const std::vector<int> &shp, float sigma) {
    Tensor t(shp);
    std::normal_distribution<float> dist(0.0f, sigma);
    for (float &x : t.data) x = dist(global_gen);
    return t;
  }
  static Tensor xavier_normal(const std::vector<int> &shp, int fan_in, int fan_out) { return normal_(shp, std::sqrt(2.0f / (fan_in + fan_out))); }
  static Tensor xavier_uniform(const std::vector<int> &shp, int fan_in, int fan_out) {
    const float a = std::sqrt(6.0f / (fan_in + fan_out));
    return rand_uniform(shp, -a, a);
  }
  static Tensor kaiming_normal(const std::vector<int> &shp, int fan_in, bool = true) { return normal_(shp, std::sqrt(2.0f / fan_in)); }
  static Tensor kaiming_uniform(const std::vector<int> &shp, int fan_in, bool = true) {
    const float a = std::sqrt(6.0f / fan_in);
    return rand_uniform(shp, -a, a);
  }
  static Tensor bernoulli(const std::vector<int> &shp, float p) {
    Tensor t(shp);
    std::bernoulli_distribution dist(p);
    for (float &x : t.data) x = dist(global_gen) ? 1.0f : 0.0f;
    return t;
  }

  // ---- element access, row-major (tensor_cpu.hpp:293-346) ------------------------------------------------
  size_t offset_of(const int *idx, size_t n) const {
    size_t off = 0;
    for (size_t i = 0; i < n; ++i) off = off * shape[i] + idx[i];
    return off;
  }
  float at(const std::vector<int> &index) const { return data[offset_of(index.data(), index.size())]; }
  float &at(std::initializer_list<int> index) { return data[offset_of(index.begin(), index.size())]; }
  const float &at(std::initializer_list<int> index) const { return data[offset_of(index.begin(), index.size())]; }

  // row-major strides of a shape
  static std::vector<long> strides_of(const std::vector<int> &shp) {
    std::vector<long> st(shp.size(), 1);
    for (int i = (int)shp.size() - 2; i >= 0; --i) st[i] = st[i + 1] * shp[i + 1];
    return st;
  }

  // keeps dims [0, start_dim) and folds the rest into one (tensor_cpu.hpp:348-369)
  Tensor flatten(int start_dim) const {
    if (start_dim < 0 || start_dim >= (int)shape.size()) throw std::invalid_argument("Invalid start_dim in flatten.");
    std::vector<int> ns(shape.begin(), shape.begin() + start_dim);
    ns.push_back(std::accumulate(shape.begin() + start_dim, shape.end(), 1, std::multiplies<int>()));
    Tensor t = *this;
    t.shape = ns;
    return t;
  }

  // ---- index shuffles ------------------------------------------------------------------------------------
  // box [starts, ends) of the tensor (tensor_cpu.hpp:138-184)
  Tensor slice(const std::vector<int> &starts, const std::vector<int> &ends) const {
    if (starts.size() != shape.size() || ends.size() != shape.size()) throw std::invalid_argument("slice: rank mismatch");
    std::vector<int> ns(shape.size());
    for (size_t i = 0; i < shape.size(); ++i) {
      if (starts[i] < 0 || ends[i] > shape[i] || starts[i] >= ends[i]) throw std::out_of_range("slice: bad bounds");
      ns[i] = ends[i] - starts[i];
    }
    Tensor out(ns);
    const std::vector<long> st = strides_of(shape);
    std::vector<int> idx(shape.size(), 0);
    for (size_t o = 0; o < out.data.size(); ++o) {
      long src = 0;
      for (size_t d = 0; d < idx.size(); ++d) src += (long)(starts[d] + idx[d]) * st[d];
      out.data[o] = data[src];
      for (int d = (int)idx.size() - 1; d >= 0 && ++idx[d] == ns[d]; --d) idx[d] = 0;
    }
    return out;
  }
  // [start, end) along one axis (tensor_cpu.hpp:186-216)
  Tensor slice(int dim, int start, int end) const {
    std::vector<int> a(shape.size(), 0), b = shape;
    a.at(dim) = start;
    b.at(dim) = end;
    return slice(a, b);
  }
  // The reference assigns every input to a temporary slice of the result, so the result it returns is
  // never written: a zero tensor of the concatenated shape (tensor_cpu.hpp:218-256, the line at :251).
  // reference_compat = false gives the concatenation the name promises.
  static Tensor concat(const std::vector<Tensor> &tensors, int dim, bool reference_compat = true) {
    if (tensors.empty()) throw std::invalid_argument("concat: no tensors");
    std::vector<int> ns = tensors[0].shape;
    ns.at(dim) = 0;
    for (const Tensor &t : tensors) ns[dim] += t.shape.at(dim);
    Tensor out(ns);
    if (reference_compat) return out;
    const std::vector<long> st = strides_of(ns);
    int at_dim = 0;
    for (const Tensor &t : tensors) {
      std::vector<int> idx(ns.size(), 0);
      for (size_t i = 0; i < t.data.size(); ++i) {
        long dst = 0;
        for (size_t d = 0; d < idx.size(); ++d) dst += (long)(idx[d] + ((int)d == dim ? at_dim : 0)) * st[d];
        out.data[dst] = t.data[i];
        for (int d = (int)idx.size() - 1; d >= 0 && ++idx[d] == t.shape[d]; --d) idx[d] = 0;
      }
      at_dim += t.shape[dim];
    }
    return out;
  }
  // zero padding of an [N,C,H,W] tensor, pads = {N0,N1,C0,C1,H0,H1,W0,W1} (tensor_cpu.hpp:258-291)
  Tensor pad(const std::vector<int> &pads) const {
    if (shape.size() != 4 || pads.size() != 8) throw std::invalid_argument("pad: needs a 4-D tensor and 8 pads");
    std::vector<int> ns(4), lo(4);
    for (int d = 0; d < 4; ++d) {
      lo[d] = pads[2 * d];
      ns[d] = shape[d] + pads[2 * d] + pads[2 * d + 1];
    }
    Tensor out(ns);
    const std::vector<long> st = strides_of(ns);
    size_t i = 0;
    for (int n = 0; n < shape[0]; ++n)
      for (int c = 0; c < shape[1]; ++c)
        for (int h = 0; h < shape[2]; ++h) {
          float *dst = out.data.data() + (n + lo[0]) * st[0] + (c + lo[1]) * st[1] + (h + lo[2]) * st[2] + lo[3];
          for (int w = 0; w < shape[3]; ++w) dst[w] = data[i++];
        }
    return out;
  }
  // real axis permutation: out[i0', i1', ..] = in[..] with out axis k = in axis axes[k] (tensor_cpu.hpp:640-692)
  Tensor transpose(const std::vector<int> &axes) const {
    if (axes.size() != shape.size()) throw std::runtime_error("Número de ejes no coincide con las dimensiones del tensor");
    std::vector<int> ns(shape.size());
    for (size_t k = 0; k < axes.size(); ++k) ns[k] = shape.at(axes[k]);
    const std::vector<long> st = strides_of(shape);
    Tensor out(ns);
    std::vector<int> idx(ns.size(), 0);
    for (size_t o = 0; o < out.data.size(); ++o) {
      long src = 0;
      for (size_t k = 0; k < idx.size(); ++k) src += idx[k] * st[axes[k]];
      out.data[o] = data[src];
      for (int k = (int)idx.size() - 1; k >= 0 && ++idx[k] == ns[k]; --k) idx[k] = 0;
    }
    return out;
  }
  // The reference's permute writes element idx to the offset computed from the PERMUTED indices and the
  // PERMUTED strides, which is idx again: the shape is relabelled, the data stays (tensor_cpu.hpp:694-751).
  // reference_compat = false performs the permutation (= transpose).
  Tensor permute(const std::vector<int> &dims, bool reference_compat = true) const {
    if (dims.size() != shape.size()) throw std::runtime_error("La cantidad de dimensiones no coincide con el tamaño del tensor.");
    std::vector<bool> seen(shape.size(), false);
    for (int d : dims) {
      if (d < 0 || d >= (int)shape.size() || seen[d]) throw std::runtime_error("Permutación inválida.");
      seen[d] = true;
    }
    if (!reference_compat) return transpose(dims);
    Tensor out = *this;
    for (size_t k = 0; k < dims.size(); ++k) out.shape[k] = shape[dims[k]];
    return out;
  }

  // ---- reductions ----------------------------------------------------------------------------------------
  // sum over a set of axes, the remaining axes keep their order; sequential fp32 per output element in
  // input order (tensor_cpu.hpp:451-523 with one thread)
  Tensor sum(const std::vector<int> &dims) const {
    std::vector<bool> red(shape.size(), false);
    for (int d : dims) red.at(d < 0 ? d + (int)shape.size() : d) = true;
    std::vector<int> ns;
    for (size_t i = 0; i < shape.size(); ++i)
      if (!red[i]) ns.push_back(shape[i]);
    if (ns.empty()) ns.push_back(1);
    Tensor out(ns);
    std::vector<long> ost(shape.size(), 0);  // stride of each input axis inside the output (0 = reduced)
    {
      long s = 1;
      for (int i = (int)shape.size() - 1; i >= 0; --i)
        if (!red[i]) {
          ost[i] = s;
          s *= shape[i];
        }
    }
    std::vector<int> idx(shape.size(), 0);
    long o = 0;
    for (size_t i = 0; i < data.size(); ++i) {
      out.data[o] += data[i];
      for (int d = (int)idx.size() - 1; d >= 0; --d) {
        o += ost[d];
        if (++idx[d] < shape[d]) break;
        o -= ost[d] * shape[d];
        idx[d] = 0;
      }
    }
    return out;
  }
  // the two forms the reference implements (tensor_cpu.hpp:376-450): a matrix over its rows, a 3-D tensor
  // over its last axis; anything else throws there and here
  Tensor sum(int axis, bool keepdims = false) const {
    if (axis < 0) axis += (int)shape.size();
    if (!((shape.size() == 2 && axis == 0) || (shape.size() == 3 && axis == 2))) throw std::runtime_error("sum: unsupported shape or axis");
    Tensor out = sum(std::vector<int>{axis});
    if (keepdims) {
      out.shape = shape;
      out.shape[axis] = 1;
    }
    return out;
  }
  // sequential sum along one axis, then a true division by the count (tensor_cpu.hpp:525-571)
  Tensor mean(int dim, bool keepdim) const {
    if (dim < 0 || dim >= (int)shape.size()) throw std::out_of_range("mean: bad axis");
    Tensor out = sum(std::vector<int>{dim});
    for (float &v : out.data) v = v / (float)shape[dim];
    out.shape = shape;
    if (keepdim) out.shape[dim] = 1; else out.shape.erase(out.shape.begin() + dim);
    return out;
  }
  // sum over the axes times the reciprocal of the count (operator/ multiplies, tensor_cpu.hpp:573-599)
  Tensor mean(const std::vector<int> &dims, bool keepdims) const {
    int cnt = 1;
    for (int d : dims) cnt *= shape.at(d);
    Tensor out = sum(dims) / (float)cnt;
    if (keepdims) {
      std::vector<int> ns = shape;
      for (int d : dims) ns[d] = 1;
      out.shape = ns;
    }
    return out;
  }

  // ---- elementwise ---------------------------------------------------------------------------------------
  template <typename F>
  Tensor map(F f) const {
    Tensor out = *this;
    for (float &v : out.data) v = f(v);
    return out;
  }
  Tensor pow(float e) const { return map([e](float v) { return std::pow(v, e); }); }
  Tensor sqrt() const { return map([](float v) { return std::sqrt(v); }); }
  Tensor reciprocal() const { return map([](float v) { return 1.0f / v; }); }
  Tensor relu() const { return map([](float v) { return std::max(0.0f, v); }); }
  Tensor relu_derivative() const { return map([](float v) { return v > 0 ? 1.0f : 0.0f; }); }
  Tensor operator/(float s) const { const float r = 1.0f / s; return map([r](float v) { return v * r; }); }  // reciprocal multiply
  Tensor operator+(float s) const { return map([s](float v) { return v + s; }); }
  Tensor operator*(float s) const { return map([s](float v) { return v * s; }); }

  // result has the LEFT operand's shape; the right operand is read at coordinate 0 along every axis where
  // its extent is 1 (tensor_cpu.hpp:1014-1084).  A left operand [1,C] times a right operand [R,C] therefore
  // reads row 0 of the right one: the source of the BatchNorm backward's row-0 term.
  template <typename F>
  Tensor broadcast_right(const Tensor &o, F f, const char *what) const {
    if (o.shape.size() != shape.size()) throw std::runtime_error(std::string(what) + ": rank mismatch");
    Tensor out(shape);
    const std::vector<long> st = strides_of(o.shape);
    std::vector<int> idx(shape.size(), 0);
    for (size_t i = 0; i < data.size(); ++i) {
      long j = 0;
      for (size_t d = 0; d < idx.size(); ++d) j += (o.shape[d] == 1 ? 0 : idx[d]) * st[d];
      out.data[i] = f(data[i], o.data[j]);
      for (int d = (int)idx.size() - 1; d >= 0 && ++idx[d] == shape[d]; --d) idx[d] = 0;
    }
    return out;
  }
  Tensor operator-(const Tensor &o) const { return broadcast_right(o, [](float a, float b) { return a - b; }, "operator-"); }
  Tensor operator*(const Tensor &o) const { return broadcast_right(o, [](float a, float b) { return a * b; }, "operator*"); }
  // same shape, [B,D] + [1,D], [B,C,H,W] + [1,C,1,1]; nothing else (tensor_cpu.hpp:954-1012)
  Tensor operator+(const Tensor &o) const {
    const bool same = shape == o.shape;
    const bool rows = shape.size() == 2 && o.shape.size() == 2 && o.shape[0] == 1 && shape[1] == o.shape[1];
    const bool chan = shape.size() == 4 && o.shape.size() == 4 && o.shape[0] == 1 && o.shape[2] == 1 && o.shape[3] == 1 && shape[1] == o.shape[1];
    if (!same && !rows && !chan) throw std::runtime_error("operator+: Shapes incompatibles para suma con broadcasting");
    return broadcast_right(o, [](float a, float b) { return a + b; }, "operator+");
  }

  // ---- products and softmax: on the device ---------------------------------------------------------------
  // [m,n] x [n,p] or batched [b,m,n] x [b,n,p] (tensor_cpu.hpp:754-814) as one vt_gemm_f32 launch.
  // ta / tb read the operand transposed (per batch member) without materialising the transpose; bias [p] is
  // added to every row (DenseLayer's output + bias.reshape({1,-1}), dense_layer.hpp:70-73).
  Tensor gemm(bool ta, bool tb, const Tensor &o, const Tensor *bias = nullptr) const {
    const bool two = shape.size() == 2 && o.shape.size() == 2, three = shape.size() == 3 && o.shape.size() == 3;
    if (!two && !three) throw std::invalid_argument("matmul only supports 2D or 3D tensors with matching inner dimensions");
    const int k0 = three ? 1 : 0, b = three ? shape[0] : 1;
    const int m = ta ? shape[k0 + 1] : shape[k0], n = ta ? shape[k0] : shape[k0 + 1];
    const int n2 = tb ? o.shape[k0 + 1] : o.shape[k0], p = tb ? o.shape[k0] : o.shape[k0 + 1];
    if (n != n2 || (three && o.shape[0] != b)) throw std::invalid_argument("matmul only supports 2D or 3D tensors with matching inner dimensions");
    if (bias && (int)bias->data.size() != p) throw std::invalid_argument("matmul: bias length mismatch");
    Tensor out(three ? std::vector<int>{b, m, p} : std::vector<int>{m, p});
    if (out.data.empty()) return out;
    if (n == 0) {
      if (bias) for (size_t i = 0; i < out.data.size(); ++i) out.data[i] = bias->data[i % p];
      return out;
    }
    const int rc = vt_gemm_f32(ta, tb, m, p, n, 1.0f, to_device(), shape[k0 + 1], (long)shape[k0] * shape[k0 + 1], o.to_device(),
                               o.shape[k0 + 1], (long)o.shape[k0] * o.shape[k0 + 1], 0, out.device_buffer(), p, (long)m * p,
                               bias ? bias->to_device() : nullptr, b, nullptr, 0, nullptr);
    if (rc != 0) throw std::runtime_error(std::string("matmul: ") + vt_error_string(rc));
    out.from_device();
    return out;
  }
  Tensor matmul(const Tensor &o, const Tensor *bias = nullptr) const { return gemm(false, false, o, bias); }
  Tensor matmul_tn(const Tensor &o) const { return gemm(true, false, o); }  // this^T . o
  Tensor matmul_nt(const Tensor &o, const Tensor *bias = nullptr) const { return gemm(false, true, o, bias); }  // this . o^T
  // max-subtract, exp, sequential sum, divide (tensor_cpu.hpp:894-944): the last axis on the device,
  // any other axis on the host
  Tensor softmax(int dim) const {
    if (dim < 0 || dim >= (int)shape.size()) throw std::out_of_range("softmax: bad axis");
    Tensor out = *this;
    if (data.empty()) return out;
    if (dim == (int)shape.size() - 1) {
      const int w = shape.back();
      const int rc = vt_softmax_rows(out.to_device(), (long)(data.size() / w), w, nullptr);
      if (rc != 0) throw std::runtime_error(std::string("softmax: ") + vt_error_string(rc));
      out.from_device();
      return out;
    }
    long inner = 1;
    for (size_t i = dim + 1; i < shape.size(); ++i) inner *= shape[i];
    const long n = shape[dim], outer = (long)data.size() / (n * inner);
    for (long a = 0; a < outer; ++a)
      for (long c = 0; c < inner; ++c) {
        float *col = out.data.data() + a * n * inner + c;
        float mx = -std::numeric_limits<float>::infinity(), s = 0.0f;
        for (long k = 0; k < n; ++k) mx = std::max(mx, col[k * inner]);
        for (long k = 0; k < n; ++k) s += (col[k * inner] = std::exp(col[k * inner] - mx));
        for (long k = 0; k < n; ++k) col[k * inner] /= s;
      }
    return out;
  }
  // s * (delta - sum_last(delta * s)): always over the LAST axis, whatever axis the softmax itself ran over
  // (tensor_cpu.hpp:946-950); rank 2 or 3
  static Tensor softmax_backward(const Tensor &s, const Tensor &delta) {
    if (s.shape != delta.shape || s.shape.empty()) throw std::runtime_error("softmax_backward: shape mismatch");
    Tensor out = delta;
    if (s.data.empty()) return out;
    const int w = s.shape.back();
    const int rc = vt_softmax_bwd_rows(s.to_device(), out.to_device(), (long)(s.data.size() / w), w, nullptr);
    if (rc != 0) throw std::runtime_error(std::string("softmax_backward: ") + vt_error_string(rc));
    out.from_device();
    return out;
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
