// Host-side Tensor operations against the reference's semantics (utils/tensor/tensor_cpu.hpp), quirks
// included.  No device call is made: runs anywhere.  Exit code 0 = every check passed.
#include <cstdio>

#include "utils/tensor.hpp"

std::mt19937 Tensor::global_gen(42);

static int failures = 0;
#define CHECK(cond)                                                      \
  do {                                                                   \
    if (!(cond)) {                                                       \
      std::fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond);     \
      ++failures;                                                        \
    }                                                                    \
  } while (0)

static Tensor iota(const std::vector<int> &shape) {
  Tensor t(shape);
  for (size_t i = 0; i < t.data.size(); ++i) t.data[i] = (float)i;
  return t;
}

int main() {
  const Tensor a = iota({2, 3, 4});
  // permute relabels the shape and leaves every byte where it was (tensor_cpu.hpp:694-751); transpose moves data
  const Tensor p = a.permute({1, 0, 2});
  CHECK((p.shape == std::vector<int>{3, 2, 4}) && p.data == a.data);
  const Tensor t = a.transpose({1, 0, 2});
  CHECK((t.shape == std::vector<int>{3, 2, 4}) && t.at({1, 0, 2}) == a.at({0, 1, 2}) && t.at({2, 1, 3}) == a.at({1, 2, 3}));
  CHECK(a.permute({1, 0, 2}, false).data == t.data);
  // concat returns an unfilled tensor of the concatenated shape (tensor_cpu.hpp:218-256)
  const Tensor c = Tensor::concat({a, a}, 2);
  CHECK((c.shape == std::vector<int>{2, 3, 8}));
  bool all_zero = true;
  for (float v : c.data) all_zero = all_zero && v == 0.0f;
  CHECK(all_zero);
  const Tensor c2 = Tensor::concat({a, a}, 2, false);
  CHECK(c2.at({1, 2, 3}) == a.at({1, 2, 3}) && c2.at({1, 2, 7}) == a.at({1, 2, 3}));
  // slices
  const Tensor s = a.slice({0, 1, 2}, {2, 3, 4});
  CHECK((s.shape == std::vector<int>{2, 2, 2}) && s.at({1, 1, 1}) == a.at({1, 2, 3}) && s.at({0, 0, 0}) == a.at({0, 1, 2}));
  CHECK(a.slice(2, 1, 3).data == a.slice({0, 0, 1}, {2, 3, 3}).data);
  // reductions
  const Tensor m = iota({3, 4});
  const Tensor col = m.sum(0);
  CHECK((col.shape == std::vector<int>{4}) && col.data[0] == 12.0f && col.data[3] == 21.0f);
  CHECK((m.sum(0, true).shape == std::vector<int>{1, 4}));
  const Tensor last = a.sum(-1, true);
  CHECK((last.shape == std::vector<int>{2, 3, 1}) && last.data[0] == 6.0f && last.data[5] == 86.0f);
  bool threw = false;
  try { a.sum(1); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);  // only the two forms the reference implements
  const Tensor mu = m.mean(0, true);
  CHECK((mu.shape == std::vector<int>{1, 4}) && mu.data[1] == 5.0f);
  const Tensor nchw = iota({2, 3, 2, 2});
  const Tensor cm = nchw.mean(std::vector<int>{0, 2, 3}, true);
  CHECK((cm.shape == std::vector<int>{1, 3, 1, 1}) && std::fabs(cm.data[1] - 11.5f) < 1e-5f);
  // operator* keeps the LEFT shape and reads index 0 of the right operand wherever ITS extent is 1; a [1,C]
  // left operand times a [R,C] right operand therefore uses row 0 only (the BatchNorm backward quirk)
  const Tensor small({1, 4}, {1.f, 2.f, 3.f, 4.f});
  const Tensor prod = small * m;
  CHECK((prod.shape == std::vector<int>{1, 4}) && prod.data[2] == 3.0f * m.at({0, 2}));
  const Tensor prod2 = m * small;
  CHECK((prod2.shape == m.shape) && prod2.at({2, 3}) == 11.0f * 4.0f);
  const Tensor diff = m - mu;
  CHECK(diff.at({0, 1}) == 1.0f - 5.0f && diff.at({2, 1}) == 9.0f - 5.0f);
  const Tensor bias({1, 4}, {10.f, 20.f, 30.f, 40.f});
  CHECK((m + bias).at({2, 0}) == 18.0f);
  threw = false;
  try { m + Tensor({3, 1}); } catch (const std::runtime_error &) { threw = true; }
  CHECK(threw);  // operator+ broadcasts only [1,D] and [1,C,1,1]
  CHECK((m / 4.0f).data[5] == 5.0f * (1.0f / 4.0f));
  // softmax over a middle axis (host path) sums to one along that axis
  const Tensor sm = (a * 0.1f).softmax(1);
  CHECK(std::fabs(sm.at({0, 0, 0}) + sm.at({0, 1, 0}) + sm.at({0, 2, 0}) - 1.0f) < 1e-6f);
  // reshape / flatten / pad / one empty-shape tensor holds one element
  CHECK((a.reshape({-1, 4}).shape == std::vector<int>{6, 4}) && (a.flatten(1).shape == std::vector<int>{2, 12}));
  const Tensor pd = nchw.pad({0, 0, 0, 0, 1, 1, 1, 1});
  CHECK((pd.shape == std::vector<int>{2, 3, 4, 4}) && pd.at({1, 2, 1, 1}) == nchw.at({1, 2, 0, 0}) && pd.at({0, 0, 0, 0}) == 0.0f);
  CHECK(Tensor(std::vector<int>{}).data.size() == 1);
  // draws come from global_gen in element order: the same seed gives the same tensor
  Tensor::set_seed(7);
  const Tensor r1 = Tensor::rand_uniform({5}, -1.f, 1.f);
  Tensor::set_seed(7);
  std::uniform_real_distribution<float> dist(-1.f, 1.f);
  std::mt19937 gen(7);
  bool same = true;
  for (float v : r1.data) same = same && v == dist(gen);
  CHECK(same);
  // a copy never shares the original's storage
  Tensor copy = a;
  copy.data[0] = -1.f;
  CHECK(a.data[0] == 0.0f);
  std::printf("tensor_ops_check: %d failure(s)\n", failures);
  return failures == 0 ? 0 : 1;
}
