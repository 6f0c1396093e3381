// Drives the SelfAttention mirror on the inputs the fixture generator gives the reference layer
// (weights from Tensor::global_gen seeded 42; x then delta uniform(-1,1) from mt19937(seed)) and
// writes key.W | query.W | out | g.key.W | g.query.W | dx as raw float32 for the Python test to compare.
//   self_attention_check <B> <T> <D> <Dk> <seed> <out.bin>
#include <cstdio>
#include <cstdlib>

#include "layers/self_attention.hpp"

std::mt19937 Tensor::global_gen(42);

int main(int argc, char **argv) {
  if (argc != 7) {
    std::fprintf(stderr, "usage: self_attention_check B T D Dk seed out.bin\n");
    return 2;
  }
  const int B = std::atoi(argv[1]), T = std::atoi(argv[2]), D = std::atoi(argv[3]), Dk = std::atoi(argv[4]);
  try {
    Tensor::set_seed(42);
    SelfAttention sa(D, Dk);
    std::mt19937 gen((unsigned)std::atoi(argv[5]));
    std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
    Tensor x({B, T, D}), delta({B, T, D});
    for (float &v : x.data) v = dist(gen);
    for (float &v : delta.data) v = dist(gen);
    Tensor out = sa.forward({x})[0];
    sa.backward(&delta);
    bool threw = false;
    try {
      SelfAttention other(D, Dk);
      other.backward();
    } catch (const std::runtime_error &) {
      threw = true;
    }
    if (!threw) return 3;
    FILE *f = std::fopen(argv[6], "wb");
    if (!f) return 4;
    const Tensor *parts[] = {&sa.key_weights, &sa.query_weights, &out, &sa.grad_key_weights, &sa.grad_query_weights,
                             &sa.get_input_deltas()};
    for (const Tensor *t : parts)
      std::fwrite(t->data.data(), sizeof(float), t->data.size(), f);
    std::fclose(f);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "self_attention_check: %s\n", e.what());
    return 1;
  }
  return 0;
}
