// Gradient exchange check for the vt_dp_* entry points: argument errors, then an in-place sum over every
// visible GPU (up to 8) issued both ways (one grouped call; one call per rank from a thread of its own).
// With one GPU the sum is the identity; with G GPUs rank r holds (r+1) * pattern and every rank must end
// with G(G+1)/2 * pattern, bit for bit (small integers: exact in fp32 whatever the reduction order).
#include <cstdio>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/vtcnn_hip.h"

#define CHECK(c)                                                   \
  do {                                                             \
    if (!(c)) {                                                    \
      std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c);      \
      return 1;                                                    \
    }                                                              \
  } while (0)

int main() {
  int have = 0;
  CHECK(hipGetDeviceCount(&have) == hipSuccess && have >= 1);
  const int G = have < 8 ? have : 8;
  vt_dp *dp = nullptr;
  int zero = 0, twice[2] = {0, 0}, beyond = have;
  CHECK(vt_dp_init(0, &zero, &dp) == VT_EINVAL);
  CHECK(vt_dp_init(1, nullptr, &dp) == VT_EINVAL);
  CHECK(vt_dp_init(2, twice, &dp) == VT_EINVAL);
  CHECK(vt_dp_init(1, &beyond, &dp) == VT_EINVAL);
  CHECK(vt_dp_wait(nullptr) == VT_EINVAL);

  std::vector<int> devices(G);
  for (int r = 0; r < G; ++r) devices[r] = r;
  CHECK(vt_dp_init(G, devices.data(), &dp) == 0);
  CHECK(vt_dp_ranks(dp) == G);
  const long n = 100000, first = 37, count = n - 100;
  std::vector<float> pattern(n);
  for (long i = 0; i < n; ++i) pattern[i] = (float)((i * 7) % 31 - 15);
  std::vector<float *> bufs(G);
  std::vector<vt_stream> streams(G);
  auto fill = [&]() {
    for (int r = 0; r < G; ++r) {
      std::vector<float> mine(n);
      for (long i = 0; i < n; ++i) mine[i] = (r + 1) * pattern[i];
      if (hipSetDevice(r) != hipSuccess) return false;
      if (hipMemcpy(bufs[r], mine.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return false;
    }
    return true;
  };
  auto verify = [&]() {
    const float total = (float)(G * (G + 1) / 2);
    for (int r = 0; r < G; ++r) {
      std::vector<float> got(n);
      if (hipSetDevice(r) != hipSuccess) return false;
      if (hipMemcpy(got.data(), bufs[r], n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return false;
      for (long i = 0; i < n; ++i) {
        const bool inside = i >= first && i < first + count;
        const float want = inside ? total * pattern[i] : (r + 1) * pattern[i];
        if (got[i] != want) {
          std::printf("rank %d element %ld: got %g want %g\n", r, i, got[i], want);
          return false;
        }
      }
    }
    return true;
  };
  for (int r = 0; r < G; ++r) {
    CHECK(hipSetDevice(r) == hipSuccess);
    CHECK(hipMalloc((void **)&bufs[r], n * sizeof(float)) == hipSuccess);
    hipStream_t s;
    CHECK(hipStreamCreate(&s) == hipSuccess);
    streams[r] = s;
  }
  CHECK(hipSetDevice(0) == hipSuccess);

  CHECK(fill());
  CHECK(vt_dp_allreduce_async(dp, bufs.data(), first, count, streams.data()) == 0);
  CHECK(vt_dp_wait(dp) == 0);
  CHECK(verify());

  CHECK(fill());
  std::vector<int> rcs(G, -100);
  std::vector<std::thread> workers;
  for (int r = 0; r < G; ++r)
    workers.emplace_back([&, r]() { rcs[r] = vt_dp_allreduce_rank_async(dp, r, bufs[r], first, count, streams[r]); });
  for (std::thread &w : workers) w.join();
  for (int r = 0; r < G; ++r) CHECK(rcs[r] == 0);
  CHECK(vt_dp_wait(dp) == 0);
  CHECK(verify());

  CHECK(vt_dp_allreduce_async(dp, bufs.data(), -1, 4, streams.data()) == VT_EINVAL);
  CHECK(vt_dp_allreduce_rank_async(dp, G, bufs[0], 0, 4, streams[0]) == VT_EINVAL);
  CHECK(vt_dp_allreduce_async(dp, bufs.data(), 0, 0, streams.data()) == 0);
  for (int r = 0; r < G; ++r) {
    (void)hipSetDevice(r);
    (void)hipStreamDestroy((hipStream_t)streams[r]);
    (void)hipFree(bufs[r]);
  }
  CHECK(vt_dp_destroy(dp) == 0);
  std::printf("dp_check: %d GPU(s) OK\n", G);
  return 0;
}
