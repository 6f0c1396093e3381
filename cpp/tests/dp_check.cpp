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

  // the overlapped form: three buckets per rank on the exchange streams, ordered behind an asynchronous refill of the
  // buffers on the compute streams and joined before an asynchronous read-back on them
  {
    std::vector<std::vector<float>> mine(G, std::vector<float>(n)), got(G, std::vector<float>(n));
    std::vector<float *> pin_in(G), pin_out(G);
    for (int r = 0; r < G; ++r) {
      CHECK(hipSetDevice(r) == hipSuccess);
      CHECK(hipHostMalloc((void **)&pin_in[r], n * sizeof(float)) == hipSuccess);
      CHECK(hipHostMalloc((void **)&pin_out[r], n * sizeof(float)) == hipSuccess);
      for (long i = 0; i < n; ++i) pin_in[r][i] = (r + 1) * pattern[i];
    }
    const long cut1 = first + count / 3, cut2 = first + 2 * (count / 3), end = first + count;
    std::vector<int> brc(G, -100);
    std::vector<std::thread> bw;
    for (int r = 0; r < G; ++r)
      bw.emplace_back([&, r]() {
        int rc = 0;
        if (hipSetDevice(r) != hipSuccess) rc = -99;
        hipStream_t cs = (hipStream_t)streams[r];
        if (rc == 0 && hipMemcpyAsync(bufs[r], pin_in[r], n * sizeof(float), hipMemcpyHostToDevice, cs) != hipSuccess) rc = -98;
        // reverse order of the arena, as gradients become final
        if (rc == 0) rc = vt_dp_exchange_begin(dp, r, bufs[r], cut2, end - cut2, streams[r]);
        if (rc == 0) rc = vt_dp_exchange_begin(dp, r, bufs[r], cut1, cut2 - cut1, streams[r]);
        if (rc == 0) rc = vt_dp_exchange_begin(dp, r, bufs[r], first, cut1 - first, streams[r]);
        if (rc == 0) rc = vt_dp_exchange_join(dp, r, streams[r]);
        if (rc == 0 && hipMemcpyAsync(pin_out[r], bufs[r], n * sizeof(float), hipMemcpyDeviceToHost, cs) != hipSuccess) rc = -97;
        if (rc == 0 && hipStreamSynchronize(cs) != hipSuccess) rc = -96;
        brc[r] = rc;
      });
    for (std::thread &w : bw) w.join();
    for (int r = 0; r < G; ++r) CHECK(brc[r] == 0);
    const float total = (float)(G * (G + 1) / 2);
    for (int r = 0; r < G; ++r)
      for (long i = 0; i < n; ++i) {
        const bool inside = i >= first && i < first + count;
        CHECK(pin_out[r][i] == (inside ? total * pattern[i] : (r + 1) * pattern[i]));
      }
    CHECK(vt_dp_exchange_begin(dp, G, bufs[0], 0, 4, streams[0]) == VT_EINVAL);
    CHECK(vt_dp_exchange_join(dp, -1, streams[0]) == VT_EINVAL);
    for (int r = 0; r < G; ++r) {
      (void)hipHostFree(pin_in[r]);
      (void)hipHostFree(pin_out[r]);
    }
  }

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
