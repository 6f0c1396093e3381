// Cost of a dependent kernel launch on the GPU side: chains of trivial kernels in one stream, timed with events
// (the host enqueues far ahead of the device).  Variants: an empty kernel; a kernel that does one load -> store
// round trip; the same chain with hipExtAnyOrderLaunch (no barrier between the packets); an event record between
// every two kernels (what a cross-stream fork puts on the critical stream).
//   hipcc --offload-arch=gfx950 -O2 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void k_empty() {}
__global__ void k_touch(float *p) {
  if (threadIdx.x == 0) p[blockIdx.x] = p[blockIdx.x] + 1.f;
}

template <typename F>
static float chain(F launch, int n, hipStream_t s) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 200; ++i) launch(i);
  hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int i = 0; i < n; ++i) launch(i);
  hipEventRecord(b, s);
  hipStreamSynchronize(s);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / n;
}

int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  float *p;
  hipMalloc(&p, 1 << 20);
  hipMemset(p, 0, 1 << 20);
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  const int n = 2000;
  printf("empty kernel, 1 workgroup:            %.2f us per launch\n", chain([&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, n, s));
  printf("empty kernel, 1024 workgroups x 256:  %.2f us per launch\n", chain([&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s); }, n, s));
  printf("load+store kernel, 32 workgroups:     %.2f us per launch\n", chain([&](int) { hipLaunchKernelGGL(k_touch, dim3(32), dim3(256), 0, s, p); }, n, s));
  printf("empty kernel, any-order launch:       %.2f us per launch\n",
         chain([&](int) { hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch); }, n, s));
  printf("empty kernel + event record:          %.2f us per pair\n", chain([&](int) {
           hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
           hipEventRecord(ev, s);
         }, n, s));
  printf("empty kernel signalling its event:    %.2f us per launch\n",
         chain([&](int) { hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, nullptr, ev, 0); }, n, s));
  return 0;
}
