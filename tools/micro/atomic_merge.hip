// Microbenchmark: what does it cost a launch of 1024 workgroups to fold per-workgroup partial sums into a
// handful of 64-bit integer accumulators with device-scope atomics (order-independent, so deterministic),
// compared with writing them to a partial slab for a later finalize launch?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_merge.hip -o tools/micro/atomic_merge.bin && ./tools/micro/atomic_merge.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_work(const float *__restrict__ in, float *__restrict__ out, float *__restrict__ slab,
                                              unsigned long long *__restrict__ acc, int nvals, int mode, int replicas) {
  // a stand-in for a per-image kernel: stream 25 KB in, 25 KB out
  const int n = blockIdx.x, tid = threadIdx.x;
  const float4 *src = reinterpret_cast<const float4 *>(in) + (size_t)n * 1568;
  float4 *dst = reinterpret_cast<float4 *>(out) + (size_t)n * 1568;
  float s = 0.f;
  for (int i = tid; i < 1568; i += 256) {
    float4 v = src[i];
    s += v.x + v.y + v.z + v.w;
    dst[i] = v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float red[4];
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  if (tid < nvals) {
    const float v = s * (1.0f + tid);
    if (mode == 0) {
      slab[(size_t)n * nvals + tid] = v;
    } else {
      const long long q = (long long)((double)v * 1048576.0);
      atomicAdd(&acc[(size_t)(n % replicas) * nvals + tid], (unsigned long long)q);
    }
  }
}

int main() {
  const int N = 1024;
  float *in, *out, *slab;
  unsigned long long *acc;
  hipMalloc(&in, (size_t)N * 1568 * 16);
  hipMalloc(&out, (size_t)N * 1568 * 16);
  hipMalloc(&slab, (size_t)N * 256 * 4);
  hipMalloc(&acc, 64 * 256 * 8);
  hipMemset(in, 0, (size_t)N * 1568 * 16);
  hipMemset(acc, 0, 64 * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int reps = 200;
  for (int nvals : {16, 32, 64, 128, 192}) {
    for (int mode = 0; mode < 2; ++mode) {
      for (int replicas : {1, 8, 64}) {
        if (mode == 0 && replicas != 1) continue;
        for (int w = 0; w < 2; ++w) {
          hipEventRecord(e0);
          for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_work, dim3(N), dim3(256), 0, 0, in, out, slab, acc, nvals, mode, replicas);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("nvals %3d  %-6s replicas %2d : %.2f us per launch\n", nvals, mode ? "atomic" : "slab", replicas, ms * 1e3 / reps);
      }
    }
  }
  return 0;
}
