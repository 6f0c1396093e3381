// Streaming rates of the device memory for footprints far beyond the 256 MB last-level cache: read only (a sum kept
// in registers), write only (a fill), and a copy, each as 16-byte accesses with one contiguous 64 KB region per
// workgroup iteration (the pattern of the row kernels); plain and nontemporal stores.
//   hipcc --offload-arch=gfx950 -O2 -o rw_rate rw_rate.hip && ./rw_rate
#include <hip/hip_runtime.h>

#include <cstdio>

// MODE 0 read, 1 write, 2 copy, 3 write nontemporal, 4 copy with nontemporal stores
template <int MODE>
__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ src, float4 *__restrict__ dst, long n4, float *out) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const long chunk = 4096;  // float4 per workgroup iteration (64 KB)
  for (long c0 = (long)blockIdx.x * chunk; c0 < n4; c0 += (long)gridDim.x * chunk) {
#pragma unroll 4
    for (long i = c0 + threadIdx.x; i < c0 + chunk && i < n4; i += 256) {
      float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
      if (MODE == 0 || MODE == 2 || MODE == 4) v = src[i];
      if (MODE == 0) {
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      } else if (MODE == 3 || MODE == 4) {
        __builtin_nontemporal_store(v.x, &dst[i].x);
        __builtin_nontemporal_store(v.y, &dst[i].y);
        __builtin_nontemporal_store(v.z, &dst[i].z);
        __builtin_nontemporal_store(v.w, &dst[i].w);
      } else {
        dst[i] = v;
      }
    }
  }
  if (MODE == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int MODE>
static void run(const char *name, const float4 *src, float4 *dst, long n4, float *out, double bytes_per_elem) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k_stream<MODE>, dim3(4096), dim3(256), 0, 0, src, dst, n4, out);
  hipEventRecord(a, 0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_stream<MODE>, dim3(4096), dim3(256), 0, 0, src, dst, n4, out);
  hipEventRecord(b, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  printf("%-34s %7.2f ms per pass, %6.0f GB/s\n", name, ms / reps, bytes_per_elem * n4 * reps / (ms * 1e-3) / 1e9);
}

int main() {
  for (long gb = 1; gb <= 8; gb *= 2) {
    const long n4 = gb * (1L << 30) / 16;
    float4 *src, *dst;
    float *out;
    if (hipMalloc(&src, n4 * 16) != hipSuccess || hipMalloc(&dst, n4 * 16) != hipSuccess || hipMalloc(&out, 16) != hipSuccess) return 1;
    hipMemset(src, 0, n4 * 16);
    printf("footprint %ld GB per array\n", gb);
    run<0>("read only", src, dst, n4, out, 16.0);
    run<1>("write only", src, dst, n4, out, 16.0);
    run<3>("write only, nontemporal", src, dst, n4, out, 16.0);
    run<2>("copy (read + write bytes)", src, dst, n4, out, 32.0);
    run<4>("copy, nontemporal stores", src, dst, n4, out, 32.0);
    hipFree(src);
    hipFree(dst);
    hipFree(out);
  }
  return 0;
}
