// Issue rate of the bf16 matrix instructions used by the kernels here: cycles per instruction on one SIMD, four
// independent accumulators, one wave per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>

#include <cstdio>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters) {
  f32x4 acc[4];
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
  bf16x8 a8, b8;
  for (int q = 0; q < 8; ++q) {
    a8[q] = (__bf16)(float)(threadIdx.x + q);
    b8[q] = (__bf16)(float)(q + 1);
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[j], 0, 0, 0);
        else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[j], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char *name, float *out) {
  const int iters = 20000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k_rate<KIND>, dim3(256), dim3(256), 0, 0, out, 100);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k_rate<KIND>, dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(b, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  const double per = ms * 1e-3 / ((double)iters * 16);  // seconds per instruction of one wave
  printf("%-28s %.2f ns per instruction per SIMD = %.1f cycles at 2.4 GHz\n", name, per * 1e9, per * 2.4e9);
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 256 * sizeof(float));
  run<0>("v_mfma_f32_16x16x16_bf16", out);
  run<1>("v_mfma_f32_16x16x32_bf16", out);
  return 0;
}
