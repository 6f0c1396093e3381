// Issue cost of the vector instructions the attention sweep is made of, per instruction of one wave, with one
// and with two waves on every SIMD (a lone wave cannot issue back-to-back), measured in shader clock cycles
// (s_memtime runs at a fixed 100 MHz, so the wall clock of the whole launch is used instead and converted with the
// clock the same launch measures on a chain of dependent v_add_f32: 1 instruction = its latency).
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// KIND: 0 v_cvt_pk_bf16_f32, 1 v_pk_mul_f32, 2 v_pk_fma_f32, 3 v_exp_f32, 4 v_and_b32, 5 v_perm_b32, 6 v_fma_f32,
// 7 v_pk_add_f32, 8 v_max3_f32, 9 v_lshlrev_b32
template <int KIND>
__global__ __launch_bounds__(512) void k_rate(float *out, int iters) {
  // eight independent chains so that latency never limits
  float x[8], y[8];
  floatx2 p2[8], q2[8];  // register pairs of the packed instructions (in place: no moves between them)
  for (int j = 0; j < 8; ++j) {
    x[j] = 1.0f + 0.001f * (threadIdx.x + j);
    y[j] = 0.5f + 0.002f * (threadIdx.x + 2 * j);
    p2[j] = floatx2{x[j], y[j]};
    q2[j] = floatx2{1.0f, 0.999f};
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (KIND == 0) {
          unsigned r;
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x[j]), "v"(y[j]));
          x[j] = __uint_as_float(r);
        } else if (KIND == 1) {
          asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2[j]) : "v"(q2[j]));
        } else if (KIND == 2) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p2[j]) : "v"(q2[j]));
        } else if (KIND == 3) {
          asm volatile("v_exp_f32 %0, %1" : "=v"(x[j]) : "v"(x[j]));
        } else if (KIND == 4) {
          asm volatile("v_and_b32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j]), "v"(y[j]));
        } else if (KIND == 5) {
          asm volatile("v_perm_b32 %0, %1, %2, %1" : "=v"(x[j]) : "v"(x[j]), "v"(y[j]));
        } else if (KIND == 6) {
          asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(x[j]) : "v"(x[j]), "v"(y[j]));
        } else if (KIND == 7) {
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2[j]) : "v"(q2[j]));
        } else if (KIND == 8) {
          asm volatile("v_max3_f32 %0, %1, %2, %1" : "=v"(x[j]) : "v"(x[j]), "v"(y[j]));
        } else {
          asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(x[j]) : "v"(x[j]));
        }
      }
  }
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += x[j] + y[j] + p2[j].x + p2[j].y;
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// a chain of dependent v_add_f32 (latency-bound: one per add latency, independent of the issue rules): the clock probe
__global__ __launch_bounds__(256) void k_chain(float *out, int iters) {
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int u = 0; u < 32; ++u) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x));
  out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int KIND>
static void run(const char *name, float *out) {
  const int iters = 20000;
  for (int waves = 4; waves <= 8; waves += 4) {  // waves per workgroup = per CU: 4 = one per SIMD, 8 = two
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(256), dim3(64 * waves), 0, 0, out, 100);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(256), dim3(64 * waves), 0, 0, out, iters);
    hipEventRecord(b, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double per = ms * 1e-3 / ((double)iters * 32);  // seconds per instruction of one wave
    printf("%-20s %d wave(s) per SIMD: %.2f ns per instruction of a wave\n", name, waves / 4, per * 1e9);
  }
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, 0, out, 100);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, 0, out, 20000);
    hipEventRecord(b, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    printf("dependent v_add_f32 chain: %.2f ns per instruction\n", ms * 1e-3 / (20000.0 * 32) * 1e9);
  }
  run<6>("v_fma_f32", out);
  run<0>("v_cvt_pk_bf16_f32", out);
  run<1>("v_pk_mul_f32", out);
  run<7>("v_pk_add_f32", out);
  run<2>("v_pk_fma_f32", out);
  run<3>("v_exp_f32", out);
  run<4>("v_and_b32", out);
  run<9>("v_lshlrev_b32", out);
  run<5>("v_perm_b32", out);
  run<8>("v_max3_f32", out);
  return 0;
}
