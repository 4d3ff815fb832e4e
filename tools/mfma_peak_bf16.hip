// Sustained issue rate of v_mfma_f32_32x32x16_bf16 (the instruction of the three-term "exact product" GEMM modes): a register-only
// loop with 1, 2 or 4 independent accumulators per wave (= how far apart two dependent MFMAs on one accumulator are issued) at
// 1..3 waves per SIMD.  Prices the 2.5 PFLOP/s dense bf16 figure (and with it the 279.6 / 419.4 TFLOP/s fp32-equivalent peaks of
// the nine- / six-product modes) against what the pipe sustains, and shows what a chain of dependent accumulations costs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[ACC];
  for (int i = 0; i < ACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)(a0 + threadIdx.x * 1e-3f + e), b[e] = (__bf16)(b0 + e);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8 / ACC * 2; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < ACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC>
void run(float* out, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 20000;
  for (int wps = 1; wps <= 3; ++wps) {
    const int blocks = 256 * wps;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop<ACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)blocks * 4 * iters * 16 * (32.0 * 32 * 16 * 2);
      if (rep) printf("%d accumulator(s), %d wave(s)/SIMD: %.1f ms  %.0f TFLOP/s bf16  (%.0f %% of 2517; /9 = %.1f, /6 = %.1f fp32-equivalent)\n", ACC, wps, ms,
                      flop / ms / 1e9, flop / ms / 1e9 / 25.17, flop / ms / 1e9 / 9, flop / ms / 1e9 / 6);
    }
  }
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  run<1>(out, e0, e1);
  run<2>(out, e0, e1);
  run<4>(out, e0, e1);
  return 0;
}
