// Sustained fp32-MFMA issue rate of the chip: a register-only v_mfma_f32_32x32x2_f32 loop at 1..4 waves per SIMD.
// Prices the 157.3 TFLOP/s datasheet peak (2.4 GHz x 256 CU x 256 FLOP/clk) against what the clocks sustain under load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[ACC];
  for (int i = 0; i < ACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < ACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; ++wps) {          // waves per SIMD = workgroups (4 waves) per CU
    const int blocks = 256 * wps;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)blocks * 4 * iters * 4 * 4 * (32.0 * 32 * 2 * 2);
      if (rep) printf("%d wave(s)/SIMD: %.1f ms  %.1f TFLOP/s  (%.0f %% of 157.3)\n", wps, ms, flop / ms / 1e9, flop / ms / 1e9 / 1.573);
    }
  }
  // long run: does the rate sag as the chip heats up?
  for (int k = 0; k < 5; ++k) {
    hipEventRecord(e0);
    for (int j = 0; j < 10; ++j) hipLaunchKernelGGL(mfma_loop<4>, dim3(768), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 10.0 * 768 * 4 * iters * 4 * 4 * (32.0 * 32 * 2 * 2);
    printf("sustained %d: %.1f TFLOP/s\n", k, flop / ms / 1e9);
  }
  return 0;
}
