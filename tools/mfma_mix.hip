// What next to an MFMA stream costs MFMA throughput?  The register-only loop of tools/mfma_peak.hip (99 % of peak) with F
// filler instructions of one kind issued after every v_mfma_f32_32x32x2_f32, at 3 waves per SIMD (the conv kernels'
// occupancy).  Everything is volatile inline asm, so program order is issue order.  The conv kernel's K loop carries
// about 2.3 VALU (a third of them 64-bit address arithmetic), 0.5 ds_read_b128 and 0.12 global loads per MFMA.
// build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o tools/bin/mfma_mix tools/mfma_mix.hip   (output: profiles/r01_mfma_mix.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int F>
__global__ __launch_bounds__(256) void mix_loop(float* out, int iters, float a0, float b0, const float* gsrc) {
  __shared__ f32x4 lds[1024];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  uint32_t x = threadIdx.x, y = 3;
  uint64_t x64 = threadIdx.x, y64 = 5, carry;
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  const uint32_t laddr = (uint32_t)(uintptr_t)(lds + (threadIdx.x & 1023));
  const float* gp = gsrc + threadIdx.x * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
        for (int f = 0; f < F; ++f) {
          if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
          if (KIND == 2) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x64) : "v"(y64));
          if (KIND == 3) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(x64), "=s"(carry) : "v"(x), "v"(y));
          if (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(laddr));
          if (KIND == 5) asm volatile("v_cmp_lt_i32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y), "v"(x) : "vcc");
          if (KIND == 6) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(gp));
        }
      }
      if (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (KIND == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
  float s = (float)x + (float)(x64 & 0xffff) + d[0];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int F>
void run(const char* name, float* out, const float* gsrc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000, blocks = 768;
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix_loop<KIND, F>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f, gsrc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double flop = (double)blocks * 4 * iters * 16 * (32.0 * 32 * 2 * 2);
  printf("%-34s x%d per MFMA: %6.1f TFLOP/s  (%.0f %% of 157.3)\n", name, F, flop / ms / 1e9, flop / ms / 1e9 / 1.573);
}

int main() {
  float *out, *gsrc;
  hipMalloc(&out, 768 * 256 * sizeof(float));
  hipMalloc(&gsrc, 1 << 20);
  hipMemset(gsrc, 0, 1 << 20);
  run<0, 0>("mfma only", out, gsrc);
  run<1, 1>("v_add_u32", out, gsrc);
  run<1, 2>("v_add_u32", out, gsrc);
  run<1, 4>("v_add_u32", out, gsrc);
  run<1, 8>("v_add_u32", out, gsrc);
  run<2, 1>("v_lshl_add_u64", out, gsrc);
  run<2, 2>("v_lshl_add_u64", out, gsrc);
  run<2, 4>("v_lshl_add_u64", out, gsrc);
  run<3, 1>("v_mad_u64_u32", out, gsrc);
  run<3, 2>("v_mad_u64_u32", out, gsrc);
  run<5, 1>("v_cmp + v_cndmask", out, gsrc);
  run<5, 2>("v_cmp + v_cndmask", out, gsrc);
  run<4, 1>("ds_read_b128 (wait per 4 MFMA)", out, gsrc);
  run<4, 2>("ds_read_b128 (wait per 4 MFMA)", out, gsrc);
  run<6, 1>("global_load_dwordx4 (wait per 4)", out, gsrc);
  return 0;
}
