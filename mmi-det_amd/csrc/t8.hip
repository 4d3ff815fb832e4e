// Pre-split ("T8") images of fp32 tensors: the stand-alone converter.  The producers that matter write their image themselves
// (optim.hip: weights after the SGD update; bn.hip: activations out of the normalise pass, gradients out of the BatchNorm
// backward); this kernel covers every other tensor a three-term GEMM wants to read pre-split, and the first image of the weights.
#include "t8.h"

namespace {
__global__ __launch_bounds__(256) void split_t8_kernel(const float* __restrict__ src, int ld, void* __restrict__ dst, int ld8, int64_t rows,
                                                       int groups) {
  const int64_t n = rows * groups;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / groups;
    const int g = (int)(i - r * groups);
    const float* s = src + r * ld + 8 * g;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(s), hi = *reinterpret_cast<const f32x4*>(s + 4);
    t8_bf16x8 t[3];
    t8_split8(lo, hi, t);
    t8_store8(dst, r * ld8 + 8 * g, t);
  }
}
}  // namespace

extern "C" int mmi_split_t8(const float* src, int ld, void* dst, int ld8, int64_t rows, int C, void* stream) {
  MMI_CHECK_ARG(src && dst && rows >= 0 && C > 0, "mmi_split_t8: bad arguments");
  MMI_CHECK_ARG(C % 8 == 0 && ld % 4 == 0 && ld8 % 8 == 0 && ld >= C && ld8 >= C && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0,
                "mmi_split_t8: channel count and image row stride must be multiples of 8, the source 16-byte aligned");
  if (rows == 0) return MMI_OK;
  const int groups = C / 8;
  const int64_t n = rows * groups;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_t8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ld, dst, ld8, rows, groups);
  MMI_CHECK_LAUNCH("mmi_split_t8");
  return MMI_OK;
}
