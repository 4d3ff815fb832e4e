// "T8": the pre-split image of an fp32 tensor for the three-term bf16 GEMM modes (igemm_kernel<..., T8>, wgrad_kernel<..., T8>).
// Per 8 consecutive channels of a row 48 bytes: [term 0: 8 bf16 | term 1: 8 bf16 | term 2: 8 bf16], term k the bf16 rounding of
// what the previous terms left of the value -- exactly the terms the GEMM kernels' own split produces when they stage a tile
// (igemm_kernel.h::split_bf16), so a GEMM on T8 operands is bit-identical to the same GEMM splitting in the kernel.  The image
// keeps the tensor's element indexing: element (row, c) of a tensor with row stride ld lives in the group at byte
// (row * ld + (c & ~7)) * 6, so channel slices (offsets and strides multiples of 8) and twin lanes address it like the fp32 tensor.
#pragma once
#include "common.h"

typedef __bf16 t8_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 t8_bf16x2 __attribute__((ext_vector_type(2)));

// three terms of eight values (two float4): same operations, in the same order, as split_bf16<3> on element pairs
__device__ __forceinline__ void t8_split8(const f32x4& lo, const f32x4& hi, t8_bf16x8 (&t)[3]) {
  f32x2 r[4] = {{lo[0], lo[1]}, {lo[2], lo[3]}, {hi[0], hi[1]}, {hi[2], hi[3]}};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const t8_bf16x2 h = __builtin_convertvector(r[q], t8_bf16x2);
      t[k][2 * q] = h[0];
      t[k][2 * q + 1] = h[1];
      if (k + 1 < 3) r[q] -= __builtin_convertvector(h, f32x2);
    }
  }
}

// store the group of eight channels starting at element index e (a multiple of 8) of the image `dst`
__device__ __forceinline__ void t8_store8(void* dst, int64_t e, const t8_bf16x8 (&t)[3]) {
  t8_bf16x8* g = reinterpret_cast<t8_bf16x8*>(reinterpret_cast<char*>(dst) + e * 6);
  g[0] = t[0];
  g[1] = t[1];
  g[2] = t[2];
}
