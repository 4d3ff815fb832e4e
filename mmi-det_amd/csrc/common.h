// Shared helpers for libmmidet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <initializer_list>

#include "../../include/mmidet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void mmi_set_error(const char* fmt, ...);

#define MMI_CHECK_ARG(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      mmi_set_error(__VA_ARGS__);     \
      return MMI_ERR_ARG;             \
    }                                 \
  } while (0)

#define MMI_CHECK_LAUNCH(name)                                             \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      mmi_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return MMI_ERR_LAUNCH;                                               \
    }                                                                      \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- library-internal cross-file helpers (not part of the C ABI) ----------------------------------------------------
// bn.hip: fp64 fold of partials[part][2][C] into out0[C] (slot 0) and out1[C] (slot 1)
int mmi_pair_colsum(float* partials, int nparts, int C, float* out0, float* out1, void* stream);  // consumes partials
// cem.hip: direct VALU convolutions for the 3<->24-channel CEM layers, reached through the public conv entry points
bool mmi_smallconv_supported(const mmi_conv_desc* d);
bool mmi_smallconv_dgrad_supported(const mmi_conv_desc* d);
int mmi_smallconv_blocks(const mmi_conv_desc* d);
int mmi_smallconv_fwd(const float* x, const float* w, const float* bias, float* y, float* stat_part, const mmi_conv_desc* d,
                      hipStream_t s);
int mmi_smallconv_dgrad(const float* dy, const float* w, float* dx, const mmi_conv_desc* d, hipStream_t s);
size_t mmi_smallconv_wgrad_workspace(const mmi_conv_desc* d);
int mmi_smallconv_wgrad(const float* dy, const float* x, float* dw, void* workspace, const mmi_conv_desc* d, hipStream_t s);

// XCD-aware bijective remap of a linear workgroup id: blocks b and b+8 share an XCD (observed round-robin), so give
// each XCD a contiguous chunk of the tile sequence; neighbouring tiles (same activation rows, all output-channel
// tiles) then hit the same 4 MiB L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

// ---- counter-based dropout: keep iff hash(seed, index) >= p * 2^32 (tokens.hip kernels and the Linear epilogues) ----
__device__ __forceinline__ uint32_t mix32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, uint32_t thresh, float inv_keep) {
  return mix32(seed, idx) >= thresh ? inv_keep : 0.f;
}
static inline uint32_t drop_thresh(float p) {
  if (p <= 0.f) return 0u;
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
// GELU (erf form, nn.GELU default) and its derivative
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float v) {
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * expf(-0.5f * v * v);
  return cdf + v * pdf;
}

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + __expf(-z)); }
__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == MMI_ACT_SILU) return z / (1.0f + expf(-z));
  if (act == MMI_ACT_LEAKY) return z > 0.f ? z : 0.1f * z;
  return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == MMI_ACT_SILU) {
    const float s = 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
  }
  if (act == MMI_ACT_LEAKY) return z > 0.f ? 1.0f : 0.1f;
  return 1.0f;
}
