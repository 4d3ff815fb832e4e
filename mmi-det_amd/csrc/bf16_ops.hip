// bf16 storage: the memory-bound glue ops of the graph on bf16 NHWC activations, plus the fp32 <-> bf16
// boundary casts.  Arithmetic in fp32, 8-byte lane accesses (four bf16) along the channel axis.  These mirror
// mmi_add / mmi_copy2d / mmi_upsample2x(_bwd) of elementwise.hip (Add common.py:914-921, Concat 740-748, nn.Upsample of the
// YAML head); SPP, Focus' space-to-depth, the CEM and the token-side fusion ops take the fp32 kernels behind a cast for now.
#include "common.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b));
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const __bf16* p) {
  const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(__bf16* p, f32x4 v) {
  *reinterpret_cast<bf16x4*>(p) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}

// out[r, :C] = a[r, :C] (+ b[r, :C]); rows x C with row strides; V = 4 (C, strides multiples of 4) or 1
template <typename TI, typename TO, int V, bool ADD>
__global__ void rows_kernel(const TI* __restrict__ a, int lda, const TI* __restrict__ b, int ldb, TO* __restrict__ out, int ldo,
                            int64_t rows, int C) {
  const int cv = C / V;
  GRID_STRIDE(e, rows * cv) {
    const int64_t r = e / cv;
    const int c = (int)(e - r * cv) * V;
    if (V == 4) {
      f32x4 v = ld4(a + r * lda + c);
      if (ADD) v += ld4(b + r * ldb + c);
      st4(out + r * ldo + c, v);
    } else {
      float v = (float)a[r * lda + c];
      if (ADD) v += (float)b[r * ldb + c];
      out[r * ldo + c] = (TO)v;
    }
  }
}

template <int V>
__global__ void upsample2x_bf16_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, int N, int H, int W, int C) {
  const int cv = C / V, Ho = 2 * H, Wo = 2 * W;
  GRID_STRIDE(e, (int64_t)N * Ho * Wo * cv) {
    const int c = (int)(e % cv) * V;
    int64_t t = e / cv;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho), n = (int)(t / Ho);
    const __bf16* src = x + (((int64_t)n * H + (oh >> 1)) * W + (ow >> 1)) * C + c;
    __bf16* dst = y + (((int64_t)n * Ho + oh) * Wo + ow) * C + c;
    if (V == 4) *reinterpret_cast<bf16x4*>(dst) = *reinterpret_cast<const bf16x4*>(src);
    else dst[0] = src[0];
  }
}

template <int V>
__global__ void upsample2x_bwd_bf16_kernel(const __bf16* __restrict__ dy, __bf16* __restrict__ dx, int N, int H, int W, int C) {
  const int cv = C / V, Ho = 2 * H, Wo = 2 * W;
  GRID_STRIDE(e, (int64_t)N * H * W * cv) {
    const int c = (int)(e % cv) * V;
    int64_t t = e / cv;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H), n = (int)(t / H);
    const __bf16* s00 = dy + (((int64_t)n * Ho + 2 * h) * Wo + 2 * w) * C + c;
    const __bf16* s10 = s00 + (int64_t)Wo * C;
    __bf16* dst = dx + (((int64_t)n * H + h) * W + w) * C + c;
    if (V == 4) st4(dst, (ld4(s00) + ld4(s00 + C)) + (ld4(s10) + ld4(s10 + C)));
    else dst[0] = (__bf16)(((float)s00[0] + (float)s00[C]) + ((float)s10[0] + (float)s10[C]));
  }
}

inline bool vec4ok(int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs, uintptr_t mask) {
  if (C % 4) return false;
  for (int l : lds)
    if (l % 4) return false;
  for (const void* p : ptrs)
    if (p && ((uintptr_t)p & mask)) return false;
  return true;
}

template <typename TI, typename TO, bool ADD>
int launch_rows(const TI* a, int lda, const TI* b, int ldb, TO* out, int ldo, int64_t rows, int C, void* stream, const char* who) {
  MMI_CHECK_ARG(a && out && (!ADD || b) && rows > 0 && C > 0 && lda >= C && ldo >= C && (!ADD || ldb >= C), "%s: bad arguments", who);
  const bool v = vec4ok(C, {lda, ldo, ADD ? ldb : 0}, {a, ADD ? (const void*)b : nullptr}, 4 * sizeof(TI) - 1) &&
                 vec4ok(C, {}, {out}, 4 * sizeof(TO) - 1);
  hipStream_t s = (hipStream_t)stream;
  if (v) hipLaunchKernelGGL((rows_kernel<TI, TO, 4, ADD>), dim3(ew_blocks(rows * C / 4)), dim3(256), 0, s, a, lda, b, ldb, out, ldo, rows, C);
  else hipLaunchKernelGGL((rows_kernel<TI, TO, 1, ADD>), dim3(ew_blocks(rows * C)), dim3(256), 0, s, a, lda, b, ldb, out, ldo, rows, C);
  MMI_CHECK_LAUNCH(who);
  return MMI_OK;
}

}  // namespace

// fp32 (rows x C, row stride ldi) -> bf16 (row stride ldo), round to nearest even; and back (exact)
extern "C" int mmi_cast_f32_bf16(const float* in, int ldi, void* out, int ldo, int64_t rows, int C, void* stream) {
  return launch_rows<float, __bf16, false>(in, ldi, nullptr, 0, (__bf16*)out, ldo, rows, C, stream, "mmi_cast_f32_bf16");
}
extern "C" int mmi_cast_bf16_f32(const void* in, int ldi, float* out, int ldo, int64_t rows, int C, void* stream) {
  return launch_rows<__bf16, float, false>((const __bf16*)in, ldi, nullptr, 0, out, ldo, rows, C, stream, "mmi_cast_bf16_f32");
}
extern "C" int mmi_add_bf16(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int64_t rows, int C, void* stream) {
  return launch_rows<__bf16, __bf16, true>((const __bf16*)a, lda, (const __bf16*)b, ldb, (__bf16*)out, ldo, rows, C, stream, "mmi_add_bf16");
}
extern "C" int mmi_copy2d_bf16(const void* in, int ldi, void* out, int ldo, int64_t rows, int C, void* stream) {
  return launch_rows<__bf16, __bf16, false>((const __bf16*)in, ldi, nullptr, 0, (__bf16*)out, ldo, rows, C, stream, "mmi_copy2d_bf16");
}

extern "C" int mmi_upsample2x_bf16(const void* x, void* y, int N, int H, int W, int C, void* stream) {
  MMI_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0, "mmi_upsample2x_bf16: bad arguments");
  if (vec4ok(C, {}, {x, y}, 7))
    hipLaunchKernelGGL(upsample2x_bf16_kernel<4>, dim3(ew_blocks((int64_t)N * H * W * C)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)x, (__bf16*)y, N, H, W, C);
  else
    hipLaunchKernelGGL(upsample2x_bf16_kernel<1>, dim3(ew_blocks((int64_t)N * H * W * C * 4)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)x, (__bf16*)y, N, H, W, C);
  MMI_CHECK_LAUNCH("mmi_upsample2x_bf16");
  return MMI_OK;
}

extern "C" int mmi_upsample2x_bwd_bf16(const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
  MMI_CHECK_ARG(dy && dx && N > 0 && C > 0 && H > 0 && W > 0, "mmi_upsample2x_bwd_bf16: bad arguments");
  if (vec4ok(C, {}, {dy, dx}, 7))
    hipLaunchKernelGGL(upsample2x_bwd_bf16_kernel<4>, dim3(ew_blocks((int64_t)N * H * W * C / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)dy, (__bf16*)dx, N, H, W, C);
  else
    hipLaunchKernelGGL(upsample2x_bwd_bf16_kernel<1>, dim3(ew_blocks((int64_t)N * H * W * C)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)dy, (__bf16*)dx, N, H, W, C);
  MMI_CHECK_LAUNCH("mmi_upsample2x_bwd_bf16");
  return MMI_OK;
}
