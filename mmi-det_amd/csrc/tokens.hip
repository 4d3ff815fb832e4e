// Token-side kernels of the cross-modal fusion transformers (GPT / GPT1_fourier): LayerNorm, GELU(erf), sigmoid,
// gate multiply, counter-based dropout, and 128-token multi-head attention held entirely in LDS.
//
// Replaces models/common.py:1147-1267 (SelfAttention, myTransformerBlock) and the elementwise glue of
// common.py:476-503, 529 of the reference.  The sequence is fixed at 2*8*8 = 128 tokens (adaptive pooling), so one
// workgroup owns one (batch, head): K/V and the 128x128 score matrix never leave the CU.  The Linear layers themselves
// run on the MFMA implicit-GEMM kernel (igemm.hip).
#include <type_traits>

#include "common.h"

namespace {

inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b));
}
#define GRID_STRIDE(e, total) \
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (total); e += (int64_t)gridDim.x * blockDim.x)

// ---- counter-based dropout (mix32 / drop_scale: common.h) ----
// out = (a [+ b]) * dropmask  (b optional, broadcast over leading dim with period bmod: positional embedding)
__global__ void dropout_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t bmod,
                               float* __restrict__ out, int64_t n, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                               uint32_t thresh, float inv_keep) {
  if (seed_dev != nullptr) seed += seed_dev[0];  // per-step device seed + per-call salt (graph replays stay random)
  GRID_STRIDE(e, n) {
    float v = a[e];
    if (b != nullptr) v += b[e % bmod];
    out[e] = thresh ? v * drop_scale(seed, (uint64_t)e, thresh, inv_keep) : v;
  }
}

__global__ void gelu_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  GRID_STRIDE(e, n) {
    y[e] = gelu_f(x[e]);
  }
}
__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
  GRID_STRIDE(e, n) {
    dx[e] = dy[e] * gelu_grad_f(x[e]);
  }
}
__global__ void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  GRID_STRIDE(e, n) y[e] = 1.0f / (1.0f + expf(-x[e]));
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
  GRID_STRIDE(e, n) {
    const float s = y[e];
    dx[e] = dy[e] * s * (1.0f - s);
  }
}
__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n) {
  GRID_STRIDE(e, n) o[e] = a[e] * b[e];
}
__global__ void scale_kernel(const float* __restrict__ a, const float* __restrict__ s, float* __restrict__ o, int64_t n) {
  const float k = s[0];
  GRID_STRIDE(e, n) o[e] = a[e] * k;
}

// ---- LayerNorm: one wave per row, two-pass in registers ----------------------------------------------------------------
template <int MAXV>  // C <= 64 * 4 * MAXV
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ stats, int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * C;
  f32x4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + c);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = v[i][k] - mean;
        q += d * d;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
  float* yr = y + (int64_t)row * C;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c), bb = *reinterpret_cast<const f32x4*>(b + c);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (v[i][k] - mean) * rstd * gg[k] + bb[k];
      *reinterpret_cast<f32x4*>(yr + c) = o;
    }
  }
}

// dx per row (wave per row) + per-block partial dgamma/dbeta (4 rows per block -> partial rows = gridDim.x)
// Optional (myTransformerBlock backward): dres = gradient arriving over the residual connection, added to dx; dmasked =
// dx * dropout mask of the branch below (seed / thresh / inv_keep), so neither a separate add nor a dropout pass runs.
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ stats, const float* __restrict__ dy,
                                                            float* __restrict__ dx, int rows, int C,
                                                            const float* __restrict__ dres, float* __restrict__ dmasked,
                                                            uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                            uint32_t thresh, float inv_keep) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (seed_dev != nullptr) seed += seed_dev[0];
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  const float* xr = x + (int64_t)row * C;
  const float* dr = dy + (int64_t)row * C;
  f32x4 xh[MAXV], dg[MAXV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c), dv = *reinterpret_cast<const f32x4*>(dr + c),
                  gg = *reinterpret_cast<const f32x4*>(g + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xh[i][k] = (xv[k] - mean) * rstd;
        dg[i][k] = dv[k] * gg[k];
        s1 += dg[i][k];
        s2 += dg[i][k] * xh[i][k];
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  const float m1 = s1 / (float)C, m2 = s2 / (float)C;
  float* dxr = dx + (int64_t)row * C;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = rstd * (dg[i][k] - m1 - xh[i][k] * m2);
      if (dres != nullptr) o += *reinterpret_cast<const f32x4*>(dres + (int64_t)row * C + c);
      *reinterpret_cast<f32x4*>(dxr + c) = o;
      if (dmasked != nullptr) {
        if (thresh) {
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] *= drop_scale(seed, (uint64_t)((int64_t)row * C + c + k), thresh, inv_keep);
        }
        *reinterpret_cast<f32x4*>(dmasked + (int64_t)row * C + c) = o;
      }
    }
  }
}

// partial column sums of dy and dy*xhat over a rows part: partials[part][2][C]
__global__ void layernorm_bwd_param_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                           const float* __restrict__ dy, float* __restrict__ partials, int rows, int C,
                                           int rows_per_part) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * rows_per_part, r1 = min(r0 + rows_per_part, rows);
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) {
      const float d = dy[(int64_t)r * C + c];
      s1 += d;
      s2 += d * (x[(int64_t)r * C + c] - stats[2 * r]) * stats[2 * r + 1];
    }
  red[0][rl][cl] = s1;
  red[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    partials[((int64_t)blockIdx.y * 2 + 0) * C + c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    partials[((int64_t)blockIdx.y * 2 + 1) * C + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
}
__global__ void pair_finalize_kernel(const float* __restrict__ partials, int nparts, int C, float* __restrict__ out0,
                                     float* __restrict__ out1) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int p = 0; p < nparts; ++p) {
    a += (double)partials[((int64_t)p * 2 + 0) * C + c];
    b += (double)partials[((int64_t)p * 2 + 1) * C + c];
  }
  out0[c] = (float)a;
  out1[c] = (float)b;
}

// ---- attention over T=128 tokens, one workgroup per (batch, head) ------------------------------------------------------
constexpr int T = 128;
constexpr int PS = T + 1;  // padded row stride of the score matrix in LDS

// Thread pair (r = t>>1, hf = t&1) owns query row r; keys/values are broadcast-read from LDS.
template <int DK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ out,
                                                       float* __restrict__ probs, int ld, int heads, float scale,
                                                       uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                       uint32_t thresh, float inv_keep) {
  __shared__ __align__(16) float sm[T * DK + T * PS];
  if (seed_dev != nullptr) seed += seed_dev[0];
  float* KV = sm;            // [T][DK]
  float* P = sm + T * DK;    // [T][PS]
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int t = threadIdx.x, r = t >> 1, hf = t & 1;
  const int64_t base = (int64_t)b * T * ld + (int64_t)h * DK;
  constexpr int DV = DK / 4;
  for (int e = t; e < T * DV; e += 256) {
    const int row = e / DV, c4 = e - row * DV;
    *reinterpret_cast<f32x4*>(KV + row * DK + c4 * 4) = *reinterpret_cast<const f32x4*>(k + base + (int64_t)row * ld + c4 * 4);
  }
  f32x4 qr[DV];
#pragma unroll
  for (int i = 0; i < DV; ++i) qr[i] = *reinterpret_cast<const f32x4*>(q + base + (int64_t)r * ld + i * 4);
  __syncthreads();
  float mx = -INFINITY;
  for (int jj = 0; jj < T / 2; ++jj) {
    const int j = hf * (T / 2) + jj;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const f32x4 kk = *reinterpret_cast<const f32x4*>(KV + j * DK + i * 4);
      s += qr[i][0] * kk[0] + qr[i][1] * kk[1] + qr[i][2] * kk[2] + qr[i][3] * kk[3];
    }
    s *= scale;
    P[r * PS + j] = s;
    mx = fmaxf(mx, s);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 1));
  float sum = 0.f;
  for (int jj = 0; jj < T / 2; ++jj) {
    const int j = hf * (T / 2) + jj;
    const float e = expf(P[r * PS + j] - mx);
    P[r * PS + j] = e;
    sum += e;
  }
  sum += __shfl_xor(sum, 1);
  const float inv = 1.0f / sum;
  float* prow = probs + ((int64_t)bh * T + r) * T;
  for (int jj = 0; jj < T / 2; ++jj) {
    const int j = hf * (T / 2) + jj;
    const float p = P[r * PS + j] * inv;
    prow[j] = p;  // saved softmax (pre-dropout) for the backward pass
    P[r * PS + j] = thresh ? p * drop_scale(seed, (uint64_t)(((int64_t)bh * T + r) * T + j), thresh, inv_keep) : p;
  }
  __syncthreads();  // all of K consumed, P complete
  for (int e = t; e < T * DV; e += 256) {
    const int row = e / DV, c4 = e - row * DV;
    *reinterpret_cast<f32x4*>(KV + row * DK + c4 * 4) = *reinterpret_cast<const f32x4*>(v + base + (int64_t)row * ld + c4 * 4);
  }
  __syncthreads();
  // out[r][hf-half of DK] = sum_j P[r][j] V[j][:]
  constexpr bool SPLIT = DK % 8 == 0;  // both threads of a pair take half a row; else (DK = 4, 20) hf=0 takes it all
  constexpr int HV = SPLIT ? (DK / 2) / 4 : 1;
  if (SPLIT || hf == 0) {
    f32x4 acc[SPLIT ? HV : DV];
#pragma unroll
    for (int i = 0; i < (SPLIT ? HV : DV); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int c0 = SPLIT ? hf * (DK / 2) : 0;
    for (int j = 0; j < T; ++j) {
      const float p = P[r * PS + j];
#pragma unroll
      for (int i = 0; i < (SPLIT ? HV : DV); ++i) acc[i] += p * *reinterpret_cast<const f32x4*>(KV + j * DK + c0 + i * 4);
    }
#pragma unroll
    for (int i = 0; i < (SPLIT ? HV : DV); ++i)
      *reinterpret_cast<f32x4*>(out + base + (int64_t)r * ld + c0 + i * 4) = acc[i];
  }
}

// Backward: four 128x128xDK products staged through one [T][DK] LDS operand buffer and the [T][PS] score buffer.
template <int DK>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const float* __restrict__ probs,
                                                       const float* __restrict__ dout, float* __restrict__ dq,
                                                       float* __restrict__ dk, float* __restrict__ dv, int ld, int heads,
                                                       float scale, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                       uint32_t thresh, float inv_keep) {
  __shared__ __align__(16) float sm[T * DK + T * PS];
  if (seed_dev != nullptr) seed += seed_dev[0];
  float* OP = sm;           // [T][DK] operand buffer
  float* S = sm + T * DK;   // [T][PS]
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int t = threadIdx.x, r = t >> 1, hf = t & 1;
  const int64_t base = (int64_t)b * T * ld + (int64_t)h * DK;
  constexpr int DV = DK / 4;
  constexpr bool SPLIT = DK % 8 == 0;
  constexpr int HV = SPLIT ? (DK / 2) / 4 : DV;
  const int c0 = SPLIT ? hf * (DK / 2) : 0;
  auto stage = [&](const float* src) {
    for (int e = t; e < T * DV; e += 256) {
      const int row = e / DV, c4 = e - row * DV;
      *reinterpret_cast<f32x4*>(OP + row * DK + c4 * 4) = *reinterpret_cast<const f32x4*>(src + base + (int64_t)row * ld + c4 * 4);
    }
  };
  const float* prow = probs + ((int64_t)bh * T + r) * T;

  // phase 1: dP'[r][j] = <dO[r], V[j]>;  dS = P * (dP - sum_j dP*P),  dP = dP' * mask/(1-p)
  stage(v);
  f32x4 dor[DV];
#pragma unroll
  for (int i = 0; i < DV; ++i) dor[i] = *reinterpret_cast<const f32x4*>(dout + base + (int64_t)r * ld + i * 4);
  __syncthreads();
  float dot = 0.f;
  for (int jj = 0; jj < T / 2; ++jj) {
    const int j = hf * (T / 2) + jj;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < DV; ++i) {
      const f32x4 vv = *reinterpret_cast<const f32x4*>(OP + j * DK + i * 4);
      s += dor[i][0] * vv[0] + dor[i][1] * vv[1] + dor[i][2] * vv[2] + dor[i][3] * vv[3];
    }
    if (thresh) s *= drop_scale(seed, (uint64_t)(((int64_t)bh * T + r) * T + j), thresh, inv_keep);
    const float p = prow[j];
    dot += s * p;
    S[r * PS + j] = s;
  }
  dot += __shfl_xor(dot, 1);
  for (int jj = 0; jj < T / 2; ++jj) {
    const int j = hf * (T / 2) + jj;
    S[r * PS + j] = prow[j] * (S[r * PS + j] - dot) * scale;  // dS, pre-multiplied by 1/sqrt(dk)
  }
  __syncthreads();

  // phase 2: dQ[r] = sum_j dS[r][j] K[j]
  stage(k);
  __syncthreads();
  if (SPLIT || hf == 0) {
    f32x4 acc[HV];
#pragma unroll
    for (int i = 0; i < HV; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < T; ++j) {
      const float s = S[r * PS + j];
#pragma unroll
      for (int i = 0; i < HV; ++i) acc[i] += s * *reinterpret_cast<const f32x4*>(OP + j * DK + c0 + i * 4);
    }
#pragma unroll
    for (int i = 0; i < HV; ++i) *reinterpret_cast<f32x4*>(dq + base + (int64_t)r * ld + c0 + i * 4) = acc[i];
  }
  __syncthreads();

  // phase 3: dK[j] = sum_r dS[r][j] Q[r]   (thread pair owns key row j = r)
  stage(q);
  __syncthreads();
  if (SPLIT || hf == 0) {
    f32x4 acc[HV];
#pragma unroll
    for (int i = 0; i < HV; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int rr = 0; rr < T; ++rr) {
      const float s = S[rr * PS + r];
#pragma unroll
      for (int i = 0; i < HV; ++i) acc[i] += s * *reinterpret_cast<const f32x4*>(OP + rr * DK + c0 + i * 4);
    }
#pragma unroll
    for (int i = 0; i < HV; ++i) *reinterpret_cast<f32x4*>(dk + base + (int64_t)r * ld + c0 + i * 4) = acc[i];
  }
  __syncthreads();

  // phase 4: dV[j] = sum_r P'[r][j] dO[r]
  for (int jj = 0; jj < T / 2; ++jj) {
    const int j = hf * (T / 2) + jj;
    const float p = prow[j];
    S[r * PS + j] = thresh ? p * drop_scale(seed, (uint64_t)(((int64_t)bh * T + r) * T + j), thresh, inv_keep) : p;
  }
  stage(dout);
  __syncthreads();
  if (SPLIT || hf == 0) {
    f32x4 acc[HV];
#pragma unroll
    for (int i = 0; i < HV; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int rr = 0; rr < T; ++rr) {
      const float s = S[rr * PS + r];
#pragma unroll
      for (int i = 0; i < HV; ++i) acc[i] += s * *reinterpret_cast<const f32x4*>(OP + rr * DK + c0 + i * 4);
    }
#pragma unroll
    for (int i = 0; i < HV; ++i) *reinterpret_cast<f32x4*>(dv + base + (int64_t)r * ld + c0 + i * 4) = acc[i];
  }
}

__global__ void seed_advance_kernel(uint64_t* seed) {
  uint64_t z = seed[0] + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  seed[0] = z ^ (z >> 31);
}


template <typename F>
int attn_dispatch(int dk, F&& f) {
  switch (dk) {
    case 4: return f(std::integral_constant<int, 4>());
    case 8: return f(std::integral_constant<int, 8>());
    case 16: return f(std::integral_constant<int, 16>());
    case 32: return f(std::integral_constant<int, 32>());
    case 64: return f(std::integral_constant<int, 64>());
    case 128: return f(std::integral_constant<int, 128>());
    case 20: return f(std::integral_constant<int, 20>());    // yolov5x widths 160/320/640/1280 over 8 heads
    case 40: return f(std::integral_constant<int, 40>());
    case 80: return f(std::integral_constant<int, 80>());
    case 160: return f(std::integral_constant<int, 160>());
    default: return -1;
  }
}

}  // namespace

extern "C" int mmi_dropout(const float* a, const float* b, int64_t bmod, float* out, int64_t n, float p, uint64_t seed,
                           const uint64_t* seed_dev, void* stream) {
  MMI_CHECK_ARG(a && out && n > 0 && p >= 0.f && p < 1.f && (!b || bmod > 0), "mmi_dropout: bad arguments");
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, bmod, out, n, seed,
                     seed_dev, drop_thresh(p), 1.0f / (1.0f - p));
  MMI_CHECK_LAUNCH("mmi_dropout");
  return MMI_OK;
}

#define EW_UNARY(NAME, KERNEL)                                                                              \
  extern "C" int NAME(const float* x, float* y, int64_t n, void* stream) {                                  \
    MMI_CHECK_ARG(x && y && n > 0, #NAME ": bad arguments");                                                \
    hipLaunchKernelGGL(KERNEL, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);             \
    MMI_CHECK_LAUNCH(#NAME);                                                                                \
    return MMI_OK;                                                                                          \
  }
#define EW_BINARY(NAME, KERNEL)                                                                             \
  extern "C" int NAME(const float* a, const float* b, float* o, int64_t n, void* stream) {                  \
    MMI_CHECK_ARG(a && b && o && n > 0, #NAME ": bad arguments");                                           \
    hipLaunchKernelGGL(KERNEL, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, o, n);          \
    MMI_CHECK_LAUNCH(#NAME);                                                                                \
    return MMI_OK;                                                                                          \
  }
EW_UNARY(mmi_gelu_fwd, gelu_kernel)
EW_BINARY(mmi_gelu_bwd, gelu_bwd_kernel)
EW_UNARY(mmi_sigmoid_fwd, sigmoid_kernel)
EW_BINARY(mmi_sigmoid_bwd, sigmoid_bwd_kernel)
EW_BINARY(mmi_mul, mul_kernel)
EW_BINARY(mmi_scale, scale_kernel)

extern "C" int mmi_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, int rows,
                                 int C, float eps, void* stream) {
  MMI_CHECK_ARG(x && gamma && beta && y && stats && rows > 0 && C > 0, "mmi_layernorm_fwd: bad arguments");
  MMI_CHECK_ARG(C % 4 == 0 && C <= 4096, "mmi_layernorm_fwd: C=%d must be a multiple of 4 and <= 4096", C);
  const dim3 grid(cdiv(rows, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (C <= 256) hipLaunchKernelGGL(layernorm_fwd_kernel<1>, grid, block, 0, s, x, gamma, beta, y, stats, rows, C, eps);
  else if (C <= 1024) hipLaunchKernelGGL(layernorm_fwd_kernel<4>, grid, block, 0, s, x, gamma, beta, y, stats, rows, C, eps);
  else hipLaunchKernelGGL(layernorm_fwd_kernel<16>, grid, block, 0, s, x, gamma, beta, y, stats, rows, C, eps);
  MMI_CHECK_LAUNCH("mmi_layernorm_fwd");
  return MMI_OK;
}

extern "C" int mmi_layernorm_bwd_parts(int rows) {
  const int p = (rows + 63) / 64;
  return p > 256 ? 256 : (p < 1 ? 1 : p);
}

namespace {
int ln_bwd_dx(const float* x, const float* gamma, const float* stats, const float* dy, float* dx, int rows, int C,
              const float* dres, float* dmasked, float p, uint64_t seed, const uint64_t* seed_dev, hipStream_t s) {
  const dim3 grid(cdiv(rows, 4)), block(256);
  const uint32_t th = drop_thresh(p);
  const float ik = 1.0f / (1.0f - p);
  if (C <= 256) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, block, 0, s, x, gamma, stats, dy, dx, rows, C, dres, dmasked, seed, seed_dev, th, ik);
  else if (C <= 1024) hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, block, 0, s, x, gamma, stats, dy, dx, rows, C, dres, dmasked, seed, seed_dev, th, ik);
  else hipLaunchKernelGGL(layernorm_bwd_kernel<16>, grid, block, 0, s, x, gamma, stats, dy, dx, rows, C, dres, dmasked, seed, seed_dev, th, ik);
  MMI_CHECK_LAUNCH("mmi_layernorm_bwd");
  return MMI_OK;
}
int ln_bwd_params(const float* x, const float* stats, const float* dy, float* partials, float* dgamma, float* dbeta, int rows,
                  int C, hipStream_t s) {
  const int nparts = mmi_layernorm_bwd_parts(rows);
  const int rpp = (rows + nparts - 1) / nparts;
  hipLaunchKernelGGL(layernorm_bwd_param_kernel, dim3(cdiv(C, 64), nparts), dim3(256), 0, s, x, stats, dy, partials, rows, C, rpp);
  MMI_CHECK_LAUNCH("mmi_layernorm_bwd(param)");
  hipLaunchKernelGGL(pair_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, partials, nparts, C, dbeta, dgamma);
  MMI_CHECK_LAUNCH("mmi_layernorm_bwd(finalize)");
  return MMI_OK;
}
}  // namespace

extern "C" int mmi_layernorm_bwd(const float* x, const float* gamma, const float* stats, const float* dy, float* dx,
                                 float* partials, float* dgamma, float* dbeta, int rows, int C, void* stream) {
  MMI_CHECK_ARG(x && gamma && stats && dy && dx && partials && dgamma && dbeta && rows > 0, "mmi_layernorm_bwd: bad arguments");
  MMI_CHECK_ARG(C % 4 == 0 && C <= 4096, "mmi_layernorm_bwd: C=%d must be a multiple of 4 and <= 4096", C);
  if (int e = ln_bwd_dx(x, gamma, stats, dy, dx, rows, C, nullptr, nullptr, 0.f, 0, nullptr, (hipStream_t)stream)) return e;
  return ln_bwd_params(x, stats, dy, partials, dgamma, dbeta, rows, C, (hipStream_t)stream);
}

extern "C" int mmi_layernorm_bwd_input(const float* x, const float* gamma, const float* stats, const float* dy,
                                       const float* dresidual, float* dx, float* dx_dropped, float p_drop, uint64_t seed,
                                       const uint64_t* seed_dev, int rows, int C, void* stream) {
  MMI_CHECK_ARG(x && gamma && stats && dy && dx && rows > 0, "mmi_layernorm_bwd_input: bad arguments");
  MMI_CHECK_ARG(C % 4 == 0 && C <= 4096, "mmi_layernorm_bwd_input: C=%d must be a multiple of 4 and <= 4096", C);
  MMI_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed_dev != nullptr || seed != 0), "mmi_layernorm_bwd_input: bad dropout arguments");
  return ln_bwd_dx(x, gamma, stats, dy, dx, rows, C, dresidual, dx_dropped, p_drop, seed, seed_dev, (hipStream_t)stream);
}

extern "C" int mmi_layernorm_bwd_params(const float* x, const float* stats, const float* dy, float* partials, float* dgamma,
                                        float* dbeta, int rows, int C, void* stream) {
  MMI_CHECK_ARG(x && stats && dy && partials && dgamma && dbeta && rows > 0 && C > 0, "mmi_layernorm_bwd_params: bad arguments");
  return ln_bwd_params(x, stats, dy, partials, dgamma, dbeta, rows, C, (hipStream_t)stream);
}

extern "C" int mmi_attention_fwd(const float* q, const float* k, const float* v, float* out, float* probs, int B,
                                 int heads, int dk, int ld, float p_drop, uint64_t seed, const uint64_t* seed_dev,
                                 void* stream) {
  MMI_CHECK_ARG(q && k && v && out && probs && B > 0 && heads > 0 && ld >= heads * dk && ld % 4 == 0, "mmi_attention_fwd: bad arguments");
  const float scale = 1.0f / sqrtf((float)dk);
  const uint32_t th = drop_thresh(p_drop);
  const float ik = 1.0f / (1.0f - p_drop);
  hipStream_t s = (hipStream_t)stream;
  const int rc = attn_dispatch(dk, [&](auto DKc) {
    constexpr int DK = decltype(DKc)::value;
    hipLaunchKernelGGL(attn_fwd_kernel<DK>, dim3(B * heads), dim3(256), 0, s, q, k, v, out, probs, ld, heads, scale, seed, seed_dev, th, ik);
    return 0;
  });
  MMI_CHECK_ARG(rc == 0, "mmi_attention_fwd: head dim %d unsupported (4,8,16,20,32,40,64,80,128,160)", dk);
  MMI_CHECK_LAUNCH("mmi_attention_fwd");
  return MMI_OK;
}

extern "C" int mmi_attention_bwd(const float* q, const float* k, const float* v, const float* probs, const float* dout,
                                 float* dq, float* dk_, float* dv, int B, int heads, int dk, int ld, float p_drop,
                                 uint64_t seed, const uint64_t* seed_dev, void* stream) {
  MMI_CHECK_ARG(q && k && v && probs && dout && dq && dk_ && dv && B > 0 && heads > 0 && ld >= heads * dk && ld % 4 == 0,
                "mmi_attention_bwd: bad arguments");
  const float scale = 1.0f / sqrtf((float)dk);
  const uint32_t th = drop_thresh(p_drop);
  const float ik = 1.0f / (1.0f - p_drop);
  hipStream_t s = (hipStream_t)stream;
  const int rc = attn_dispatch(dk, [&](auto DKc) {
    constexpr int DK = decltype(DKc)::value;
    hipLaunchKernelGGL(attn_bwd_kernel<DK>, dim3(B * heads), dim3(256), 0, s, q, k, v, probs, dout, dq, dk_, dv, ld, heads, scale,
                       seed, seed_dev, th, ik);
    return 0;
  });
  MMI_CHECK_ARG(rc == 0, "mmi_attention_bwd: head dim %d unsupported (4,8,16,20,32,40,64,80,128,160)", dk);
  MMI_CHECK_LAUNCH("mmi_attention_bwd");
  return MMI_OK;
}

extern "C" int mmi_seed_advance(uint64_t* seed_dev, void* stream) {
  MMI_CHECK_ARG(seed_dev != nullptr, "mmi_seed_advance: null pointer");
  hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, seed_dev);
  MMI_CHECK_LAUNCH("mmi_seed_advance");
  return MMI_OK;
}
