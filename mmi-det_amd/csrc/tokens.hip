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
                                                       float* __restrict__ probs, int ld, int ldo, int heads, float scale,
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
      *reinterpret_cast<f32x4*>(out + (int64_t)b * T * ldo + (int64_t)h * DK + (int64_t)r * ldo + c0 + i * 4) = acc[i];
  }
}

// Backward: four 128x128xDK products staged through one [T][DK] LDS operand buffer and the [T][PS] score buffer.
template <int DK>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const float* __restrict__ probs,
                                                       const float* __restrict__ dout, float* __restrict__ dq,
                                                       float* __restrict__ dk, float* __restrict__ dv, int ld, int ldo, int heads,
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
  const int64_t base_o = (int64_t)b * T * ldo + (int64_t)h * DK;  // dout has its own row stride
  auto stage = [&](const float* src, int64_t sbase, int sld) {
    for (int e = t; e < T * DV; e += 256) {
      const int row = e / DV, c4 = e - row * DV;
      *reinterpret_cast<f32x4*>(OP + row * DK + c4 * 4) = *reinterpret_cast<const f32x4*>(src + sbase + (int64_t)row * sld + c4 * 4);
    }
  };
  const float* prow = probs + ((int64_t)bh * T + r) * T;

  // phase 1: dP'[r][j] = <dO[r], V[j]>;  dS = P * (dP - sum_j dP*P),  dP = dP' * mask/(1-p)
  stage(v, base, ld);
  f32x4 dor[DV];
#pragma unroll
  for (int i = 0; i < DV; ++i) dor[i] = *reinterpret_cast<const f32x4*>(dout + base_o + (int64_t)r * ldo + i * 4);
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
  stage(k, base, ld);
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
  stage(q, base, ld);
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
  stage(dout, base_o, ldo);
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

// ---- MFMA attention (head dims that are multiples of 8) ---------------------------------------------------------------
// The same five 128 x 128 x DK products on v_mfma_f32_32x32x2_f32 (exact fp32 products, as everywhere on this path).  One
// workgroup per (batch, head), wave w owns query rows (or, for dK / dV, key rows) [32w, 32w+32).  Operand conventions of the
// instruction: A lane l holds A[m = l&31][k = l>>5], B lane l holds B[k = l>>5][n = l&31]; a lane that reads four
// consecutive k of its row as one f32x4 at k0 = 8g + 4(l>>5) feeds four MFMAs (element e contracts k0 + e on both sides).
//   row-major operands ("[row][k]", contraction along the row: Q, K in QK^T; dO, V in dO V^T; P, dS along the keys) are read
//   as ds_read_b128 from a buffer whose row stride is DKP+4 (or T+4) floats = an odd number of 16-byte groups;
//   column operands ("[k][col]": V in PV, K in dS K, Q in dS^T Q, dO in P^T dO, and the transposed dS / P) as ds_read_b32
//   of 32 consecutive floats.
// The A operand of the first product (Q, dO) never touches LDS: each lane loads its own fragments from global memory.
constexpr int PSM = T + 4;

__device__ __forceinline__ int mfma_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }  // C/D layout of 32x32
__device__ __forceinline__ float half_wave_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int DK>
struct AttnGeom {
  static constexpr int DKP = (DK + 31) / 32 * 32;  // columns of the [T][DK] operand tile, zero padded to whole MFMA tiles
  static constexpr int RS = DKP + 4;               // its row stride in floats
  static constexpr int NG = DK / 8;                // 8-k groups along the head dimension
  static constexpr int CT = DKP / 32;              // 32-column tiles of a [T][DK] result
};

// [T][DK] rows of one (batch, head) -> OP[row][RS], columns DK..DKP zeroed
template <int DK>
__device__ __forceinline__ void attn_stage(float* OP, const float* __restrict__ src, int64_t base, int ld, int t) {
  using G = AttnGeom<DK>;
  constexpr int V4 = G::DKP / 4;
  for (int e = t; e < T * V4; e += 256) {
    const int row = e / V4, c4 = e - row * V4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c4 * 4 < DK) v = *reinterpret_cast<const f32x4*>(src + base + (int64_t)row * ld + c4 * 4);
    *reinterpret_cast<f32x4*>(OP + row * G::RS + c4 * 4) = v;
  }
}

// acc[jt] (+)= A_frag . rowmajor(OP)^T : 32 rows of this wave x 128 columns (rows of OP), contraction over DK
template <int DK>
__device__ __forceinline__ void attn_rows_times_rowsT(const f32x4 (&af)[AttnGeom<DK>::NG], const float* OP, int l31, int lh,
                                                      f32x16 (&acc)[4]) {
  using G = AttnGeom<DK>;
#pragma unroll
  for (int g = 0; g < G::NG; ++g) {
    f32x4 b[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) b[jt] = *reinterpret_cast<const f32x4*>(OP + (jt * 32 + l31) * G::RS + g * 8 + lh * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][e], b[jt][e], acc[jt], 0, 0, 0);
  }
}

// o[ct] = S-operand x OP : 32 result rows of this wave x DKP columns, contraction over the 128 rows of OP.
// TRANS = false: A[m][k] = S[(32w + m) * PSM + k]  (P V, dS K);  TRANS = true: A[m][k] = S[k * PSM + 32w + m]  (dS^T Q, P^T dO)
template <int DK, bool TRANS>
__device__ __forceinline__ void attn_scores_times_cols(const float* S, const float* OP, int w, int l31, int lh,
                                                       f32x16 (&o)[AttnGeom<DK>::CT]) {
  using G = AttnGeom<DK>;
#pragma unroll
  for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
#pragma unroll 4
  for (int g = 0; g < T / 8; ++g) {
    f32x4 a;
    if (!TRANS) {
      a = *reinterpret_cast<const f32x4*>(S + (32 * w + l31) * PSM + g * 8 + lh * 4);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = S[(g * 8 + lh * 4 + e) * PSM + 32 * w + l31];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* brow = OP + (g * 8 + lh * 4 + e) * G::RS + l31;
#pragma unroll
      for (int ct = 0; ct < G::CT; ++ct) o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], brow[ct * 32], o[ct], 0, 0, 0);
    }
  }
}

template <int DK>
__device__ __forceinline__ void attn_store_rows(float* __restrict__ dst, int64_t base, int ld, int w, int l31, int lh,
                                                const f32x16 (&o)[AttnGeom<DK>::CT]) {
#pragma unroll
  for (int ct = 0; ct < AttnGeom<DK>::CT; ++ct) {
    const int col = ct * 32 + l31;
    if (col < DK) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[base + (int64_t)(32 * w + mfma_row(r, lh)) * ld + col] = o[ct][r];
    }
  }
}

template <int DK>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ out,
                                                            float* __restrict__ probs, int ld, int ldo, int heads, float scale,
                                                            uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                            uint32_t thresh, float inv_keep) {
  using G = AttnGeom<DK>;
  __shared__ __align__(16) float sm[T * G::RS + T * PSM];
  if (seed_dev != nullptr) seed += seed_dev[0];
  float* OP = sm;
  float* P = sm + T * G::RS;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, lh = lane >> 5;
  const int64_t base = (int64_t)b * T * ld + (int64_t)h * DK;
  f32x4 qa[G::NG];
#pragma unroll
  for (int g = 0; g < G::NG; ++g) qa[g] = *reinterpret_cast<const f32x4*>(q + base + (int64_t)(32 * w + l31) * ld + g * 8 + lh * 4);
  attn_stage<DK>(OP, k, base, ld, t);
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
  attn_rows_times_rowsT<DK>(qa, OP, l31, lh, acc);
  // softmax over the 128 keys of a row: 4 column tiles in this lane x the 32 lanes of its half wave
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 32 * w + mfma_row(r, lh);
    float sv[4];
    float m = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      sv[jt] = acc[jt][r] * scale;
      m = fmaxf(m, sv[jt]);
    }
    m = half_wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      sv[jt] = expf(sv[jt] - m);
      sum += sv[jt];
    }
    const float inv = 1.0f / half_wave_sum(sum);
    float* prow = probs + ((int64_t)bh * T + row) * T;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int j = jt * 32 + l31;
      const float pv = sv[jt] * inv;
      prow[j] = pv;  // saved softmax (pre-dropout) for the backward pass
      P[row * PSM + j] = thresh ? pv * drop_scale(seed, (uint64_t)(((int64_t)bh * T + row) * T + j), thresh, inv_keep) : pv;
    }
  }
  __syncthreads();  // every wave is done with K; P complete
  attn_stage<DK>(OP, v, base, ld, t);
  __syncthreads();
  f32x16 o[G::CT];
  attn_scores_times_cols<DK, false>(P, OP, w, l31, lh, o);
  attn_store_rows<DK>(out, (int64_t)b * T * ldo + (int64_t)h * DK, ldo, w, l31, lh, o);
}

template <int DK>
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ probs,
                                                            const float* __restrict__ dout, float* __restrict__ dq,
                                                            float* __restrict__ dk, float* __restrict__ dv, int ld, int ldo, int heads,
                                                            float scale, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                            uint32_t thresh, float inv_keep) {
  using G = AttnGeom<DK>;
  __shared__ __align__(16) float sm[T * G::RS + T * PSM];
  if (seed_dev != nullptr) seed += seed_dev[0];
  float* OP = sm;
  float* S = sm + T * G::RS;
  const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, lh = lane >> 5;
  const int64_t base = (int64_t)b * T * ld + (int64_t)h * DK;
  const int64_t base_o = (int64_t)b * T * ldo + (int64_t)h * DK;  // dout has its own row stride

  // phase 1: dP'[r][j] = <dO[r], V[j]>;  dS = P * (dP - sum_j dP*P) / sqrt(dk),  dP = dP' * mask/(1-p)
  {
    f32x4 da[G::NG];
#pragma unroll
    for (int g = 0; g < G::NG; ++g)
      da[g] = *reinterpret_cast<const f32x4*>(dout + base_o + (int64_t)(32 * w + l31) * ldo + g * 8 + lh * 4);
    attn_stage<DK>(OP, v, base, ld, t);
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
    attn_rows_times_rowsT<DK>(da, OP, l31, lh, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * w + mfma_row(r, lh);
      const float* prow = probs + ((int64_t)bh * T + row) * T;
      float pv[4], dp[4];
      float dot = 0.f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int j = jt * 32 + l31;
        pv[jt] = prow[j];
        dp[jt] = acc[jt][r];
        if (thresh) dp[jt] *= drop_scale(seed, (uint64_t)(((int64_t)bh * T + row) * T + j), thresh, inv_keep);
        dot += dp[jt] * pv[jt];
      }
      dot = half_wave_sum(dot);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) S[row * PSM + jt * 32 + l31] = pv[jt] * (dp[jt] - dot) * scale;
    }
  }
  __syncthreads();  // V consumed, dS complete (phase 3 reads every row of it)

  f32x16 o[G::CT];
  // phase 2: dQ[r] = sum_j dS[r][j] K[j]
  attn_stage<DK>(OP, k, base, ld, t);
  __syncthreads();
  attn_scores_times_cols<DK, false>(S, OP, w, l31, lh, o);
  attn_store_rows<DK>(dq, base, ld, w, l31, lh, o);
  __syncthreads();

  // phase 3: dK[j] = sum_r dS[r][j] Q[r]   (wave w owns keys 32w..32w+31)
  attn_stage<DK>(OP, q, base, ld, t);
  __syncthreads();
  attn_scores_times_cols<DK, true>(S, OP, w, l31, lh, o);
  attn_store_rows<DK>(dk, base, ld, w, l31, lh, o);
  __syncthreads();  // dS and Q consumed

  // phase 4: dV[j] = sum_r P'[r][j] dO[r],  P' = P * mask/(1-p)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 32 * w + mfma_row(r, lh);
    const float* prow = probs + ((int64_t)bh * T + row) * T;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const int j = jt * 32 + l31;
      const float pv = prow[j];
      S[row * PSM + j] = thresh ? pv * drop_scale(seed, (uint64_t)(((int64_t)bh * T + row) * T + j), thresh, inv_keep) : pv;
    }
  }
  attn_stage<DK>(OP, dout, base_o, ldo, t);
  __syncthreads();
  attn_scores_times_cols<DK, true>(S, OP, w, l31, lh, o);
  attn_store_rows<DK>(dv, base, ld, w, l31, lh, o);
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
  return mmi_attention_fwd_strided(q, k, v, out, probs, B, heads, dk, ld, ld, p_drop, seed, seed_dev, stream);
}

extern "C" int mmi_attention_fwd_strided(const float* q, const float* k, const float* v, float* out, float* probs, int B,
                                         int heads, int dk, int ld, int ldo, float p_drop, uint64_t seed,
                                         const uint64_t* seed_dev, void* stream) {
  MMI_CHECK_ARG(q && k && v && out && probs && B > 0 && heads > 0 && ld >= heads * dk && ld % 4 == 0 && ldo >= heads * dk && ldo % 4 == 0,
                "mmi_attention_fwd: bad arguments");
  MMI_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0, "mmi_attention_fwd: operands must be 16-byte aligned");
  const float scale = 1.0f / sqrtf((float)dk);
  const uint32_t th = drop_thresh(p_drop);
  const float ik = 1.0f / (1.0f - p_drop);
  hipStream_t s = (hipStream_t)stream;
  const int rc = attn_dispatch(dk, [&](auto DKc) {
    constexpr int DK = decltype(DKc)::value;
    if constexpr (DK % 8 == 0)
      hipLaunchKernelGGL(attn_fwd_mfma_kernel<DK>, dim3(B * heads), dim3(256), 0, s, q, k, v, out, probs, ld, ldo, heads, scale, seed, seed_dev, th, ik);
    else
      hipLaunchKernelGGL(attn_fwd_kernel<DK>, dim3(B * heads), dim3(256), 0, s, q, k, v, out, probs, ld, ldo, heads, scale, seed, seed_dev, th, ik);
    return 0;
  });
  MMI_CHECK_ARG(rc == 0, "mmi_attention_fwd: head dim %d unsupported (4,8,16,20,32,40,64,80,128,160)", dk);
  MMI_CHECK_LAUNCH("mmi_attention_fwd");
  return MMI_OK;
}

extern "C" int mmi_attention_bwd(const float* q, const float* k, const float* v, const float* probs, const float* dout,
                                 float* dq, float* dk_, float* dv, int B, int heads, int dk, int ld, float p_drop,
                                 uint64_t seed, const uint64_t* seed_dev, void* stream) {
  return mmi_attention_bwd_strided(q, k, v, probs, dout, dq, dk_, dv, B, heads, dk, ld, ld, p_drop, seed, seed_dev, stream);
}

extern "C" int mmi_attention_bwd_strided(const float* q, const float* k, const float* v, const float* probs,
                                         const float* dout, float* dq, float* dk_, float* dv, int B, int heads, int dk,
                                         int ld, int ldo, float p_drop, uint64_t seed, const uint64_t* seed_dev,
                                         void* stream) {
  MMI_CHECK_ARG(q && k && v && probs && dout && dq && dk_ && dv && B > 0 && heads > 0 && ld >= heads * dk && ld % 4 == 0 &&
                    ldo >= heads * dk && ldo % 4 == 0,
                "mmi_attention_bwd: bad arguments");
  MMI_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk_ | (uintptr_t)dv) & 15) == 0,
                "mmi_attention_bwd: operands must be 16-byte aligned");
  const float scale = 1.0f / sqrtf((float)dk);
  const uint32_t th = drop_thresh(p_drop);
  const float ik = 1.0f / (1.0f - p_drop);
  hipStream_t s = (hipStream_t)stream;
  const int rc = attn_dispatch(dk, [&](auto DKc) {
    constexpr int DK = decltype(DKc)::value;
    if constexpr (DK % 8 == 0)
      hipLaunchKernelGGL(attn_bwd_mfma_kernel<DK>, dim3(B * heads), dim3(256), 0, s, q, k, v, probs, dout, dq, dk_, dv, ld, ldo,
                         heads, scale, seed, seed_dev, th, ik);
    else
      hipLaunchKernelGGL(attn_bwd_kernel<DK>, dim3(B * heads), dim3(256), 0, s, q, k, v, probs, dout, dq, dk_, dv, ld, ldo, heads, scale,
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
