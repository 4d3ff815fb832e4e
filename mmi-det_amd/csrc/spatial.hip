// Spatial side of the fusion stack (HBM-bound, no MFMA): adaptive 8x8 average pooling into the token layout, bilinear
// 8x8 -> HxW upsampling fused with the Add2 residual, the Fusion-Focus spectral split (8x8 DFT through LDS), its
// separation loss, and the Info-Guided / Contrast-Bridge statistics in ONE pass over the P2 tensors.
//
// Replaces, in the reference: nn.AdaptiveAvgPool2d + flatten/cat/permute (models/common.py:395-396, 505-524, 1331-1343),
// F.interpolate(bilinear) + Add2 (common.py:548-550, 1364-1366, 924-935), extract_frequency2 (common.py:37-69),
// Seperation_loss (common.py:128-139), compute_contrastive_loss / compute_fusing_loss2 / compute_EntropyLoss
// (models/yolo_test.py:338-486).
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b));
}
#define GRID_STRIDE(e, total) \
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (total); e += (int64_t)gridDim.x * blockDim.x)

// adaptive pooling window of output index o over `in` inputs into 8 bins: [floor(o*in/8), ceil((o+1)*in/8))
__device__ __forceinline__ void pool_win(int o, int in, int& s, int& e) {
  s = (o * in) / 8;
  e = ((o + 1) * in + 7) / 8;
}

// block = (image n, bin oi*8+oj, 128-channel group); threads = 32 channel quads x 8 pixel lanes
__global__ __launch_bounds__(256) void avgpool8_fwd_kernel(const float* __restrict__ x, int ldx, int H, int W, int C,
                                                           float* __restrict__ out, int64_t obs, int old) {
  __shared__ f32x4 red[8][32];
  const int n = blockIdx.x >> 6, bin = blockIdx.x & 63, oi = bin >> 3, oj = bin & 7;
  const int cq = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.y * 128 + cq * 4;
  int hs, he, ws, we;
  pool_win(oi, H, hs, he);
  pool_win(oj, W, ws, we);
  const int ww = we - ws, np = (he - hs) * ww;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int p = pl; p < np; p += 8) {
      const int h = hs + p / ww, w = ws + p % ww;
      s += *reinterpret_cast<const f32x4*>(x + (((int64_t)n * H + h) * W + w) * ldx + c);
    }
  red[pl][cq] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    f32x4 t = red[0][cq];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += red[i][cq];
    const float inv = 1.0f / (float)np;
    *reinterpret_cast<f32x4*>(out + (int64_t)n * obs + (int64_t)bin * old + c) = t * inv;
  }
}

__global__ void avgpool8_bwd_kernel(const float* __restrict__ dp, int64_t dbs, int dld, const float* __restrict__ skip,
                                    int ldskip, float* __restrict__ dx, int lddx, int N, int H, int W, int C) {
  const int cv = C / 4;
  const int64_t total = (int64_t)N * H * W * cv;
  GRID_STRIDE(e, total) {
    const int c = (int)(e % cv) * 4;
    int64_t t = e / cv;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int oi = 0; oi < 8; ++oi) {
      int hs, he;
      pool_win(oi, H, hs, he);
      if (h < hs || h >= he) continue;
      for (int oj = 0; oj < 8; ++oj) {
        int ws, we;
        pool_win(oj, W, ws, we);
        if (w < ws || w >= we) continue;
        const float inv = 1.0f / (float)((he - hs) * (we - ws));
        acc += *reinterpret_cast<const f32x4*>(dp + (int64_t)n * dbs + (int64_t)(oi * 8 + oj) * dld + c) * inv;
      }
    }
    const int64_t pix = ((int64_t)n * H + h) * W + w;
    if (skip != nullptr) acc = *reinterpret_cast<const f32x4*>(skip + pix * ldskip + c) + acc;   // (the map's other consumer)
    *reinterpret_cast<f32x4*>(dx + pix * lddx + c) = acc;
  }
}

// ATen upsample_bilinear2d (align_corners=False, size given): src = (dst+0.5)*in/out - 0.5 clamped at 0
__device__ __forceinline__ void bil_coef(int dst, int out_size, int& i0, int& i1, float& l0, float& l1) {
  const float scale = 8.0f / (float)out_size;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + (i0 < 7 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

__device__ __forceinline__ f32x4 bil_sample(const float* __restrict__ tok, int64_t base, int tld, int c, int h0, int h1,
                                            float a0, float a1, int w0, int w1, float b0, float b1) {
  const f32x4 v00 = *reinterpret_cast<const f32x4*>(tok + base + (int64_t)(h0 * 8 + w0) * tld + c);
  const f32x4 v01 = *reinterpret_cast<const f32x4*>(tok + base + (int64_t)(h0 * 8 + w1) * tld + c);
  const f32x4 v10 = *reinterpret_cast<const f32x4*>(tok + base + (int64_t)(h1 * 8 + w0) * tld + c);
  const f32x4 v11 = *reinterpret_cast<const f32x4*>(tok + base + (int64_t)(h1 * 8 + w1) * tld + c);
  return a0 * (b0 * v00 + b1 * v01) + a1 * (b0 * v10 + b1 * v11);
}

// out = x + bilinear(tok)   (x may be null)
__global__ void upsample_add_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ tok, int64_t tbs,
                                        int tld, float* __restrict__ out, int ldo, int N, int H, int W, int C) {
  const int cv = C / 4;
  const int64_t total = (int64_t)N * H * W * cv;
  GRID_STRIDE(e, total) {
    const int c = (int)(e % cv) * 4;
    int64_t t = e / cv;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    int h0, h1, w0, w1;
    float a0, a1, b0, b1;
    bil_coef(h, H, h0, h1, a0, a1);
    bil_coef(w, W, w0, w1, b0, b1);
    f32x4 v = bil_sample(tok, (int64_t)n * tbs, tld, c, h0, h1, a0, a1, w0, w1, b0, b1);
    const int64_t pix = ((int64_t)n * H + h) * W + w;
    if (x != nullptr) v += *reinterpret_cast<const f32x4*>(x + pix * ldx + c);
    *reinterpret_cast<f32x4*>(out + pix * ldo + c) = v;
  }
}

// dtok[n][i*8+j][c] = sum_{h,w} wh(h,i) ww(w,j) dout[n,h,w,c]; block = (n, cell, 128-channel group)
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(const float* __restrict__ dout, int ldd,
                                                               float* __restrict__ dtok, int64_t tbs, int tld, int H, int W,
                                                               int C) {
  __shared__ f32x4 red[8][32];
  __shared__ float wh[1024], wv[1024];
  const int n = blockIdx.x >> 6, cell = blockIdx.x & 63, ci = cell >> 3, cj = cell & 7;
  const int cq = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.y * 128 + cq * 4;
  for (int h = threadIdx.x; h < H; h += 256) {
    int i0, i1;
    float l0, l1;
    bil_coef(h, H, i0, i1, l0, l1);
    wh[h] = (i0 == ci ? l0 : 0.f) + (i1 == ci ? l1 : 0.f);
  }
  for (int w = threadIdx.x; w < W; w += 256) {
    int i0, i1;
    float l0, l1;
    bil_coef(w, W, i0, i1, l0, l1);
    wv[w] = (i0 == cj ? l0 : 0.f) + (i1 == cj ? l1 : 0.f);
  }
  __syncthreads();
  // conservative support of cell index i along a length-L axis: src in (i-1, i+1)
  const int hlo = max(0, (int)floorf(((float)ci - 0.5f) * (float)H / 8.0f - 0.5f) - 1);
  const int hhi = min(H, (int)ceilf(((float)ci + 1.5f) * (float)H / 8.0f - 0.5f) + 2);
  const int wlo = max(0, (int)floorf(((float)cj - 0.5f) * (float)W / 8.0f - 0.5f) - 1);
  const int whi = min(W, (int)ceilf(((float)cj + 1.5f) * (float)W / 8.0f - 0.5f) + 2);
  const int ww = whi - wlo, np = (hhi - hlo) * ww;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int p = pl; p < np; p += 8) {
      const int h = hlo + p / ww, w = wlo + p % ww;
      const float k = wh[h] * wv[w];
      if (k != 0.f) s += k * *reinterpret_cast<const f32x4*>(dout + (((int64_t)n * H + h) * W + w) * ldd + c);
    }
  red[pl][cq] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    f32x4 t = red[0][cq];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += red[i][cq];
    *reinterpret_cast<f32x4*>(dtok + (int64_t)n * tbs + (int64_t)cell * tld + c) = t;
  }
}

// ---- Fusion-Focus spectral split: 8x8 2-D DFT through LDS, bin mask, inverse, fp16 rounding, times the pooled map ----
// One wave per (batch, channel) plane; lane = m*8+n.  keep_mask bit (u*8+v) = keep unshifted bin (u,v).
__constant__ float kCos8[8] = {1.f, 0.70710678118654752f, 0.f, -0.70710678118654752f, -1.f, -0.70710678118654752f, 0.f, 0.70710678118654752f};
__constant__ float kSin8[8] = {0.f, 0.70710678118654752f, 1.f, 0.70710678118654752f, 0.f, -0.70710678118654752f, -1.f, -0.70710678118654752f};

__global__ __launch_bounds__(256) void ffm_highpass_kernel(const float* __restrict__ pooled, float* __restrict__ out, int B,
                                                           int C, unsigned long long keep_mask) {
  __shared__ float re[4][64], im[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t plane = (int64_t)blockIdx.x * 4 + wv;
  const bool live = plane < (int64_t)B * C;
  const int b = live ? (int)(plane / C) : 0, c = live ? (int)(plane % C) : 0;
  const int m = lane >> 3, n = lane & 7;
  const float x = live ? pooled[((int64_t)b * 64 + lane) * C + c] : 0.f;
  re[wv][lane] = x;
  __syncthreads();
  // rows: F1[m][v] = sum_n x[m][n] e^{-2 pi i v n / 8}     (lane = (m, v=n))
  float ar = 0.f, ai = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float xv = re[wv][m * 8 + k];
    const int tw = (n * k) & 7;
    ar += xv * kCos8[tw];
    ai -= xv * kSin8[tw];
  }
  __syncthreads();
  re[wv][lane] = ar;
  im[wv][lane] = ai;
  __syncthreads();
  // columns: F[u][v] = sum_m F1[m][v] e^{-2 pi i u m / 8}  (lane = (u=m, v=n))
  float fr = 0.f, fi = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float pr = re[wv][k * 8 + n], pi = im[wv][k * 8 + n];
    const int tw = (m * k) & 7;
    const float cs = kCos8[tw], sn = -kSin8[tw];
    fr += pr * cs - pi * sn;
    fi += pr * sn + pi * cs;
  }
  if (!((keep_mask >> lane) & 1ull)) fr = fi = 0.f;
  __syncthreads();
  re[wv][lane] = fr;
  im[wv][lane] = fi;
  __syncthreads();
  // inverse along u: G1[m][v] = sum_u F[u][v] e^{+2 pi i u m / 8}
  float gr = 0.f, gi = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float pr = re[wv][k * 8 + n], pi = im[wv][k * 8 + n];
    const int tw = (m * k) & 7;
    const float cs = kCos8[tw], sn = kSin8[tw];
    gr += pr * cs - pi * sn;
    gi += pr * sn + pi * cs;
  }
  __syncthreads();
  re[wv][lane] = gr;
  im[wv][lane] = gi;
  __syncthreads();
  // inverse along v, real part only: g[m][n] = Re sum_v G1[m][v] e^{+2 pi i v n / 8} / 64
  float r = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int tw = (n * k) & 7;
    r += re[wv][m * 8 + k] * kCos8[tw] - im[wv][m * 8 + k] * kSin8[tw];
  }
  r *= (1.0f / 64.0f);
  const float hi = __half2float(__float2half(r));  // .half() of the reference (common.py:66-67)
  if (live) out[((int64_t)b * 64 + lane) * C + c] = hi * x;  // torch.mul(high, pooled)  (common.py:440-441)
}

// Seperation_loss over the rows M[(b,ch)][s] = g[(b*64+s)*8+ch] of four gate tensors (B,64,8):
// sum_{i<j} <M_i,M_j> / (l(l-1)) = (|sum_i M_i|^2 - sum_i |M_i|^2) / (2 l (l-1))
__global__ __launch_bounds__(256) void separation_loss_kernel(const float* __restrict__ g0, const float* __restrict__ g1,
                                                              const float* __restrict__ g2, const float* __restrict__ g3,
                                                              int B, float* __restrict__ out) {
  __shared__ double sv[4][64], sq[256];
  const int s = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const float* srcs[4] = {g0, g1, g2, g3};
  double vs = 0.0, q = 0.0;
  for (int k = 0; k < 4; ++k) {
    const int nrows = k < 2 ? 8 * B : B;  // the high-frequency gates contribute their first B rows only (common.py:487-489)
    for (int r = rl; r < nrows; r += 4) {
      const int b = r >> 3, ch = r & 7;
      const double v = (double)srcs[k][((int64_t)b * 64 + s) * 8 + ch];
      vs += v;
      q += v * v;
    }
  }
  sv[rl][s] = vs;
  sq[threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < 64) {
    const double t = sv[0][s] + sv[1][s] + sv[2][s] + sv[3][s];
    sv[0][s] = t * t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double n2 = 0.0, qq = 0.0;
    for (int i = 0; i < 64; ++i) n2 += sv[0][i];
    for (int i = 0; i < 256; ++i) qq += sq[i];
    const double l = 18.0 * B;
    out[0] = (float)((n2 - qq) / (2.0 * l * (l - 1.0)));
  }
}

// ---- IGM + CBM statistics: one pass over in_rgb / in_ir, the fused map recomputed on the fly from the 8x8 tokens ----
// acc[0..7] = sum a, b, f, a^2, b^2, f^2, a f, b f ; acc[8..10] = sum_pixels |normalize(d)|^2 for the 3 pairings
// Histograms: an LDS atomic instruction is served one lane per clock per BANK, so what it costs is its worst bank: with one copy
// of the 3 x 256 bins every wave instruction piles up on the few banks its lanes' bins fall into (measured: ~64 clocks, the round-3
// kernel's 275 us), and copies laid out copy-major change nothing (bank = bin % 64 whatever the copy: tried in round 2, slower).
// Here every bin has HC copies side by side -- word (hist * 256 + bin) * HC + (lane % HC) -- so a lane's bank is its own copy
// index plus HC * (bin parity bits): lanes of one instruction collide at most 64 / HC-fold (two-fold at HC = 32) whatever the data.
constexpr int HC = 32;   // 3 x 256 x 32 x 4 B = 96 KB of LDS: one 1024-thread workgroup per CU
__global__ __launch_bounds__(1024) void fusion_stats_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                            int ldb, const float* __restrict__ tok, int N, int H, int W, int C,
                                                            double* __restrict__ acc, unsigned int* __restrict__ hist) {
  __shared__ unsigned int lh[3 * 256 * HC];
  __shared__ double lacc[11];
  for (int i = threadIdx.x; i < 3 * 256 * HC; i += 1024) lh[i] = 0u;
  if (threadIdx.x < 11) lacc[threadIdx.x] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, hc = lane & (HC - 1);
  const int64_t npix = (int64_t)N * H * W;
  const int cv = C / 4;
  // A wave covers 64 channel quads: with C < 256 that is several pixels side by side (C = 128 at P2: two), so that every
  // vector -- and every LDS-atomic -- instruction works on full waves; lpp lanes per pixel, ppw pixels per wave.
  const int lpp = cv >= 64 ? 64 : (cv > 16 ? 32 : (cv > 8 ? 16 : 8)), ppw = 64 / lpp;
  const int sub = lane / lpp, ql = lane - sub * lpp;
  float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float cs[3] = {0.f, 0.f, 0.f};
  const int64_t img = (int64_t)H * W;
  const int64_t wpix = (int64_t)gridDim.x * 16 * ppw;
  for (int64_t base = ((int64_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * ppw; base < npix; base += wpix) {
    const int64_t pix = base + sub;
    const bool live = pix < npix;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    bool pair = false;
    if (live) {
      const int n = (int)(pix / img);
      const int64_t rem = pix - (int64_t)n * img;
      const int h = (int)(rem / W), w = (int)(rem % W);
      int h0, h1, w0, w1;
      float a0, a1, b0, b1;
      bil_coef(h, H, h0, h1, a0, a1);
      bil_coef(w, W, w0, w1, b0, b1);
      pair = n + 1 < N;
      for (int q = ql; q < cv; q += lpp) {
        const int c = q * 4;
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + pix * lda + c);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + pix * ldb + c);
        const int64_t tb = (int64_t)n * 128 * C;
        const f32x4 fv = 0.5f * (bil_sample(tok, tb, C, c, h0, h1, a0, a1, w0, w1, b0, b1) +
                                 bil_sample(tok, tb + 64 * (int64_t)C, C, c, h0, h1, a0, a1, w0, w1, b0, b1));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float x = av[k], y = bv[k], f = fv[k];
          m[0] += x; m[1] += y; m[2] += f; m[3] += x * x; m[4] += y * y; m[5] += f * f; m[6] += x * f; m[7] += y * f;
          if (x >= 0.f && x <= 1.f) atomicAdd(&lh[(0 * 256 + min((int)(x * 256.0f), 255)) * HC + hc], 1u);
          if (y >= 0.f && y <= 1.f) atomicAdd(&lh[(1 * 256 + min((int)(y * 256.0f), 255)) * HC + hc], 1u);
          if (f >= 0.f && f <= 1.f) atomicAdd(&lh[(2 * 256 + min((int)(f * 256.0f), 255)) * HC + hc], 1u);
        }
        const f32x4 dp = av - bv;
        d0 += dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2] + dp[3] * dp[3];
        if (pair) {
          const f32x4 an = *reinterpret_cast<const f32x4*>(a + (pix + img) * lda + c);
          const f32x4 bn = *reinterpret_cast<const f32x4*>(b + (pix + img) * ldb + c);
          const f32x4 dn = av - bn, dm = an - bv;
          d1 += dn[0] * dn[0] + dn[1] * dn[1] + dn[2] * dn[2] + dn[3] * dn[3];
          d2 += dm[0] * dm[0] + dm[1] * dm[1] + dm[2] * dm[2] + dm[3] * dm[3];
        }
      }
    }
    for (int o = lpp >> 1; o > 0; o >>= 1) {       // sums over the pixel's own lanes
      d0 += __shfl_xor(d0, o);
      d1 += __shfl_xor(d1, o);
      d2 += __shfl_xor(d2, o);
    }
    if (pair && ql == 0) {  // F.normalize(d, dim=1): d / max(|d|, 1e-12); sum_c of its square
      const float n0 = fmaxf(sqrtf(d0), 1e-12f), n1 = fmaxf(sqrtf(d1), 1e-12f), n2 = fmaxf(sqrtf(d2), 1e-12f);
      cs[0] += d0 / (n0 * n0);
      cs[1] += d1 / (n1 * n1);
      cs[2] += d2 / (n2 * n2);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float v = m[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) atomicAdd(&lacc[k], (double)v);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {                     // (cs lives in the first lane of every pixel group)
    float v = cs[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) atomicAdd(&lacc[8 + k], (double)v);
  }
  __syncthreads();
  if (threadIdx.x < 11) atomicAdd(&acc[threadIdx.x], lacc[threadIdx.x]);
  if (threadIdx.x < 768) {                          // fold the HC copies of this thread's bin (rotated start: no bank pile-up)
    unsigned int v = 0u;
#pragma unroll 8
    for (int c = 0; c < HC; ++c) v += lh[threadIdx.x * HC + ((c + threadIdx.x) & (HC - 1))];
    if (v) atomicAdd(&hist[threadIdx.x], v);
  }
}

// out[0] = SSIMloss (yolo_test.py:444-486), out[1] = Entropy_loss (406-429), out[2] = ContrastiveValue (338-404)
__global__ __launch_bounds__(256) void fusion_stats_finalize_kernel(const double* __restrict__ acc,
                                                                    const unsigned int* __restrict__ hist, double nelem,
                                                                    double npair_elem, float* __restrict__ out) {
  __shared__ double ent[3][256];
  __shared__ double tot[3];
  const int t = threadIdx.x;
  if (t < 3) {
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += (double)hist[t * 256 + i];
    tot[t] = s;
  }
  __syncthreads();
  for (int k = 0; k < 3; ++k) {
    const double cnt = (double)hist[k * 256 + t];
    const float p = (float)(cnt / tot[k]);  // hist /= hist.sum() in fp32
    ent[k][t] = cnt > 0.0 ? (double)(p * log2f(p)) : 0.0;
  }
  __syncthreads();
  if (t == 0) {
    double e[3];
    for (int k = 0; k < 3; ++k) {
      double s = 0.0;
      for (int i = 0; i < 256; ++i) s += ent[k][i];
      e[k] = -s;
    }
    out[1] = (float)((e[0] + e[1]) - e[2]);
    const double ma = acc[0] / nelem, mb = acc[1] / nelem, mf = acc[2] / nelem;
    const double va = acc[3] / nelem - ma * ma, vb = acc[4] / nelem - mb * mb, vf = acc[5] / nelem - mf * mf;
    const double caf = acc[6] / nelem - ma * mf, cbf = acc[7] / nelem - mb * mf;
    const double c1 = 0.01 * 0.01, c2 = 0.03 * 0.03;
    const double sa = (2 * ma * mf + c1) * (2 * caf + c2) / ((ma * ma + mf * mf + c1) * (va + vf + c2));
    const double sb = (2 * mb * mf + c1) * (2 * cbf + c2) / ((mb * mb + mf * mf + c1) * (vb + vf + c2));
    out[0] = (float)(0.5 * (1.0 - sa) + 0.5 * (1.0 - sb));  // + |std(f) - std(f)| = 0
    const double p0 = exp(acc[8] / npair_elem), p1 = exp(acc[9] / npair_elem) - 1.0, p2 = exp(acc[10] / npair_elem) - 1.0;
    out[2] = (float)((2.0 * p0 + p1 + p2) / 4.0);  // NaN when B == 1, as in the reference
  }
}

}  // namespace

extern "C" int mmi_avgpool8_fwd(const float* x, int ldx, int N, int H, int W, int C, float* out, int64_t out_batch_stride,
                                int out_ld, void* stream) {
  MMI_CHECK_ARG(x && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && out_ld % 4 == 0 &&
                    out_batch_stride % 4 == 0, "mmi_avgpool8_fwd: bad arguments");
  hipLaunchKernelGGL(avgpool8_fwd_kernel, dim3(N * 64, cdiv(C, 128)), dim3(256), 0, (hipStream_t)stream, x, ldx, H, W, C, out,
                     out_batch_stride, out_ld);
  MMI_CHECK_LAUNCH("mmi_avgpool8_fwd");
  return MMI_OK;
}

extern "C" int mmi_avgpool8_bwd_acc(const float* dpool, int64_t d_batch_stride, int d_ld, const float* skip, int ldskip, float* dx,
                                    int lddx, int N, int H, int W, int C, void* stream) {
  MMI_CHECK_ARG(dpool && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && lddx % 4 == 0 && d_ld % 4 == 0 &&
                    d_batch_stride % 4 == 0 && (!skip || (ldskip >= C && ldskip % 4 == 0 && ((uintptr_t)skip & 15) == 0)),
                "mmi_avgpool8_bwd: bad arguments");
  hipLaunchKernelGGL(avgpool8_bwd_kernel, dim3(ew_blocks((int64_t)N * H * W * C / 4)), dim3(256), 0, (hipStream_t)stream,
                     dpool, d_batch_stride, d_ld, skip, ldskip, dx, lddx, N, H, W, C);
  MMI_CHECK_LAUNCH("mmi_avgpool8_bwd");
  return MMI_OK;
}

extern "C" int mmi_avgpool8_bwd(const float* dpool, int64_t d_batch_stride, int d_ld, float* dx, int lddx, int N, int H,
                                int W, int C, void* stream) {
  return mmi_avgpool8_bwd_acc(dpool, d_batch_stride, d_ld, nullptr, 0, dx, lddx, N, H, W, C, stream);
}

extern "C" int mmi_upsample_add_fwd(const float* x, int ldx, const float* tok, int64_t tok_batch_stride, int tok_ld,
                                    float* out, int ldo, int N, int H, int W, int C, void* stream) {
  MMI_CHECK_ARG(tok && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && tok_ld % 4 == 0 && ldo % 4 == 0 &&
                    tok_batch_stride % 4 == 0 && (!x || ldx % 4 == 0), "mmi_upsample_add_fwd: bad arguments");
  hipLaunchKernelGGL(upsample_add_fwd_kernel, dim3(ew_blocks((int64_t)N * H * W * C / 4)), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, tok, tok_batch_stride, tok_ld, out, ldo, N, H, W, C);
  MMI_CHECK_LAUNCH("mmi_upsample_add_fwd");
  return MMI_OK;
}

extern "C" int mmi_upsample_add_bwd(const float* dout, int ldd, float* dtok, int64_t tok_batch_stride, int tok_ld, int N,
                                    int H, int W, int C, void* stream) {
  MMI_CHECK_ARG(dout && dtok && N > 0 && H > 0 && W > 0 && H <= 1024 && W <= 1024 && C > 0 && C % 4 == 0 && ldd % 4 == 0 &&
                    tok_ld % 4 == 0 && tok_batch_stride % 4 == 0, "mmi_upsample_add_bwd: bad arguments");
  hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(N * 64, cdiv(C, 128)), dim3(256), 0, (hipStream_t)stream, dout, ldd, dtok,
                     tok_batch_stride, tok_ld, H, W, C);
  MMI_CHECK_LAUNCH("mmi_upsample_add_bwd");
  return MMI_OK;
}

extern "C" int mmi_ffm_highpass(const float* pooled, float* out, int B, int C, uint64_t keep_mask, void* stream) {
  MMI_CHECK_ARG(pooled && out && B > 0 && C > 0, "mmi_ffm_highpass: bad arguments");
  hipLaunchKernelGGL(ffm_highpass_kernel, dim3(cdiv((int64_t)B * C, 4)), dim3(256), 0, (hipStream_t)stream, pooled, out, B,
                     C, (unsigned long long)keep_mask);
  MMI_CHECK_LAUNCH("mmi_ffm_highpass");
  return MMI_OK;
}

extern "C" int mmi_separation_loss(const float* m_rgb, const float* m_ir, const float* m_rgb_hi, const float* m_ir_hi, int B,
                                   float* out, void* stream) {
  MMI_CHECK_ARG(m_rgb && m_ir && m_rgb_hi && m_ir_hi && out && B > 0, "mmi_separation_loss: bad arguments");
  hipLaunchKernelGGL(separation_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, m_rgb, m_ir, m_rgb_hi, m_ir_hi, B,
                     out);
  MMI_CHECK_LAUNCH("mmi_separation_loss");
  return MMI_OK;
}

extern "C" size_t mmi_fusion_stats_workspace(void) { return 11 * sizeof(double) + 768 * sizeof(unsigned int); }

extern "C" int mmi_fusion_stats(const float* in_rgb, int lda, const float* in_ir, int ldb, const float* tokens, int N,
                                int H, int W, int C, void* workspace, float* out3, void* stream) {
  MMI_CHECK_ARG(in_rgb && in_ir && tokens && workspace && out3 && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 &&
                    lda % 4 == 0 && ldb % 4 == 0, "mmi_fusion_stats: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (int e = mmi_fill_bytes(workspace, 0, mmi_fusion_stats_workspace(), s)) return e;
  double* acc = (double*)workspace;
  unsigned int* hist = (unsigned int*)(acc + 11);
  const int64_t npix = (int64_t)N * H * W;
  int blocks = (int)((npix + 15) / 16);
  const int cus = 256;     // one 1024-thread workgroup per CU (96 KB of LDS histograms each)
  if (blocks > cus) blocks = cus;
  hipLaunchKernelGGL(fusion_stats_kernel, dim3(blocks), dim3(1024), 0, s, in_rgb, lda, in_ir, ldb, tokens, N, H, W, C, acc, hist);
  MMI_CHECK_LAUNCH("mmi_fusion_stats");
  hipLaunchKernelGGL(fusion_stats_finalize_kernel, dim3(1), dim3(256), 0, s, (const double*)acc, (const unsigned int*)hist,
                     (double)npix * C, (double)(N - 1) * H * W * C, out3);
  MMI_CHECK_LAUNCH("mmi_fusion_stats(finalize)");
  return MMI_OK;
}
