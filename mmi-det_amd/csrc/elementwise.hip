// Layout conversion, adds, strided copies, nearest upsample, SPP max-pools.  All HBM-bound: channel axis contiguous,
// 16-byte lane accesses wherever strides allow, grid-stride loops sized to fill 256 CUs.
//
// Replaces: the /255 split views -> NHWC (train.py:743-745 boundary), Focus slicing (models/common.py:708),
// Add/Add2/Concat (common.py:914-935, 740-748), nn.Upsample nearest x2 (YAML head), SPP max-pools (common.py:681-693),
// the Detect view/permute (models/yolo_test.py:54-55).
#include "common.h"

namespace {

inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b));
}
#define GRID_STRIDE(e, total) \
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (total); e += (int64_t)gridDim.x * blockDim.x)

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                    float* __restrict__ y, int N, int C, int H, int W) {
  const int64_t total = (int64_t)N * H * W;
  GRID_STRIDE(e, total) {
    const int w = (int)(e % W);
    const int64_t t = e / W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const float* src = x + n * sn + h * sh + w * sw;
    float* dst = y + e * C;
    for (int c = 0; c < C; ++c) dst[c] = src[c * sc];
  }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W) {
  const int64_t total = (int64_t)N * C * H * W;
  GRID_STRIDE(e, total) {
    const int w = (int)(e % W);
    int64_t t = e / W;
    const int h = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    y[e] = x[(((int64_t)n * H + h) * W + w) * C + c];
  }
}

// y(N,H/2,W/2,4C): channel q*C+c <- x(n, 2*oh+dy, 2*ow+dx, c), q = dy + 2*dx   (common.py:708 slice order)
// (ldd: row stride of the depth-side (N,H/2,W/2,4C) tensor, which may be a channel block of a wider buffer)
__global__ void s2d_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C, int ldd, int inverse) {
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)N * Ho * Wo * 4 * C;
  GRID_STRIDE(e, total) {
    const int cc = (int)(e % (4 * C));
    int64_t t = e / (4 * C);
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int q = cc / C, c = cc - q * C;
    const int dy = q & 1, dx = q >> 1;
    const int64_t xi = (((int64_t)n * H + 2 * oh + dy) * W + 2 * ow + dx) * C + c;
    const int64_t di = (e / (4 * C)) * ldd + cc;
    if (inverse) out[xi] = in[di];  // in: (N,H/2,W/2,4C) gradient, out: (N,H,W,C) gradient
    else out[di] = in[xi];
  }
}

template <int V>
__global__ void add_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                           float* __restrict__ o, int ldo, int64_t rows, int C) {
  const int cv = C / V;
  const int64_t total = rows * cv;
  GRID_STRIDE(e, total) {
    const int64_t r = e / cv;
    const int c = (int)(e - r * cv) * V;
    if (V == 4) {
      const f32x4 va = *reinterpret_cast<const f32x4*>(a + r * lda + c);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(b + r * ldb + c);
      *reinterpret_cast<f32x4*>(o + r * ldo + c) = va + vb;
    } else {
      o[r * ldo + c] = a[r * lda + c] + b[r * ldb + c];
    }
  }
}

template <int V>
__global__ void copy2d_kernel(const float* __restrict__ a, int lda, float* __restrict__ o, int ldo, int64_t rows, int C) {
  const int cv = C / V;
  const int64_t total = rows * cv;
  GRID_STRIDE(e, total) {
    const int64_t r = e / cv;
    const int c = (int)(e - r * cv) * V;
    if (V == 4) *reinterpret_cast<f32x4*>(o + r * ldo + c) = *reinterpret_cast<const f32x4*>(a + r * lda + c);
    else o[r * ldo + c] = a[r * lda + c];
  }
}

template <int V>
__global__ void upsample2x_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int N, int H, int W, int C) {
  const int cv = C / V, Ho = 2 * H, Wo = 2 * W;
  const int64_t total = (int64_t)N * Ho * Wo * cv;
  GRID_STRIDE(e, total) {
    const int c = (int)(e % cv) * V;
    int64_t t = e / cv;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const float* src = x + (((int64_t)n * H + (oh >> 1)) * W + (ow >> 1)) * ldx + c;
    float* dst = y + (((int64_t)n * Ho + oh) * Wo + ow) * ldy + c;
    if (V == 4) *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src);
    else dst[0] = src[0];
  }
}

template <int V>
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ skip, int ldskip,
                                      float* __restrict__ dx, int N, int H, int W, int C) {
  const int cv = C / V, Ho = 2 * H, Wo = 2 * W;
  const int64_t total = (int64_t)N * H * W * cv;
  GRID_STRIDE(e, total) {
    const int c = (int)(e % cv) * V;
    int64_t t = e / cv;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const float* s00 = dy + (((int64_t)n * Ho + 2 * h) * Wo + 2 * w) * lddy + c;
    const float* s10 = s00 + (int64_t)Wo * lddy;
    const int64_t pix = ((int64_t)n * H + h) * W + w;
    float* dst = dx + pix * C + c;
    if (V == 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(s00), b = *reinterpret_cast<const f32x4*>(s00 + lddy),
                  cc = *reinterpret_cast<const f32x4*>(s10), d = *reinterpret_cast<const f32x4*>(s10 + lddy);
      f32x4 r = (a + b) + (cc + d);
      if (skip != nullptr) r = *reinterpret_cast<const f32x4*>(skip + pix * ldskip + c) + r;   // (the map's other consumer)
      *reinterpret_cast<f32x4*>(dst) = r;
    } else {
      float r = (s00[0] + s00[lddy]) + (s10[0] + s10[lddy]);
      if (skip != nullptr) r = skip[pix * ldskip + c] + r;
      dst[0] = r;
    }
  }
}

// 5x5 stride-1 max-pool with implicit -inf padding; src/dst are channel slices (row strides lds/ldd)
template <int V>
__global__ void maxpool5_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int N, int H,
                                int W, int C) {
  const int cv = C / V;
  const int64_t total = (int64_t)N * H * W * cv;
  GRID_STRIDE(e, total) {
    const int c = (int)(e % cv) * V;
    int64_t t = e / cv;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float m[V];
#pragma unroll
    for (int k = 0; k < V; ++k) m[k] = -INFINITY;
    for (int dh = -2; dh <= 2; ++dh) {
      const int hh = h + dh;
      if (hh < 0 || hh >= H) continue;
      for (int dw = -2; dw <= 2; ++dw) {
        const int ww = w + dw;
        if (ww < 0 || ww >= W) continue;
        const float* p = src + (((int64_t)n * H + hh) * W + ww) * lds + c;
        if (V == 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
          for (int k = 0; k < V; ++k) m[k] = fmaxf(m[k], v[k]);
        } else {
          m[0] = fmaxf(m[0], p[0]);
        }
      }
    }
    float* o = dst + (((int64_t)n * H + h) * W + w) * ldd + c;
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = m[k];
  }
}

// SPP forward in ONE launch (models/common.py:681-693 of the reference: cat([x, mp5(x), mp9(x), mp13(x)]) with mp9 = mp5 o mp5,
// mp13 = mp5 o mp9 under -inf padding): a workgroup owns one image x CG channels, keeps the map in LDS ([pixel][CG] floats) and
// runs the three cascaded 5x5 pools as separable row / column passes on it, writing every stage to its slice of the concat
// buffer as it is produced -- one read of x and four writes instead of a copy launch plus three pool launches that each re-read
// 25 taps through L2 (82 -> ~15 us at 16 x 20 x 20 x 512).
template <int CG>
__global__ __launch_bounds__(512) void spp_fwd_tiled_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo, int H,
                                                           int W, int C) {
  extern __shared__ __align__(16) unsigned char spp_fsmem[];
  const int HW = H * W, E = HW * CG;
  float* A = reinterpret_cast<float*>(spp_fsmem);
  float* B = A + E;
  const int n = blockIdx.x, c0 = blockIdx.y * CG, t = threadIdx.x;
  // a thread owns channel c of the pixels p0, p0 + PP, p0 + 2 PP, ...: (h, w) advance by additions, no division per element
  constexpr int PP = 512 / CG;
  const int c = t % CG, p0 = t / CG, h00 = p0 / W, w00 = p0 - h00 * W, dh = PP / W, dw = PP - dh * W;
  const bool cok = c0 + c < C;
  const float* xp = x + (int64_t)n * HW * ldx + c0 + c;
  float* op = out + (int64_t)n * HW * ldo + c0 + c;
  for (int pix = p0; pix < HW; pix += PP) {
    float v = -INFINITY;
    if (cok) {
      v = xp[(int64_t)pix * ldx];
      op[(int64_t)pix * ldo] = v;
    }
    A[pix * CG + c] = v;
  }
  __syncthreads();
  for (int k = 1; k <= 3; ++k) {
    for (int pix = p0, w = w00; pix < HW; pix += PP) {           // rows: B = max over w - 2 .. w + 2
      const int e = pix * CG + c;
      float m = A[e];
      if (w >= 1) m = fmaxf(m, A[e - CG]);
      if (w >= 2) m = fmaxf(m, A[e - 2 * CG]);
      if (w + 1 < W) m = fmaxf(m, A[e + CG]);
      if (w + 2 < W) m = fmaxf(m, A[e + 2 * CG]);
      B[e] = m;
      w += dw;
      if (w >= W) w -= W;
    }
    __syncthreads();
    const int rs = W * CG;
    for (int pix = p0, h = h00, w = w00; pix < HW; pix += PP) {   // columns: A = max over h - 2 .. h + 2, the stage's output
      const int e = pix * CG + c;
      float m = B[e];
      if (h >= 1) m = fmaxf(m, B[e - rs]);
      if (h >= 2) m = fmaxf(m, B[e - 2 * rs]);
      if (h + 1 < H) m = fmaxf(m, B[e + rs]);
      if (h + 2 < H) m = fmaxf(m, B[e + 2 * rs]);
      A[e] = m;
      if (cok) op[(int64_t)pix * ldo + (int64_t)k * C] = m;
      h += dh;
      w += dw;
      if (w >= W) {
        w -= W;
        ++h;
      }
    }
    __syncthreads();
  }
}

// SPP backward, fallback for maps too large for the LDS-tiled form below (more than ~44 x 44 positions, i.e. P5 of a
// > 1400 x 1400 input): thread = one (pixel, channel, pool k in {5,9,13}); routes dcat[..,(1+k)C+c] to the first arg-max of
// the window of x (row-major scan, strict >, as ATen's max_pool2d) with a float atomic -- the one place left where two
// runs may differ in the last bit.
__global__ void spp_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dcat, int ldd,
                               float* __restrict__ dx, int lddx, int N, int H, int W, int C) {
  const int64_t total = (int64_t)N * H * W * C * 3;
  GRID_STRIDE(e, total) {
    const int c = (int)(e % C);
    int64_t t = e / C;
    const int pk = (int)(t % 3);
    t /= 3;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const int rad = 2 + 2 * pk;  // 5,9,13 -> radius 2,4,6
    float best = -INFINITY;
    int bh = -1, bw = -1;
    for (int hh = max(h - rad, 0); hh <= min(h + rad, H - 1); ++hh)
      for (int ww = max(w - rad, 0); ww <= min(w + rad, W - 1); ++ww) {
        const float v = x[(((int64_t)n * H + hh) * W + ww) * ldx + c];
        if (v > best || bh < 0) {
          best = v;
          bh = hh;
          bw = ww;
        }
      }
    const float g = dcat[(((int64_t)n * H + h) * W + w) * ldd + (1 + pk) * C + c];
    atomicAdd(dx + (((int64_t)n * H + bh) * W + bw) * lddx + c, g);
  }
}

// SPP backward, tiled form: one workgroup = one image x CG channels with the whole H x W map in LDS.  The arg-max of a
// (2r+1)^2 window with ATen's tie rule (first maximum in row-major order) is separable: per row the first column holding the
// row-window maximum (rv, rc), then the first row holding the maximum of those (bh) -- output (oh, ow) routes its gradient to
// input (bh, rc[bh][ow]).  Routing is a GATHER, so the result is run-to-run bit-identical (no float atomics), and since round 3
// the gather is separable too: first every (h, ow) sums, over the rows oh of its column window, the gradients of the outputs
// (oh, ow) whose bh is h (they all go to the same input, (h, rc[h][ow])); then every input (h, w) sums, over the columns ow of its
// row window, the column sums whose rc[h][ow] is w.  4(2r+1) LDS visits per element and pool instead of 2(2r+1) + (2r+1)^2
// (108 instead of 329 over the three pools: 0.36 -> 0.12 ms per launch at 16 x 20 x 20 x 512).  Identity branch first, then the
// 5x5, 9x9 and 13x13 pools; sums in ascending row, then ascending column order.
constexpr int SPP_BYTES_PER_ELEM = 18;  // xs, rv, acc, gv (float) + rc (u8) + bh (u8)
constexpr int SPP_THREADS = 1024;       // LDS allows one workgroup per CU: a large one, so that 16 waves hide the LDS latency
template <int CG>
__global__ __launch_bounds__(1024) void spp_bwd_tiled_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dcat,
                                                            int ldd, float* __restrict__ dx, int lddx, int H, int W, int C) {
  extern __shared__ __align__(16) unsigned char spp_smem[];
  const int HW = H * W, n = blockIdx.y, c0 = blockIdx.x * CG, E = HW * CG;
  float* xs = reinterpret_cast<float*>(spp_smem);   // [HW][CG]
  float* rv = xs + E;                                // row-window maxima; after the arg-max rows are known: column sums
  float* acc = rv + E;                               // gradient accumulator (element e is only ever touched by its owner thread)
  float* gv = acc + E;                               // this pool's output gradients
  unsigned char* rc = reinterpret_cast<unsigned char*>(gv + E);     // column of the row-window maximum (W <= 255)
  unsigned char* bhm = rc + E;                                       // arg-max row of every output pixel (H <= 255)
  const int t = threadIdx.x;
  for (int e = t; e < E; e += SPP_THREADS) {
    const int pix = e / CG, c = e - pix * CG;
    const bool ok = c0 + c < C;
    xs[e] = ok ? x[((int64_t)n * HW + pix) * ldx + c0 + c] : 0.f;
    acc[e] = ok ? dcat[((int64_t)n * HW + pix) * ldd + c0 + c] : 0.f;   // identity branch of the concat
  }
  __syncthreads();
  for (int pk = 0; pk < 3; ++pk) {
    const int rad = 2 + 2 * pk;
    for (int e = t; e < E; e += SPP_THREADS) {
      const int pix = e / CG, c = e - pix * CG, h = pix / W, w = pix - h * W;
      float best = -INFINITY;
      int bw = -1;
      for (int ww = max(w - rad, 0); ww <= min(w + rad, W - 1); ++ww) {
        const float v = xs[(h * W + ww) * CG + c];
        if (v > best || bw < 0) best = v, bw = ww;
      }
      rv[e] = best;
      rc[e] = (unsigned char)bw;
    }
    __syncthreads();
    for (int e = t; e < E; e += SPP_THREADS) {
      const int pix = e / CG, c = e - pix * CG, h = pix / W, w = pix - h * W;
      float best = -INFINITY;
      int bh = -1;
      for (int hh = max(h - rad, 0); hh <= min(h + rad, H - 1); ++hh) {
        const float v = rv[(hh * W + w) * CG + c];
        if (v > best || bh < 0) best = v, bh = hh;
      }
      bhm[e] = (unsigned char)bh;
      gv[e] = c0 + c < C ? dcat[((int64_t)n * HW + pix) * ldd + (1 + pk) * C + c0 + c] : 0.f;
    }
    __syncthreads();   // (rv is read no more: it now takes the column sums)
    for (int e = t; e < E; e += SPP_THREADS) {
      const int pix = e / CG, c = e - pix * CG, h = pix / W, ow = pix - h * W;
      float s = 0.f;
      for (int oh = max(h - rad, 0); oh <= min(h + rad, H - 1); ++oh) {
        const int o = (oh * W + ow) * CG + c;
        if (bhm[o] == (unsigned char)h) s += gv[o];
      }
      rv[e] = s;
    }
    __syncthreads();
    for (int e = t; e < E; e += SPP_THREADS) {
      const int pix = e / CG, c = e - pix * CG, h = pix / W, w = pix - h * W;
      float s = acc[e];
      for (int ow = max(w - rad, 0); ow <= min(w + rad, W - 1); ++ow) {
        const int o = (h * W + ow) * CG + c;
        if (rc[o] == (unsigned char)w) s += rv[o];
      }
      acc[e] = s;
    }
    __syncthreads();
  }
  for (int e = t; e < E; e += SPP_THREADS) {
    const int pix = e / CG, c = e - pix * CG;
    if (c0 + c < C) dx[((int64_t)n * HW + pix) * lddx + c0 + c] = acc[e];
  }
}

// Detect head: in (B, P=ny*nx, na*no) -> out (B, na, P, no)   [inverse: the gradient goes the other way]
__global__ void head_permute_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int na, int no,
                                    int P, int inverse) {
  const int64_t total = (int64_t)B * na * P * no;
  GRID_STRIDE(e, total) {
    const int o = (int)(e % no);
    int64_t t = e / no;
    const int pix = (int)(t % P);
    t /= P;
    const int a = (int)(t % na);
    const int b = (int)(t / na);
    const int64_t ii = (((int64_t)b * P + pix) * na + a) * no + o;
    if (inverse) out[ii] = in[e];
    else out[e] = in[ii];
  }
}

inline bool vec4(int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
  if (C % 4) return false;
  for (int l : lds)
    if (l % 4) return false;
  for (const void* p : ptrs)
    if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

// uint8 (N, 2*C3, H, W) loader batch -> two fp32 NHWC images x/255 (train.py:743-745 in one pass: the reference makes
// a float copy, divides, and hands out two strided channel slices that the first layers then re-layout)
__global__ void u8_pair_to_nhwc_kernel(const uint8_t* __restrict__ in, float* __restrict__ a, float* __restrict__ b, int N,
                                       int H, int W) {
  const int64_t hw = (int64_t)H * W, total = (int64_t)N * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / hw, pix = i - n * hw;
    const uint8_t* src = in + n * 6 * hw + pix;
    float v[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = (float)src[c * hw] / 255.0f;  // a true division, as `imgs.float() / 255.0`
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a[i * 3 + c] = v[c];
      b[i * 3 + c] = v[3 + c];
    }
  }
}

}  // namespace

extern "C" int mmi_u8_pair_to_nhwc(const uint8_t* in, float* rgb, float* ir, int N, int H, int W, void* stream) {
  MMI_CHECK_ARG(in && rgb && ir && N > 0 && H > 0 && W > 0, "mmi_u8_pair_to_nhwc: bad arguments");
  hipLaunchKernelGGL(u8_pair_to_nhwc_kernel, dim3(ew_blocks((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, in,
                     rgb, ir, N, H, W);
  MMI_CHECK_LAUNCH("mmi_u8_pair_to_nhwc");
  return MMI_OK;
}

extern "C" int mmi_nchw_to_nhwc(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* y, int N, int C,
                                int H, int W, void* stream) {
  MMI_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0, "mmi_nchw_to_nhwc: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_blocks((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, x, sn,
                     sc, sh, sw, y, N, C, H, W);
  MMI_CHECK_LAUNCH("mmi_nchw_to_nhwc");
  return MMI_OK;
}

extern "C" int mmi_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream) {
  MMI_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0, "mmi_nhwc_to_nchw: bad arguments");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_blocks((int64_t)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                     y, N, C, H, W);
  MMI_CHECK_LAUNCH("mmi_nhwc_to_nchw");
  return MMI_OK;
}

extern "C" int mmi_space_to_depth_ld(const float* in, float* out, int N, int H, int W, int C, int ldd, int inverse, void* stream) {
  MMI_CHECK_ARG(in && out && N > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && ldd >= 4 * C, "mmi_space_to_depth: bad arguments");
  hipLaunchKernelGGL(s2d_kernel, dim3(ew_blocks((int64_t)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, in, out, N, H, W,
                     C, ldd, inverse);
  MMI_CHECK_LAUNCH("mmi_space_to_depth");
  return MMI_OK;
}
extern "C" int mmi_space_to_depth(const float* in, float* out, int N, int H, int W, int C, int inverse, void* stream) {
  return mmi_space_to_depth_ld(in, out, N, H, W, C, 4 * C, inverse, stream);
}

extern "C" int mmi_add(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int64_t rows, int C,
                       void* stream) {
  MMI_CHECK_ARG(a && b && out && rows > 0 && C > 0 && lda >= C && ldb >= C && ldo >= C, "mmi_add: bad arguments");
  if (vec4(C, {lda, ldb, ldo}, {a, b, out}))
    hipLaunchKernelGGL(add_kernel<4>, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
                       out, ldo, rows, C);
  else
    hipLaunchKernelGGL(add_kernel<1>, dim3(ew_blocks(rows * C)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out,
                       ldo, rows, C);
  MMI_CHECK_LAUNCH("mmi_add");
  return MMI_OK;
}

extern "C" int mmi_copy2d(const float* in, int ldi, float* out, int ldo, int64_t rows, int C, void* stream) {
  MMI_CHECK_ARG(in && out && rows > 0 && C > 0 && ldi >= C && ldo >= C, "mmi_copy2d: bad arguments");
  if (vec4(C, {ldi, ldo}, {in, out}))
    hipLaunchKernelGGL(copy2d_kernel<4>, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, in, ldi, out,
                       ldo, rows, C);
  else
    hipLaunchKernelGGL(copy2d_kernel<1>, dim3(ew_blocks(rows * C)), dim3(256), 0, (hipStream_t)stream, in, ldi, out, ldo,
                       rows, C);
  MMI_CHECK_LAUNCH("mmi_copy2d");
  return MMI_OK;
}

// x rows with stride ldx, y rows with stride ldy: either may be a channel slice of a wider buffer (the neck's concat buffers)
extern "C" int mmi_upsample2x_ld(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, void* stream) {
  MMI_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0 && ldx >= C && ldy >= C, "mmi_upsample2x: bad arguments");
  if (vec4(C, {ldx, ldy}, {x, y}))
    hipLaunchKernelGGL(upsample2x_kernel<4>, dim3(ew_blocks((int64_t)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       y, ldy, N, H, W, C);
  else
    hipLaunchKernelGGL(upsample2x_kernel<1>, dim3(ew_blocks((int64_t)N * H * W * C * 4)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, y, ldy, N, H, W, C);
  MMI_CHECK_LAUNCH("mmi_upsample2x");
  return MMI_OK;
}
extern "C" int mmi_upsample2x(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  return mmi_upsample2x_ld(x, C, y, C, N, H, W, C, stream);
}

extern "C" int mmi_upsample2x_bwd_acc(const float* dy, int lddy, const float* skip, int ldskip, float* dx, int N, int H, int W, int C,
                                      void* stream) {
  MMI_CHECK_ARG(dy && dx && N > 0 && C > 0 && H > 0 && W > 0 && lddy >= C && (!skip || ldskip >= C), "mmi_upsample2x_bwd: bad arguments");
  if (vec4(C, {lddy, skip ? ldskip : 0}, {dy, dx, skip}))
    hipLaunchKernelGGL(upsample2x_bwd_kernel<4>, dim3(ew_blocks((int64_t)N * H * W * C / 4)), dim3(256), 0,
                       (hipStream_t)stream, dy, lddy, skip, ldskip, dx, N, H, W, C);
  else
    hipLaunchKernelGGL(upsample2x_bwd_kernel<1>, dim3(ew_blocks((int64_t)N * H * W * C)), dim3(256), 0,
                       (hipStream_t)stream, dy, lddy, skip, ldskip, dx, N, H, W, C);
  MMI_CHECK_LAUNCH("mmi_upsample2x_bwd");
  return MMI_OK;
}

extern "C" int mmi_upsample2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
  return mmi_upsample2x_bwd_acc(dy, C, nullptr, 0, dx, N, H, W, C, stream);
}

extern "C" int mmi_spp_pool_fwd(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int C, void* stream) {
  MMI_CHECK_ARG(x && out && N > 0 && C > 0 && H > 0 && W > 0 && ldx >= C && ldo >= 4 * C, "mmi_spp_pool_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = (int64_t)N * H * W;
  static const bool tiled_off = getenv("MMIDET_SPP_FWD_TILED") != nullptr && atoi(getenv("MMIDET_SPP_FWD_TILED")) == 0;   // (A/B switch)
  // one launch when the map fits LDS (two buffers of H*W*CG floats): CG = 16 up to 1024 positions, 8 up to 2048, 4 up to 4096
  const int hw = H * W;
  const int cg = hw <= 1024 ? 16 : (hw <= 2048 ? 8 : (hw <= 4096 ? 4 : 0));
  const bool alias = (const void*)x == (const void*)out;
  if (cg != 0 && !tiled_off && !alias && N <= 65535) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)spp_fwd_tiled_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
      (void)hipFuncSetAttribute((const void*)spp_fwd_tiled_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
      (void)hipFuncSetAttribute((const void*)spp_fwd_tiled_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
      attr = true;
    }
    const size_t lds = (size_t)2 * hw * cg * sizeof(float);
    const dim3 grid(N, cdiv(C, cg)), block(512);
    if (cg == 16) hipLaunchKernelGGL(spp_fwd_tiled_kernel<16>, grid, block, lds, s, x, ldx, out, ldo, H, W, C);
    else if (cg == 8) hipLaunchKernelGGL(spp_fwd_tiled_kernel<8>, grid, block, lds, s, x, ldx, out, ldo, H, W, C);
    else hipLaunchKernelGGL(spp_fwd_tiled_kernel<4>, grid, block, lds, s, x, ldx, out, ldo, H, W, C);
    MMI_CHECK_LAUNCH("mmi_spp_pool_fwd(tiled)");
    return MMI_OK;
  }
  if (int e = mmi_copy2d(x, ldx, out, ldo, rows, C, stream)) return e;
  const bool v = vec4(C, {ldo}, {out});
  for (int k = 0; k < 3; ++k) {  // mp5 = P(x), mp9 = P(mp5), mp13 = P(mp9)
    const float* src = out + (int64_t)k * C;
    float* dst = out + (int64_t)(k + 1) * C;
    if (v)
      hipLaunchKernelGGL(maxpool5_kernel<4>, dim3(ew_blocks(rows * C / 4)), dim3(256), 0, s, src, ldo, dst, ldo, N, H, W, C);
    else
      hipLaunchKernelGGL(maxpool5_kernel<1>, dim3(ew_blocks(rows * C)), dim3(256), 0, s, src, ldo, dst, ldo, N, H, W, C);
    MMI_CHECK_LAUNCH("mmi_spp_pool_fwd");
  }
  return MMI_OK;
}

extern "C" int mmi_spp_pool_bwd(const float* x, int ldx, const float* dcat, int ldd, float* dx, int lddx, int N, int H,
                                int W, int C, void* stream) {
  MMI_CHECK_ARG(x && dcat && dx && N > 0 && C > 0 && H > 0 && W > 0 && ldx >= C && ldd >= 4 * C && lddx >= C,
                "mmi_spp_pool_bwd: bad arguments");
  const int64_t rows = (int64_t)N * H * W;
  {  // tiled form when the map fits in LDS with at least 4 channels per workgroup
    const size_t per_c = (size_t)H * W * SPP_BYTES_PER_ELEM;
    // channels per workgroup: 8 = two workgroups per CU at 20 x 20 and all 1024 of them resident at once (16: one per CU, two
    // rounds).  Backward at 16 x 20 x 20 x 512 (tools/bench_spp.py): 150 us at 16, 117 at 8, 131 at 4.  MMIDET_SPP_BWD_CG: A/B.
    static const int cg_max = [] {
      const char* e = getenv("MMIDET_SPP_BWD_CG");
      const int v = e ? atoi(e) : 8;
      return v == 4 || v == 16 ? v : 8;
    }();
    const int cg = (cg_max >= 16 && per_c * 16 <= 150 * 1024) ? 16 : ((cg_max >= 8 && per_c * 8 <= 150 * 1024) ? 8 : (per_c * 4 <= 150 * 1024 ? 4 : 0));
    if (cg > 0 && W <= 255 && H <= 255) {
      const dim3 grid(cdiv(C, cg), N), block(SPP_THREADS);
      const size_t lds = per_c * cg;
      hipStream_t s = (hipStream_t)stream;
      static bool raised = false;   // dynamic LDS beyond 64 KB has to be allowed per kernel
      if (!raised) {
        (void)hipFuncSetAttribute((const void*)spp_bwd_tiled_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)spp_bwd_tiled_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)spp_bwd_tiled_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipGetLastError();
        raised = true;
      }
      if (cg == 16) hipLaunchKernelGGL(spp_bwd_tiled_kernel<16>, grid, block, lds, s, x, ldx, dcat, ldd, dx, lddx, H, W, C);
      else if (cg == 8) hipLaunchKernelGGL(spp_bwd_tiled_kernel<8>, grid, block, lds, s, x, ldx, dcat, ldd, dx, lddx, H, W, C);
      else hipLaunchKernelGGL(spp_bwd_tiled_kernel<4>, grid, block, lds, s, x, ldx, dcat, ldd, dx, lddx, H, W, C);
      MMI_CHECK_LAUNCH("mmi_spp_pool_bwd(tiled)");
      return MMI_OK;
    }
  }
  if (int e = mmi_copy2d(dcat, ldd, dx, lddx, rows, C, stream)) return e;
  hipLaunchKernelGGL(spp_bwd_kernel, dim3(ew_blocks(rows * C * 3)), dim3(256), 0, (hipStream_t)stream, x, ldx, dcat, ldd,
                     dx, lddx, N, H, W, C);
  MMI_CHECK_LAUNCH("mmi_spp_pool_bwd");
  return MMI_OK;
}

extern "C" int mmi_head_permute(const float* in, float* out, int B, int na, int no, int P, int inverse, void* stream) {
  MMI_CHECK_ARG(in && out && B > 0 && na > 0 && no > 0 && P > 0, "mmi_head_permute: bad arguments");
  hipLaunchKernelGGL(head_permute_kernel, dim3(ew_blocks((int64_t)B * na * P * no)), dim3(256), 0, (hipStream_t)stream,
                     in, out, B, na, no, P, inverse);
  MMI_CHECK_LAUNCH("mmi_head_permute");
  return MMI_OK;
}
