// Contour Enhancement Module kernels: the full-resolution 3->24->3 channel path of
// AdaptiveModule3 / EnhanceConv2d (models/common.py:751-911 of the reference).  With 3 or 24 channels an MFMA tile is
// 3/64 .. 24/64 occupied (the generic implicit GEMM ran these layers at 1-12 TFLOP/s, ~21 ms per step); here they are
// direct VALU convolutions with the 648 weights held in scalar registers, and the 24->24 "Sobel bank" conv is never
// materialised at all: every one of its 24 output channels is  factor[o] * stencil_{o%8}(sum_c r_c) + bias[o],  i.e. eight
// fixed 3x3 stencils of ONE channel-sum map, which is a memory-bound elementwise op (and so is its backward).
#include "common.h"

namespace {


constexpr int TS = 16;  // 16x16 pixel tile per 256-thread workgroup

// out[pix][o] = sum_{tap,c} in[pix + tap - 1][c] * Wsel(o, tap, c)  (+ bias[o]); 3x3, stride 1, zero pad 1, NHWC.
//   TRANSPOSED = false: forward conv,  Wsel(o,tap,c) = w[(o*9 + tap)*CIN + c]            (w is OHWI [COUT][9][CIN])
//   TRANSPOSED = true : data gradient, Wsel(o,tap,c) = w[(c*9 + 8 - tap)*COUT + o]       (w is OHWI [CIN][9][COUT])
// Weight indices are compile-time after unrolling and the pointer is uniform: hipcc keeps the weights in SGPRs.
template <int CIN, int COUT, bool TRANSPOSED>
__global__ __launch_bounds__(256) void smallconv_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int ldy,
                                                        float* __restrict__ stat_part, int H, int W) {
  const int tx = threadIdx.x & (TS - 1), ty = threadIdx.x >> 4;
  const int ow = blockIdx.x * TS + tx, oh = blockIdx.y * TS + ty, n = blockIdx.z;
  const bool live = ow < W && oh < H;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ih = oh + tap / 3 - 1, iw = ow + tap % 3 - 1;
    if (live && ih >= 0 && iw >= 0 && ih < H && iw < W) {
      const float* src = x + (((int64_t)n * H + ih) * W + iw) * ldx;
      float xin[CIN];
      if (CIN % 4 == 0) {
#pragma unroll
        for (int c = 0; c < CIN; c += 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
          xin[c] = v[0]; xin[c + 1] = v[1]; xin[c + 2] = v[2]; xin[c + 3] = v[3];
        }
      } else {
#pragma unroll
        for (int c = 0; c < CIN; ++c) xin[c] = src[c];
      }
#pragma unroll
      for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int o = 0; o < COUT; ++o)
          acc[o] += xin[c] * (TRANSPOSED ? w[(c * 9 + 8 - tap) * COUT + o] : w[(o * 9 + tap) * CIN + c]);
    }
  }
  if (bias != nullptr) {
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] += bias[o];
  }
  if (live && y != nullptr) {  // (y == nullptr: statistics pre-pass of the fused CEM forward, nothing is stored)
    float* dst = y + (((int64_t)n * H + oh) * W + ow) * ldy;
#pragma unroll
    for (int o = 0; o < COUT; ++o) dst[o] = acc[o];
  }
  if (stat_part != nullptr) {  // BatchNorm batch statistics: per-block column sums of y and y*y
    __shared__ float red[2][4][COUT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      float s1 = live ? acc[o] : 0.f, s2 = s1 * s1;
#pragma unroll
      for (int k = 32; k > 0; k >>= 1) {
        s1 += __shfl_xor(s1, k);
        s2 += __shfl_xor(s2, k);
      }
      if (lane == 0) {
        red[0][wv][o] = s1;
        red[1][wv][o] = s2;
      }
    }
    __syncthreads();
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (threadIdx.x < 2 * COUT) {
      const int s = threadIdx.x / COUT, o = threadIdx.x - s * COUT;
      stat_part[((int64_t)blk * 2 + s) * COUT + o] = red[s][0][o] + red[s][1][o] + red[s][2][o] + red[s][3][o];
    }
  }
}

// dW[co][tap][ci] = sum_pix dy[pix][co] * x[pix + tap - 1][ci]; each workgroup walks 16x16 pixel tiles with x (haloed)
// and dy staged in LDS; per-workgroup partials, summed afterwards (rowsum_kernel).  One of the two channel counts is 3:
// a thread owns one (tap, wide-channel) pair and the three narrow channels, so a pixel costs it one 16-byte LDS read of the
// narrow operand (padded to 4 floats; the dy form is a wave-wide broadcast) plus one 4-byte read of the wide one for three
// FMAs -- a third of the LDS instructions of the output-per-thread form, which was LDS-issue bound.
// Round 2: the next tile's operands are fetched into registers while the current tile is being multiplied (the wide operand
// with 16-byte loads when VEC), and the three narrow channels of a pixel cost two VALU instructions -- one packed FMA
// (v_pk_fma_f32, the wide value broadcast to both halves) + one FMA -- instead of three.
// BNF (conv2 of the CEM only: CIN = 3, the image, which takes no gradient): the wide operand is not given but made on the way
// into LDS -- dy = BatchNorm2+LeakyReLU backward of (dr, y2) with the finished dgamma / dbeta (the arithmetic of
// bn_bwd_apply_kernel, term for term) -- so dy2 is never written or read: one 629 MB write + read and one launch less per call.
// A thread keeps one 4-channel group for all its pieces (t % 6), so the group's constants stay in registers.
struct BnApply {
  const float* y;          // y2 [pixels][24]
  const float* mi;         // mean[24] | invstd[24]
  const float* gamma;
  const float* beta;
  const float* dgamma;     // finished sums (mmi_bn_act_bwd with dy = NULL)
  const float* dbeta;
  float inv_rows;
  int frozen;
};
template <int CIN, int COUT, bool VEC, bool BNF = false>
__global__ __launch_bounds__(256) void smallconv_wgrad_kernel(const float* __restrict__ x, int ldx,
                                                              const float* __restrict__ dy, int ldy,
                                                              float* __restrict__ partials, int N, int H, int W, BnApply bn) {
  static_assert((CIN == 3) != (COUT == 3), "one side has 3 channels");
  static_assert(!BNF || (CIN == 3 && VEC), "the fused BatchNorm backward is conv2's");
  constexpr bool XN = CIN == 3;                    // x is the narrow operand (conv2: 3 -> 24), else dy is (conv3: 24 -> 3)
  constexpr int WIDE = XN ? COUT : CIN;            // 24
  constexpr int NOUT = COUT * 9 * CIN;
  constexpr int XE = XN ? 4 : CIN, DE = XN ? COUT : 4;   // floats per pixel in LDS
  constexpr int XP = (TS + 2) * (TS + 2), DP = TS * TS;  // pixels of the two tiles (x with its halo)
  constexpr int WP = XN ? DP : XP, NP_ = XN ? XP : DP;   // pixels of the wide / narrow tile
  constexpr int WV = BNF ? (WP + 41) / 42 : (WP * 6 + 255) / 256;   // 16-byte pieces of the wide tile per thread (BNF: 42 pixels x 6 groups per round)
  constexpr int NV = (NP_ + 255) / 256;                  // narrow pixels per thread
  __shared__ __align__(16) float xs[XP * XE];
  __shared__ __align__(16) float ds[DP * DE];
  const int t = threadIdx.x;
  const bool owner = t < 9 * WIDE;
  const int tap = owner ? t / WIDE : 0, wc = owner ? t - tap * WIDE : 0;
  const int xoff = ((tap / 3) * (TS + 2) + tap % 3) * XE + (XN ? 0 : wc);   // of the tap, relative to the pixel
  f32x2 acc01 = {0.f, 0.f};
  float acc2 = 0.f;
  const int tw = (W + TS - 1) / TS, th = (H + TS - 1) / TS;
  const int ntiles = tw * th * N;
  const float* wsrc = XN ? dy : x;
  const float* nsrc = XN ? x : dy;
  const int ldw = XN ? ldy : ldx, ldn = XN ? ldx : ldy;
  f32x4 wreg[WV];
  f32x4 yreg[BNF ? WV : 1];
  unsigned wmask = 0;                                    // (BNF) bit i: piece i of the loaded tile lies inside the image
  float nreg[NV][3];
  const int bcg = t % 6, bpg = t / 6;                    // (BNF) channel group and first pixel of the thread
  f32x4 cm, cis, cga, cbe, ck1, ck2;
  if constexpr (BNF) {
    const int c = bcg * 4;      // (dword loads: parameters and their gradients may sit at any 4-byte offset of a packed buffer / bucket)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      cm[k] = bn.mi[c + k];
      cis[k] = bn.mi[24 + c + k];
      cga[k] = bn.gamma[c + k];
      cbe[k] = bn.beta[c + k];
      ck1[k] = bn.dbeta[c + k];
      ck2[k] = bn.dgamma[c + k];
    }
  }
  auto gload = [&](int tile) {
    const int n = tile / (tw * th), r = tile - n * tw * th;
    const int h0 = (r / tw) * TS, w0 = (r % tw) * TS;
    if constexpr (BNF) {
      wmask = 0;
#pragma unroll
      for (int i = 0; i < WV; ++i) {
        const int p = bpg + i * 42, ih = h0 + p / TS, iw = w0 + p % TS;
        wreg[i] = yreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < 252 && p < WP && ih < H && iw < W) {
          wmask |= 1u << i;
          const int64_t off = (((int64_t)n * H + ih) * W + iw) * 24 + bcg * 4;
          wreg[i] = *reinterpret_cast<const f32x4*>(wsrc + off);
          yreg[i] = *reinterpret_cast<const f32x4*>(bn.y + off);
        }
      }
    } else {
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = t + i * 256, p = v / 6, c4 = (v - p * 6) * 4;
      wreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (v < WP * 6) {
        const int ih = XN ? h0 + p / TS : h0 + p / (TS + 2) - 1, iw = XN ? w0 + p % TS : w0 + p % (TS + 2) - 1;
        if (ih >= 0 && iw >= 0 && ih < H && iw < W) {
          const float* src = wsrc + (((int64_t)n * H + ih) * W + iw) * ldw + c4;
          if (VEC) wreg[i] = *reinterpret_cast<const f32x4*>(src);
          else wreg[i] = f32x4{src[0], src[1], src[2], src[3]};
        }
      }
    }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int p = t + i * 256;
      nreg[i][0] = nreg[i][1] = nreg[i][2] = 0.f;
      if (p < NP_) {
        const int ih = XN ? h0 + p / (TS + 2) - 1 : h0 + p / TS, iw = XN ? w0 + p % (TS + 2) - 1 : w0 + p % TS;
        if (ih >= 0 && iw >= 0 && ih < H && iw < W) {
          const float* src = nsrc + (((int64_t)n * H + ih) * W + iw) * ldn;
          nreg[i][0] = src[0];
          nreg[i][1] = src[1];
          nreg[i][2] = src[2];
        }
      }
    }
  };
  auto lstore = [&]() {
    float* wdst = XN ? ds : xs;
    float* ndst = XN ? xs : ds;
    if constexpr (BNF) {
#pragma unroll
      for (int i = 0; i < WV; ++i) {
        const int p = bpg + i * 42;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {      // (bn_bwd_apply_kernel's arithmetic; zero for the pixels of the tile outside the image)
          const float xh = (yreg[i][k] - cm[k]) * cis[k];
          const float dz = wreg[i][k] * ((xh * cga[k] + cbe[k]) > 0.f ? 1.0f : 0.1f);
          const float tt = bn.frozen ? dz : dz - ck1[k] * bn.inv_rows - xh * ck2[k] * bn.inv_rows;
          o[k] = (wmask >> i & 1u) ? cga[k] * cis[k] * tt : 0.f;
        }
        if (t < 252 && p < WP) *reinterpret_cast<f32x4*>(wdst + (p * 6 + bcg) * 4) = o;
      }
    } else {
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = t + i * 256;
      if (v < WP * 6) *reinterpret_cast<f32x4*>(wdst + v * 4) = wreg[i];      // [pixel][24]: piece v lies at float 4 v
    }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int p = t + i * 256;
      if (p < NP_) *reinterpret_cast<f32x4*>(ndst + p * 4) = f32x4{nreg[i][0], nreg[i][1], nreg[i][2], 0.f};
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) gload(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                       // the previous tile has been consumed
    lstore();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) gload(tile + gridDim.x);      // in flight during the multiply below
    if (owner) {
      for (int pr = 0; pr < TS; ++pr) {
        const float* xrow = xs + pr * (TS + 2) * XE + xoff;
        const float* drow = ds + pr * TS * DE;
#pragma unroll
        for (int pc = 0; pc < TS; ++pc) {
          f32x4 nv;
          float wv;
          if (XN) {
            nv = *reinterpret_cast<const f32x4*>(xrow + pc * 4);
            wv = drow[pc * DE + wc];
          } else {
            nv = *reinterpret_cast<const f32x4*>(drow + pc * 4);
            wv = xrow[pc * XE];
          }
          acc01 += f32x2{wv, wv} * f32x2{nv[0], nv[1]};
          acc2 += wv * nv[2];
        }
      }
    }
  }
  if (owner) {
    const float acc[3] = {acc01[0], acc01[1], acc2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int co = XN ? wc : k, ci = XN ? k : wc;
      partials[(int64_t)blockIdx.x * NOUT + (co * 9 + tap) * CIN + ci] = acc[k];
    }
  }
}

// out[i] = sum_p partials[p][i] in fp64, fixed order: 16 columns x 16 part-lanes per workgroup (a serial loop over 2048
// parts per thread was 0.73 ms of pure load latency), lanes folded through LDS
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ partials, int nparts, int n,
                                                      float* __restrict__ out) {
  __shared__ double red[16][17];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cl;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (i < n) {
    int p = pl;
    for (; p + 48 < nparts; p += 64) {
      s0 += (double)partials[(int64_t)p * n + i];
      s1 += (double)partials[(int64_t)(p + 16) * n + i];
      s2 += (double)partials[(int64_t)(p + 32) * n + i];
      s3 += (double)partials[(int64_t)(p + 48) * n + i];
    }
    for (; p < nparts; p += 16) s0 += (double)partials[(int64_t)p * n + i];
  }
  red[pl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pl == 0 && i < n) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][cl];
    out[i] = (float)s;
  }
}

// ---- the stencil bank of EnhanceConv2d (models/common.py:838-882), index = out_channel % 8 ---------------------------
__device__ __forceinline__ void stencils8(const float r[9], float s[8]) {
  // r = 3x3 neighbourhood of the channel-sum map, row-major (r[4] = centre)
  s[0] = -r[0] - 2.f * r[1] - r[2] + r[6] + 2.f * r[7] + r[8];
  s[1] = -r[0] + r[2] - 2.f * r[3] + 2.f * r[5] - r[6] + r[8];
  s[2] = -2.f * r[0] - r[1] - r[3] + r[5] + r[7] + 2.f * r[8];
  s[3] = s[2];  // the reference's second diagonal branch is identical (common.py:849-862)
  s[4] = r[1] + r[3] - 4.f * r[4] + r[5] + r[7];
  s[5] = r[1] + r[3] + 4.f * r[4] + r[5] + r[7];
  s[6] = -r[0] - r[1] - r[2] + r[6] + r[7] + r[8];
  s[7] = -r[0] + r[2] - r[3] + r[5] - r[6] + r[8];
}
// transpose: contribution of D_k at neighbour offset d to the centre = stencil_k[-d] * D_k; given the 3x3 neighbourhood
// dn[k][9] of the eight D maps, returns sum_k sum_d stencil_k[8-d] * dn[k][d]
__device__ __forceinline__ float stencils8_t(const float dn[8][9]) {
  float a = 0.f;
  const float* d;
  d = dn[0]; a += -d[8] - 2.f * d[7] - d[6] + d[2] + 2.f * d[1] + d[0];
  d = dn[1]; a += -d[8] + d[6] - 2.f * d[5] + 2.f * d[3] - d[2] + d[0];
  d = dn[2]; a += -2.f * d[8] - d[7] - d[5] + d[3] + d[1] + 2.f * d[0];
  d = dn[3]; a += -2.f * d[8] - d[7] - d[5] + d[3] + d[1] + 2.f * d[0];
  d = dn[4]; a += d[7] + d[5] - 4.f * d[4] + d[3] + d[1];
  d = dn[5]; a += d[7] + d[5] + 4.f * d[4] + d[3] + d[1];
  d = dn[6]; a += -d[8] - d[7] - d[6] + d[2] + d[1] + d[0];
  d = dn[7]; a += -d[8] + d[6] - d[5] + d[3] - d[2] + d[0];
  return a;
}

template <int C>
__global__ void chansum_kernel(const float* __restrict__ r, int ldr, float* __restrict__ R, int64_t npix) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const float* src = r + p * ldr;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; c += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  R[p] = s;
}

__device__ __forceinline__ void load9(const float* __restrict__ R, int n, int h, int w, int H, int W, float r[9]) {
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int ih = h + k / 3 - 1, iw = w + k % 3 - 1;
    r[k] = (ih >= 0 && iw >= 0 && ih < H && iw < W) ? R[((int64_t)n * H + ih) * W + iw] : 0.f;
  }
}

// t[pix][o] = r[pix][o] + factor[o] * stencil_{o%8}(R)(pix) + bias[o]
template <int C>
__global__ void sobel_add_fwd_kernel(const float* __restrict__ r, int ldr, const float* __restrict__ R,
                                     const float* __restrict__ factor, const float* __restrict__ bias,
                                     float* __restrict__ t, int ldt, int N, int H, int W) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (int64_t)N * H * W) return;
  const int w = (int)(p % W), h = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
  float nb[9], s[8];
  load9(R, n, h, w, H, W, nb);
  stencils8(nb, s);
  const float* src = r + p * ldr;
  float* dst = t + p * ldt;
#pragma unroll
  for (int c = 0; c < C; c += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = v[k] + factor[c + k] * s[(c + k) & 7] + bias[c + k];
    *reinterpret_cast<f32x4*>(dst + c) = o;
  }
}

// ---- fused CEM forward: x -> conv2 -> BN2 + LeakyReLU -> r + stencil bank -> conv3, one 16x16 output
// tile per workgroup with everything between x and y3 in LDS: x is staged with a halo of 3, r (24 channels) and its channel
// sum are computed on the 20x20 region the stencils of the 18x18 t-region need, t on the 18x18 region conv3 needs.  BN2's
// batch statistics come from a recompute pre-pass (smallconv_kernel<3,24> with y = nullptr).  What the backward needs is
// written on the way -- y2, t (24 channels each), the channel-sum map and y3 with its statistics partials -- so HBM sees
// 1 read of x and 1 write each of y2 / t instead of the 4 reads + 4 writes of 24-channel maps of the unfused chain.
// Positions outside the image contribute zeros exactly where the reference's zero padding puts them (r for the stencils,
// t for conv3).
// Round 4: in TRAINING the kernel runs as its FROMY2 form behind cem_conv2_fwd_kernel (conv2 evaluated once, y2 stored with BN2's
// statistics): phase 1 then reads y2 instead of recomputing it on the halo region.  The recomputing form below stays for inference
// and for `MMIDET_CEM_TWO_PASS=0`.

// (the read-only operands are separate __restrict__ kernel arguments, not struct members: only then does hipcc fetch the 1300
//  uniform weights with scalar loads; through a by-value struct they became vector loads held in 256 VGPRs)
struct CemOut {
  float* y2;
  float* t;
  float* chansum;
  float* y3;
  float* stat_part;
};
template <int OB, bool Y2S, int WPE, bool FROMY2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void cem_fused_fwd_kernel(const float* __restrict__ px, const float* __restrict__ pw2,
                                                            const float* __restrict__ pmi2, const float* __restrict__ pg2,
                                                            const float* __restrict__ pb2, const float* __restrict__ pfactor,
                                                            const float* __restrict__ psbias, const float* __restrict__ pw3,
                                                            CemOut out, int ldx, int H_, int W_) {
  struct {
    const float* __restrict__ x; const float* __restrict__ w2; const float* __restrict__ mi2; const float* __restrict__ g2;
    const float* __restrict__ b2; const float* __restrict__ factor; const float* __restrict__ sbias; const float* __restrict__ w3;
    float* y2; float* t; float* chansum; float* y3; float* stat_part; int ldx, H, W;
  } p{px, pw2, pmi2, pg2, pb2, pfactor, psbias, pw3, out.y2, out.t, out.chansum, out.y3, out.stat_part, ldx, H_, W_};
  constexpr int XS = TS + 6, RS = TS + 4, TT = TS + 2;       // 22, 20, 18
  // Record pitch of a position's 24 channels in LDS: 28 floats.  At 24 (96 B) the 16-byte reads of the stencil and conv3 phases
  // fell on 8 of the 16 bank groups and the scalar stores of the conv2 phase on 8 of the 64 banks -- 97 M of the kernel's 172 M
  // LDS cycles were bank conflicts (round 4, profiles/r04_pmc_cem_module.txt); 112 B = 7 bank groups makes the vector reads
  // conflict-free and halves the stores' pile-up.  The room comes from the stencil phase, which now writes t IN PLACE over r (t at a
  // position needs r at that position and the channel sums of its neighbours, which live in cs): no separate t tile, 52 KB of LDS
  // instead of 77, three workgroups per CU instead of two.
  constexpr int RP = 28;
  __shared__ float xs[FROMY2 ? 4 : XS * XS * 3];
  __shared__ __align__(16) float rs[RS * RS * RP];
  __shared__ float cs[RS * RS];
  // y2 (conv2's raw output, kept for the backward) goes to HBM through LDS: a thread of the conv2 phase owns two positions and
  // produces their channels two at a time, so storing from its registers meant 24 dword stores per position whose 64 lanes hit 64
  // different cache lines -- 5.4 M store instructions per launch, each a lane-by-lane pass through the memory pipe, about half of
  // the kernel's time.  From the tile in LDS the same bytes leave as 16-byte lane accesses over contiguous 1.5 KB runs.
  // (Y2S = false, OB = 4 only: a position's four channels of an iteration leave as one 16-byte store from registers -- a
  //  quarter of the LDS form's instructions again at stride 96 B, but no 24 KB tile: three workgroups per CU)
  static_assert(Y2S || OB == 4 || FROMY2, "direct y2 stores need four channels per iteration");
  __shared__ __align__(16) float y2s[Y2S && !FROMY2 ? TS * TS * 24 : 4];
  __shared__ float red[2][4][3];
  const int t = threadIdx.x, n = blockIdx.z;
  const int h0 = blockIdx.y * TS, w0 = blockIdx.x * TS;
  const int H = p.H, W = p.W;
  if constexpr (FROMY2) {
    // Training (second half; the first is cem_conv2_fwd_kernel): r = LeakyReLU(BN2(y2)) on the 20x20 region from the STORED y2, two
    // positions per thread, all twelve 16-byte loads of a thread in flight together.  Same arithmetic on the same y2 as the
    // recomputing form, so r, t and y3 are bit-identical to it.
    if (t < RS * RS / 2) {
      f32x4 yv[2][6];
      int qq[2];
      bool in[2], interior[2];
      int64_t pix[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        qq[h] = t + h * (RS * RS / 2);
        const int i = qq[h] / RS, j = qq[h] % RS, ih = h0 - 2 + i, iw = w0 - 2 + j;
        in[h] = ih >= 0 && iw >= 0 && ih < H && iw < W;
        interior[h] = in[h] && i >= 2 && i < RS - 2 && j >= 2 && j < RS - 2;
        pix[h] = ((int64_t)n * H + ih) * W + iw;
        const float* src = p.y2 + pix[h] * 24;
#pragma unroll
        for (int g = 0; g < 6; ++g) yv[h][g] = in[h] ? *reinterpret_cast<const f32x4*>(src + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int o = g * 4 + k;
            const float z = (yv[h][g][k] - p.mi2[o]) * p.mi2[24 + o] * p.g2[o] + p.b2[o];
            v[k] = in[h] ? (z > 0.f ? z : 0.1f * z) : 0.f;
          }
          *reinterpret_cast<f32x4*>(rs + qq[h] * RP + g * 4) = v;
          sum += (v[0] + v[1]) + (v[2] + v[3]);       // (the grouping of chansum_kernel)
        }
        cs[qq[h]] = sum;
        if (p.chansum != nullptr && interior[h]) p.chansum[pix[h]] = sum;
      }
    }
  } else {
    {      // (all six loads of a thread in flight at once: as a rolled loop each waited for its own HBM round trip)
      constexpr int NX = (XS * XS * 3 + 255) / 256;
      float xv[NX];
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const int e = t + u * 256, c = e % 3, q = e / 3, ih = h0 - 3 + q / XS, iw = w0 - 3 + q % XS;
        xv[u] = (e < XS * XS * 3 && ih >= 0 && iw >= 0 && ih < H && iw < W) ? p.x[(((int64_t)n * H + ih) * W + iw) * p.ldx + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < NX; ++u)
        if (t + u * 256 < XS * XS * 3) xs[t + u * 256] = xv[u];
    }
    __syncthreads();
  // ---- r = LeakyReLU(BN2(conv2(x))) on the 20x20 region; zero outside the image.  A thread owns TWO positions (q, q + 200)
  // and keeps their accumulators as the halves of float2 registers: every FMA is a v_pk_fma_f32 whose weight operand is one
  // SGPR broadcast to both halves, i.e. twice the fp32 VALU rate of the scalar form, and the 648 weights are fetched once
  // for two pixels.  Per-position sums keep the tap-major / channel-minor order of smallconv_kernel<3, 24>.
  if (t < RS * RS / 2) {
    int qq[2], ypos[2];
    bool in[2], interior[2];
    f32x2 xin[27];                       // the two positions' 3x3x3 input patches, position h in half h
    {
      const float* sp[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        qq[h] = t + h * (RS * RS / 2);
        const int i = qq[h] / RS, j = qq[h] % RS, ih = h0 - 2 + i, iw = w0 - 2 + j;
        in[h] = ih >= 0 && iw >= 0 && ih < H && iw < W;
        interior[h] = in[h] && i >= 2 && i < RS - 2 && j >= 2 && j < RS - 2;
        ypos[h] = (i - 2) * TS + (j - 2);      // (only used where interior)
        sp[h] = xs + (i * XS + j) * 3;
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int off = ((tap / 3) * XS + tap % 3) * 3 + c;
          xin[tap * 3 + c] = f32x2{sp[0][off], sp[1][off]};
        }
    }
    const int64_t pix0 = ((int64_t)n * H + (h0 - 2 + qq[0] / RS)) * W + (w0 - 2 + qq[0] % RS);
    const int64_t pix1 = ((int64_t)n * H + (h0 - 2 + qq[1] / RS)) * W + (w0 - 2 + qq[1] % RS);
    f32x2 sum{0.f, 0.f};
    // OB output channels per iteration: their OB*27 weights are one contiguous run of w2, fetched with scalar loads inside
    // the iteration (the address depends on the loop variable, so nothing is hoisted and nothing spills)
#pragma unroll 1
    for (int o0 = 0; o0 < 24; o0 += OB) {
      const float* wp = p.w2 + o0 * 27;
      f32x2 acc[OB];
#pragma unroll
      for (int k = 0; k < OB; ++k) acc[k] = f32x2{0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 27; ++e)            // e = tap * 3 + c: the order of smallconv_kernel<3, 24>
#pragma unroll
        for (int k = 0; k < OB; ++k) {
          const float w = wp[k * 27 + e];
          acc[k] += xin[e] * f32x2{w, w};
        }
      f32x2 v[OB];
#pragma unroll
      for (int k = 0; k < OB; ++k) {
        const float m = p.mi2[o0 + k], is = p.mi2[24 + o0 + k], g = p.g2[o0 + k], be = p.b2[o0 + k];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float z = (acc[k][h] - m) * is * g + be;
          v[k][h] = in[h] ? (z > 0.f ? z : 0.1f * z) : 0.f;
        }
        rs[qq[0] * RP + o0 + k] = v[k][0];
        rs[qq[1] * RP + o0 + k] = v[k][1];
      }
      if (p.y2 != nullptr) {
        if constexpr (Y2S) {
#pragma unroll
          for (int k = 0; k < OB; ++k) {
            if (interior[0]) y2s[ypos[0] * 24 + o0 + k] = acc[k][0];
            if (interior[1]) y2s[ypos[1] * 24 + o0 + k] = acc[k][1];
          }
        } else {
          if (interior[0]) *reinterpret_cast<f32x4*>(p.y2 + pix0 * 24 + o0) = f32x4{acc[0][0], acc[1 % OB][0], acc[2 % OB][0], acc[3 % OB][0]};
          if (interior[1]) *reinterpret_cast<f32x4*>(p.y2 + pix1 * 24 + o0) = f32x4{acc[0][1], acc[1 % OB][1], acc[2 % OB][1], acc[3 % OB][1]};
        }
      }
      // channel sum with the grouping of chansum_kernel: ((c0 + c1) + (c2 + c3)) per group of four, groups added in order
      if (OB == 4) {
        sum += (v[0] + v[1]) + (v[2 % OB] + v[3 % OB]);
      } else {
#pragma unroll
        for (int k = 0; k < OB; ++k) sum += v[k];
      }
    }
    cs[qq[0]] = sum[0];
    cs[qq[1]] = sum[1];
    if (p.chansum != nullptr) {
      if (interior[0]) p.chansum[pix0] = sum[0];
      if (interior[1]) p.chansum[pix1] = sum[1];
    }
  }
  }      // (!FROMY2)
  __syncthreads();
  if (Y2S && !FROMY2 && p.y2 != nullptr) {
#pragma unroll
    for (int e = t; e < TS * TS * 6; e += 256) {
      const int pos = e / 6, g4 = (e - pos * 6) * 4, ih = h0 + pos / TS, iw = w0 + pos % TS;
      if (ih < H && iw < W)
        *reinterpret_cast<f32x4*>(p.y2 + (((int64_t)n * H + ih) * W + iw) * 24 + g4) = *reinterpret_cast<const f32x4*>(y2s + pos * 24 + g4);
    }
  }
  // ---- t = r + factor * stencil(chansum) + bias on the 18x18 region; zero outside the image
#pragma unroll 1
  for (int q = t; q < TT * TT; q += 256) {
    asm volatile("" ::: "memory");
    const int i = q / TT, j = q % TT, ih = h0 - 1 + i, iw = w0 - 1 + j;
    const bool in = ih >= 0 && iw >= 0 && ih < H && iw < W;
    float nb[9], st[8];
#pragma unroll
    for (int k = 0; k < 9; ++k) nb[k] = cs[(i + k / 3) * RS + j + k % 3];
    stencils8(nb, st);
    float* rsrc = rs + ((i + 1) * RS + j + 1) * RP;      // r of this position in, t of this position out
#pragma unroll
    for (int o = 0; o < 24; o += 4) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(rsrc + o);
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = in ? rv[k] + p.factor[o + k] * st[(o + k) & 7] + p.sbias[o + k] : 0.f;
      *reinterpret_cast<f32x4*>(rsrc + o) = v;
    }
  }
  __syncthreads();
  if (p.t != nullptr) {      // t of the tile's own 16x16 positions, for the backward: coalesced, as y2 above
#pragma unroll
    for (int e = t; e < TS * TS * 6; e += 256) {
      const int pos = e / 6, g4 = (e - pos * 6) * 4, pi = pos / TS, pj = pos % TS, ih = h0 + pi, iw = w0 + pj;
      if (ih < H && iw < W)
        *reinterpret_cast<f32x4*>(p.t + (((int64_t)n * H + ih) * W + iw) * 24 + g4) =
            *reinterpret_cast<const f32x4*>(rs + ((pi + 2) * RS + pj + 2) * RP + g4);
    }
  }
  // ---- y3 = conv3(t) on the 16x16 tile (tap-major, channel-minor: the order of smallconv_kernel<24,3>)
  const int ti = t >> 4, tj = t & 15, oh = h0 + ti, ow = w0 + tj;
  const bool live = oh < H && ow < W;
  // (even / odd input channels accumulate in the two halves of a float2 -- v_pk_fma_f32 with an SGPR pair of adjacent weights --
  //  and are added at the end)
  f32x2 a2[3] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll 1          // (fully unrolled, the 54 ds_read_b128 of the nine taps are all issued up front: 216 VGPRs)
  for (int tap = 0; tap < 9; ++tap) {
    const float* src = rs + ((ti + 1 + tap / 3) * RS + tj + 1 + tap % 3) * RP;      // t on the 18x18 interior of the 20x20 grid
#pragma unroll
    for (int c = 0; c < 24; c += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        const float* wp = p.w3 + (o * 9 + tap) * 24 + c;
        a2[o] += f32x2{v[0], v[1]} * f32x2{wp[0], wp[1]};
        a2[o] += f32x2{v[2], v[3]} * f32x2{wp[2], wp[3]};
      }
    }
  }
  const float a3[3] = {a2[0][0] + a2[0][1], a2[1][0] + a2[1][1], a2[2][0] + a2[2][1]};
  if (live) {
    float* dst = p.y3 + (((int64_t)n * H + oh) * W + ow) * 3;
    dst[0] = a3[0];
    dst[1] = a3[1];
    dst[2] = a3[2];
  }
  if (p.stat_part != nullptr) {
    const int lane = t & 63, wv = t >> 6;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float s1 = live ? a3[o] : 0.f, s2 = s1 * s1;
#pragma unroll
      for (int k = 32; k > 0; k >>= 1) {
        s1 += __shfl_xor(s1, k);
        s2 += __shfl_xor(s2, k);
      }
      if (lane == 0) {
        red[0][wv][o] = s1;
        red[1][wv][o] = s2;
      }
    }
    __syncthreads();
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (t < 6) {
      const int sidx = t / 3, o = t - sidx * 3;
      p.stat_part[((int64_t)blk * 2 + sidx) * 3 + o] = red[sidx][0][o] + red[sidx][1][o] + red[sidx][2][o] + red[sidx][3][o];
    }
  }
}

// ---- training forward, first half: y2 = conv2(x) computed ONCE -- stored for the backward and for the second half, with BN2's
// batch-statistics partials on the way.  (Rounds 2-4 recomputed conv2: a statistics pre-pass that stored nothing, then again on
// the 20x20 halo region of every 16x16 tile of the fused kernel -- 2.56 evaluations of the module's most expensive layer per
// pixel.)  A workgroup walks 32x16 tiles; a thread owns the positions (i, j) and (i + 8, j) of a tile as the halves of float2
// accumulators (v_pk_fma_f32, the weight one SGPR broadcast to both halves), four output channels per group so that a position's
// group leaves as one 16-byte store, and keeps the running sum / sum of squares of its positions for all 24 channels in registers
// across its tiles: the 48 wave reductions happen once per workgroup, not once per tile (they were 288 ds_bpermute per wave and
// tile in smallconv_kernel<3, 24>).  Per-position sums keep that kernel's tap-major / channel-minor order: y2 is bit-identical.
constexpr int C2W = 32, C2H = 16;
// (two waves per SIMD, 202 VGPRs: capped at 168 for three, the kernel spills 200 B in the channel loop and the training forward goes
//  from 0.87 to 1.10 ms)
template <bool STAGE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void cem_conv2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w2,
                                                            float* __restrict__ y2, float* __restrict__ stat_part, int ldx, int N,
                                                            int H, int W) {
  constexpr int XW = C2W + 2, XH = C2H + 2, XN = XW * XH * 3;
  // STAGE: y2 leaves through LDS as contiguous 3 KB runs.  From registers a position's four channels are one 16-byte store, but the
  // lanes of a store lie 96 B apart and a position's six groups are stored at six different times: the PMC counted 994 MB written
  // for 629 MB of y2 (profiles/r04_pmc_cem_module_after.txt).  The kernel runs two workgroups per CU either way (202 VGPRs).
  constexpr int YP = 28;
  __shared__ float xs[XN];
  __shared__ __align__(16) float y2s[STAGE ? C2W * C2H * YP : 4];
  __shared__ float red[2][4][24];
  const int t = threadIdx.x, i0 = t >> 5, j = t & 31;
  const int tw = (W + C2W - 1) / C2W, th = (H + C2H - 1) / C2H, ntiles = tw * th * N;
  float s1[24], s2[24];
#pragma unroll
  for (int o = 0; o < 24; ++o) s1[o] = s2[o] = 0.f;
#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = tile / (tw * th), rt = tile - n * tw * th, h0 = (rt / tw) * C2H, w0 = (rt % tw) * C2W;
    __syncthreads();                     // the previous tile's patches have been read
    {
      constexpr int NX = (XN + 255) / 256;
      float xv[NX];
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const int e = t + u * 256, c = e % 3, q = e / 3, ih = h0 - 1 + q / XW, iw = w0 - 1 + q % XW;
        xv[u] = (e < XN && ih >= 0 && iw >= 0 && ih < H && iw < W) ? x[(((int64_t)n * H + ih) * W + iw) * ldx + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < NX; ++u)
        if (t + u * 256 < XN) xs[t + u * 256] = xv[u];
    }
    __syncthreads();
    f32x2 xin[27];
    {
      const float* sp0 = xs + (i0 * XW + j) * 3;
      const float* sp1 = xs + ((i0 + 8) * XW + j) * 3;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int off = ((tap / 3) * XW + tap % 3) * 3 + c;
          xin[tap * 3 + c] = f32x2{sp0[off], sp1[off]};
        }
    }
    const bool live0 = h0 + i0 < H && w0 + j < W, live1 = h0 + i0 + 8 < H && w0 + j < W;
    float* dst0 = y2 + (((int64_t)n * H + h0 + i0) * W + w0 + j) * 24;
    float* dst1 = dst0 + (int64_t)8 * W * 24;
#pragma unroll
    for (int o0 = 0; o0 < 24; o0 += 4) {
      // (the weight address is laundered per group: as a loop invariant hipcc fetched all 648 weights ahead of the tile loop and
      //  kept them in VGPR lanes -- 1700 v_readlane per tile)
      int zo = 0;
      asm volatile("" : "+s"(zo));          // (an offset, not the pointer: a laundered pointer loses its scalar-load path)
      const float* wg = w2 + o0 * 27 + zo;
      f32x2 acc[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
      for (int e = 0; e < 27; ++e)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float w = wg[k * 27 + e];
          acc[k] += xin[e] * f32x2{w, w};
        }
      if constexpr (STAGE) {
        *reinterpret_cast<f32x4*>(y2s + t * YP + o0) = f32x4{acc[0][0], acc[1][0], acc[2][0], acc[3][0]};
        *reinterpret_cast<f32x4*>(y2s + (t + 256) * YP + o0) = f32x4{acc[0][1], acc[1][1], acc[2][1], acc[3][1]};
      } else {
        if (live0) *reinterpret_cast<f32x4*>(dst0 + o0) = f32x4{acc[0][0], acc[1][0], acc[2][0], acc[3][0]};
        if (live1) *reinterpret_cast<f32x4*>(dst1 + o0) = f32x4{acc[0][1], acc[1][1], acc[2][1], acc[3][1]};
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float m0 = live0 ? acc[k][0] : 0.f, m1 = live1 ? acc[k][1] : 0.f;
        s1[o0 + k] += m0 + m1;
        s2[o0 + k] += m0 * m0 + m1 * m1;
      }
    }
    if constexpr (STAGE) {
      __syncthreads();
#pragma unroll
      for (int e = t; e < C2W * C2H * 6; e += 256) {
        const int pos = e / 6, g4 = (e - pos * 6) * 4, ih = h0 + pos / C2W, iw = w0 + pos % C2W;
        if (ih < H && iw < W)
          *reinterpret_cast<f32x4*>(y2 + (((int64_t)n * H + ih) * W + iw) * 24 + g4) = *reinterpret_cast<const f32x4*>(y2s + pos * YP + g4);
      }
    }
  }
  const int lane = t & 63, wv = t >> 6;
#pragma unroll
  for (int o = 0; o < 24; ++o) {
    float a = s1[o], b = s2[o];
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) {
      a += __shfl_xor(a, k);
      b += __shfl_xor(b, k);
    }
    if (lane == 0) {
      red[0][wv][o] = a;
      red[1][wv][o] = b;
    }
  }
  __syncthreads();
  if (t < 48) {
    const int sidx = t / 24, o = t - sidx * 24;
    stat_part[((int64_t)blockIdx.x * 2 + sidx) * 24 + o] = red[sidx][0][o] + red[sidx][1][o] + red[sidx][2][o] + red[sidx][3][o];
  }
}

// backward pass 1: D[pix][k] = sum_{o%8==k} factor[o]*dt[pix][o]; partial sums of dbias[o] = sum dt_o and
// dfactor[o] = sum dt_o * stencil_{o%8}(R): partials[block][2][C].  A thread walks SOBEL_PPT pixels (256 apart, so a wave
// still reads consecutive pixels) before the 48 wave reductions, which would otherwise cost more than the HBM traffic.
constexpr int SOBEL_PPT = 8;
template <int C>
__global__ __launch_bounds__(256) void sobel_bwd1_kernel(const float* __restrict__ dt, int ldd, const float* __restrict__ R,
                                                         const float* __restrict__ factor, float* __restrict__ D,
                                                         float* __restrict__ partials, int N, int H, int W) {
  __shared__ float red[2][4][C];
  const int64_t npix = (int64_t)N * H * W;
  float db[C], df[C], fac[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    db[c] = df[c] = 0.f;
    fac[c] = factor[c];
  }
  for (int it = 0; it < SOBEL_PPT; ++it) {
    const int64_t p = ((int64_t)blockIdx.x * SOBEL_PPT + it) * 256 + threadIdx.x;
    if (p >= npix) break;
    const int w = (int)(p % W), h = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    float nb[9], s[8], dk[8];
    load9(R, n, h, w, H, W, nb);
    stencils8(nb, s);
#pragma unroll
    for (int k = 0; k < 8; ++k) dk[k] = 0.f;
    const float* src = dt + p * ldd;
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        db[c + k] += v[k];
        df[c + k] += v[k] * s[(c + k) & 7];
        dk[(c + k) & 7] += fac[c + k] * v[k];
      }
    }
    f32x4 d0 = {dk[0], dk[1], dk[2], dk[3]}, d1 = {dk[4], dk[5], dk[6], dk[7]};
    *reinterpret_cast<f32x4*>(D + p * 8) = d0;
    *reinterpret_cast<f32x4*>(D + p * 8 + 4) = d1;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float a = db[c], b = df[c];
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) {
      a += __shfl_xor(a, k);
      b += __shfl_xor(b, k);
    }
    if (lane == 0) {
      red[0][wv][c] = a;
      red[1][wv][c] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * C) {
    const int s = threadIdx.x / C, c = threadIdx.x - s * C;
    partials[((int64_t)blockIdx.x * 2 + s) * C + c] = red[s][0][c] + red[s][1][c] + red[s][2][c] + red[s][3][c];
  }
}

// backward pass 2: dr[pix][c] = dt[pix][c] + sum_k stencil_k^T(D_k)(pix)
template <int C>
__global__ void sobel_bwd2_kernel(const float* __restrict__ dt, int ldd, const float* __restrict__ D, float* __restrict__ dr,
                                  int lddr, int N, int H, int W) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (int64_t)N * H * W) return;
  const int w = (int)(p % W), h = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
  float dn[8][9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int ih = h + q / 3 - 1, iw = w + q % 3 - 1;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (ih >= 0 && iw >= 0 && ih < H && iw < W) {
      const float* src = D + (((int64_t)n * H + ih) * W + iw) * 8;
      a = *reinterpret_cast<const f32x4*>(src);
      b = *reinterpret_cast<const f32x4*>(src + 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dn[k][q] = a[k];
      dn[4 + k][q] = b[k];
    }
  }
  const float add = stencils8_t(dn);
  const float* src = dt + p * ldd;
  float* dst = dr + p * lddr;
#pragma unroll
  for (int c = 0; c < C; c += 4) {
    f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += add;
    *reinterpret_cast<f32x4*>(dst + c) = v;
  }
}

// ---- fused middle of the CEM backward (round 2): dy3 -> dt = conv3^T(dy3) -> stencil-bank backward -> dr, one 16x16 tile at
// a time with dt (18x18x24) and the eight D maps (18x18x8) in LDS: replaces the transposed smallconv_kernel (writes dt),
// sobel_bwd1 (reads dt, writes D) and sobel_bwd2 (reads dt and D, writes dr) -- 1.26 GB of dt and 0.42 GB of D never touch HBM.
// dt is computed with packed fp32 FMAs, two positions per thread (as the forward's conv2), the weights of OB output channels
// per scalar fetch.  The workgroups walk the tiles grid-stride and keep their dbias / dfactor partial sums in registers, so the
// 48 wave reductions happen once per workgroup (partials[block][2][24] -> mmi_pair_colsum, as sobel_bwd1).
// dt is zero outside the image (those positions hold no stencil output in the forward), dy3 likewise (zero padding).
// BN = true additionally folds BatchNorm2's backward REDUCTION into the last phase: with y2 (conv2's raw output) and BN2's
// mean / invstd / gamma / beta, dz = dr * LeakyReLU'(z) and the per-channel sums of dz and dz * xhat go to bn_partials[block][2][24]
// (the layout mmi_bn_act_bwd_apply folds), so bn_bwd_reduce's pass over y2 and dr (1.26 GB) disappears.
template <int OB, bool BN>
__global__ __launch_bounds__(256) void cem_bwd_mid_kernel(const float* __restrict__ dy3, const float* __restrict__ w3,
                                                          const float* __restrict__ R, const float* __restrict__ factor,
                                                          float* __restrict__ dr, float* __restrict__ partials,
                                                          const float* __restrict__ y2, const float* __restrict__ mi2,
                                                          const float* __restrict__ g2, const float* __restrict__ b2,
                                                          float* __restrict__ bn_partials, int N, int H, int W) {
  constexpr int YS = TS + 4, DT = TS + 2;          // 20, 18
  __shared__ float dys[YS * YS * 3];
  __shared__ __align__(16) float dts[DT * DT * 24];
  __shared__ __align__(16) float Ds[DT * DT * 8];
  __shared__ float Rs[DT * DT];
  __shared__ float red[2][4][24];
  const int t = threadIdx.x;
  const int tw = (W + TS - 1) / TS, th = (H + TS - 1) / TS, ntiles = tw * th * N;
  float db[24], df[24], s1[BN ? 24 : 1], s2[BN ? 24 : 1];
#pragma unroll
  for (int c = 0; c < 24; ++c) db[c] = df[c] = 0.f;
#pragma unroll
  for (int c = 0; c < (BN ? 24 : 1); ++c) s1[c] = s2[c] = 0.f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = tile / (tw * th), rr = tile - n * tw * th;
    const int h0 = (rr / tw) * TS, w0 = (rr % tw) * TS;
    __syncthreads();
    for (int e = t; e < YS * YS * 3; e += 256) {
      const int c = e % 3, q = e / 3, ih = h0 - 2 + q / YS, iw = w0 - 2 + q % YS;
      dys[e] = (ih >= 0 && iw >= 0 && ih < H && iw < W) ? dy3[(((int64_t)n * H + ih) * W + iw) * 3 + c] : 0.f;
    }
    for (int q = t; q < DT * DT; q += 256) {
      const int ih = h0 - 1 + q / DT, iw = w0 - 1 + q % DT;
      Rs[q] = (ih >= 0 && iw >= 0 && ih < H && iw < W) ? R[((int64_t)n * H + ih) * W + iw] : 0.f;
    }
    __syncthreads();
    // ---- dt on the 18x18 region, two positions (q, q + 162) per thread
    if (t < DT * DT / 2) {
      int qq[2];
      bool in[2];
      f32x2 xin[27];
      {
        const float* sp[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          qq[h] = t + h * (DT * DT / 2);
          const int i = qq[h] / DT, j = qq[h] % DT, ih = h0 - 1 + i, iw = w0 - 1 + j;
          in[h] = ih >= 0 && iw >= 0 && ih < H && iw < W;
          sp[h] = dys + ((i + 2) * YS + j + 2) * 3;          // the position itself in dys; tap (ky, kx) reads (i + 2 - ky, j + 2 - kx)
        }
#pragma unroll
        for (int tt = 0; tt < 9; ++tt) {                     // e = tt * 3 + o with tap = 8 - tt: the order of the transposed smallconv_kernel
          const int ky = (8 - tt) / 3, kx = (8 - tt) % 3;
#pragma unroll
          for (int o = 0; o < 3; ++o) {
            const int off = -(ky * YS + kx) * 3 + o;
            xin[tt * 3 + o] = f32x2{sp[0][off], sp[1][off]};
          }
        }
      }
#pragma unroll 1
      for (int c0 = 0; c0 < 24; c0 += OB) {
        f32x2 acc[OB];
#pragma unroll
        for (int k = 0; k < OB; ++k) acc[k] = f32x2{0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < 9; ++tt)
#pragma unroll
          for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int k = 0; k < OB; ++k) {
              const float w = w3[(o * 9 + (8 - tt)) * 24 + c0 + k];
              acc[k] += xin[tt * 3 + o] * f32x2{w, w};
            }
#pragma unroll
        for (int k = 0; k < OB; ++k) {
          dts[qq[0] * 24 + c0 + k] = in[0] ? acc[k][0] : 0.f;
          dts[qq[1] * 24 + c0 + k] = in[1] ? acc[k][1] : 0.f;
        }
      }
    }
    __syncthreads();
    // ---- D_k = sum_{o % 8 == k} factor[o] * dt_o on the 18x18 region
    for (int q = t; q < DT * DT; q += 256) {
      float dk[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) dk[k] = 0.f;
#pragma unroll
      for (int c = 0; c < 24; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(dts + q * 24 + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) dk[(c + k) & 7] += factor[c + k] * v[k];
      }
      *reinterpret_cast<f32x4*>(Ds + q * 8) = f32x4{dk[0], dk[1], dk[2], dk[3]};
      *reinterpret_cast<f32x4*>(Ds + q * 8 + 4) = f32x4{dk[4], dk[5], dk[6], dk[7]};
    }
    __syncthreads();
    // ---- the tile's own 16x16 positions: dr = dt + sum_k stencil_k^T(D_k); dbias / dfactor sums
    const int ti = t >> 4, tj = t & 15, oh = h0 + ti, ow = w0 + tj;
    if (oh < H && ow < W) {
      const int q = (ti + 1) * DT + tj + 1;
      float dn[8][9], nb[9], st[8];
#pragma unroll
      for (int d = 0; d < 9; ++d) {
        const int qn = q + (d / 3 - 1) * DT + d % 3 - 1;
        const f32x4 a = *reinterpret_cast<const f32x4*>(Ds + qn * 8), b = *reinterpret_cast<const f32x4*>(Ds + qn * 8 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          dn[k][d] = a[k];
          dn[4 + k][d] = b[k];
        }
        nb[d] = Rs[qn];
      }
      const float add = stencils8_t(dn);
      stencils8(nb, st);
      const int64_t pix = ((int64_t)n * H + oh) * W + ow;
      float* dst = dr + pix * 24;
#pragma unroll
      for (int c = 0; c < 24; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(dts + q * 24 + c);
        const f32x4 o4 = f32x4{v[0] + add, v[1] + add, v[2] + add, v[3] + add};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          db[c + k] += v[k];
          df[c + k] += v[k] * st[(c + k) & 7];
        }
        *reinterpret_cast<f32x4*>(dst + c) = o4;
        if constexpr (BN) {
          const f32x4 yy = *reinterpret_cast<const f32x4*>(y2 + pix * 24 + c);
#pragma unroll
          for (int k = 0; k < 4; ++k) {        // (the arithmetic of bn_bwd_reduce_kernel)
            const float xh = (yy[k] - mi2[c + k]) * mi2[24 + c + k];
            const float z = xh * g2[c + k] + b2[c + k];
            const float dz = o4[k] * (z > 0.f ? 1.0f : 0.1f);
            s1[c + k] += dz;
            s2[c + k] += dz * xh;
          }
        }
      }
    }
  }
  const int lane = t & 63, wv = t >> 6;
#pragma unroll
  for (int c = 0; c < 24; ++c) {
    float a = db[c], b = df[c];
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) {
      a += __shfl_xor(a, k);
      b += __shfl_xor(b, k);
    }
    if (lane == 0) {
      red[0][wv][c] = a;
      red[1][wv][c] = b;
    }
  }
  __syncthreads();
  if (t < 48) {
    const int sidx = t / 24, c = t - sidx * 24;
    partials[((int64_t)blockIdx.x * 2 + sidx) * 24 + c] = red[sidx][0][c] + red[sidx][1][c] + red[sidx][2][c] + red[sidx][3][c];
  }
  if constexpr (BN) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 24; ++c) {
      float a = s1[c], b = s2[c];
#pragma unroll
      for (int k = 32; k > 0; k >>= 1) {
        a += __shfl_xor(a, k);
        b += __shfl_xor(b, k);
      }
      if (lane == 0) {
        red[0][wv][c] = a;
        red[1][wv][c] = b;
      }
    }
    __syncthreads();
    if (t < 48) {
      const int sidx = t / 24, c = t - sidx * 24;
      bn_partials[((int64_t)blockIdx.x * 2 + sidx) * 24 + c] = red[sidx][0][c] + red[sidx][1][c] + red[sidx][2][c] + red[sidx][3][c];
    }
  }
}

}  // namespace

// ---- dispatch helpers used by igemm.hip's public conv entry points ------------------------------------------------------
bool mmi_smallconv_supported(const mmi_conv_desc* d) {
  const bool a = d->Cin == 3 && d->Cout == 24, b = d->Cin == 24 && d->Cout == 3;
  return (a || b) && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && (d->Cin != 24 || d->ldx % 4 == 0);
}
int mmi_smallconv_blocks(const mmi_conv_desc* d) { return cdiv(d->W, TS) * cdiv(d->H, TS) * d->N; }

int mmi_smallconv_fwd(const float* x, const float* w, const float* bias, float* y, float* stat_part, const mmi_conv_desc* d,
                      hipStream_t s) {
  const dim3 grid(cdiv(d->W, TS), cdiv(d->H, TS), d->N), block(256);
  if (d->Cin == 3)
    hipLaunchKernelGGL((smallconv_kernel<3, 24, false>), grid, block, 0, s, x, d->ldx, w, bias, y, d->ldy, stat_part, d->H, d->W);
  else
    hipLaunchKernelGGL((smallconv_kernel<24, 3, false>), grid, block, 0, s, x, d->ldx, w, bias, y, d->ldy, stat_part, d->H, d->W);
  MMI_CHECK_LAUNCH("mmi_conv_fwd(small-channel)");
  return MMI_OK;
}

// dx (Cin channels) from dy (Cout channels): only the 24-channel input needs it (the 3-channel one is the image)
bool mmi_smallconv_dgrad_supported(const mmi_conv_desc* d) { return mmi_smallconv_supported(d) && d->Cin == 24; }
int mmi_smallconv_dgrad(const float* dy, const float* w, float* dx, const mmi_conv_desc* d, hipStream_t s) {
  const dim3 grid(cdiv(d->W, TS), cdiv(d->H, TS), d->N), block(256);
  hipLaunchKernelGGL((smallconv_kernel<3, 24, true>), grid, block, 0, s, dy, d->ldy, w, (const float*)nullptr, dx, d->ldx,
                     (float*)nullptr, d->H, d->W);
  MMI_CHECK_LAUNCH("mmi_conv_dgrad(small-channel)");
  return MMI_OK;
}

constexpr int WG_BLOCKS = 2048;
size_t mmi_smallconv_wgrad_workspace(const mmi_conv_desc* d) {
  const int blocks = mmi_smallconv_blocks(d) < WG_BLOCKS ? mmi_smallconv_blocks(d) : WG_BLOCKS;
  return (size_t)blocks * d->Cout * 9 * d->Cin * sizeof(float);
}
int mmi_smallconv_wgrad(const float* dy, const float* x, float* dw, void* workspace, const mmi_conv_desc* d, hipStream_t s) {
  const int blocks = mmi_smallconv_blocks(d) < WG_BLOCKS ? mmi_smallconv_blocks(d) : WG_BLOCKS;
  const int nout = d->Cout * 9 * d->Cin;
  float* part = (float*)workspace;
  if (d->Cin == 3) {      // (VEC: the 24-channel operand fetched with 16-byte loads)
    if (d->ldy % 4 == 0 && ((uintptr_t)dy & 15) == 0)
      hipLaunchKernelGGL((smallconv_wgrad_kernel<3, 24, true>), dim3(blocks), dim3(256), 0, s, x, d->ldx, dy, d->ldy, part, d->N, d->H, d->W, BnApply{});
    else
      hipLaunchKernelGGL((smallconv_wgrad_kernel<3, 24, false>), dim3(blocks), dim3(256), 0, s, x, d->ldx, dy, d->ldy, part, d->N, d->H, d->W, BnApply{});
  } else {
    if (d->ldx % 4 == 0 && ((uintptr_t)x & 15) == 0)
      hipLaunchKernelGGL((smallconv_wgrad_kernel<24, 3, true>), dim3(blocks), dim3(256), 0, s, x, d->ldx, dy, d->ldy, part, d->N, d->H, d->W, BnApply{});
    else
      hipLaunchKernelGGL((smallconv_wgrad_kernel<24, 3, false>), dim3(blocks), dim3(256), 0, s, x, d->ldx, dy, d->ldy, part, d->N, d->H, d->W, BnApply{});
  }
  MMI_CHECK_LAUNCH("mmi_conv_wgrad(small-channel)");
  hipLaunchKernelGGL(rowsum_kernel, dim3(cdiv(nout, 16)), dim3(256), 0, s, (const float*)part, blocks, nout, dw);
  MMI_CHECK_LAUNCH("mmi_conv_wgrad(small-channel reduce)");
  return MMI_OK;
}

// conv2's weight gradient with BatchNorm2 + LeakyReLU's backward applied on the way in (BNF above): dw2 = wgrad(x, dy2(dr, y2)).
// dgamma2 / dbeta2 are the finished sums (mmi_bn_act_bwd with dy = NULL, same stream, before this call).
extern "C" size_t mmi_cem_conv2_wgrad_bn_workspace(int N, int H, int W) {
  const int tiles = N * cdiv(H, TS) * cdiv(W, TS);
  return (size_t)(tiles < WG_BLOCKS ? tiles : WG_BLOCKS) * 24 * 9 * 3 * sizeof(float);
}
extern "C" int mmi_cem_conv2_wgrad_bn(const float* dr, const float* y2, const float* x, int ldx, const float* mean_invstd2,
                                      const float* gamma2, const float* beta2, const float* dgamma2, const float* dbeta2, int frozen,
                                      float* dw2, void* workspace, size_t workspace_bytes, int N, int H, int W, void* stream) {
  MMI_CHECK_ARG(dr && y2 && x && mean_invstd2 && gamma2 && beta2 && dgamma2 && dbeta2 && dw2 && workspace && N > 0 && H > 0 && W > 0 &&
                    ldx >= 3, "mmi_cem_conv2_wgrad_bn: bad arguments");
  MMI_CHECK_ARG((((uintptr_t)dr | (uintptr_t)y2) & 15) == 0, "mmi_cem_conv2_wgrad_bn: dr / y2 must be 16-byte aligned");
  MMI_CHECK_ARG(workspace_bytes >= mmi_cem_conv2_wgrad_bn_workspace(N, H, W), "mmi_cem_conv2_wgrad_bn: workspace too small");
  const int tiles = N * cdiv(H, TS) * cdiv(W, TS), blocks = tiles < WG_BLOCKS ? tiles : WG_BLOCKS;
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  const BnApply bn{y2, mean_invstd2, gamma2, beta2, dgamma2, dbeta2, 1.0f / (float)((int64_t)N * H * W), frozen};
  hipLaunchKernelGGL((smallconv_wgrad_kernel<3, 24, true, true>), dim3(blocks), dim3(256), 0, s, x, ldx, dr, 24, part, N, H, W, bn);
  MMI_CHECK_LAUNCH("mmi_cem_conv2_wgrad_bn");
  hipLaunchKernelGGL(rowsum_kernel, dim3(cdiv(24 * 27, 16)), dim3(256), 0, s, (const float*)part, blocks, 24 * 27, dw2);
  MMI_CHECK_LAUNCH("mmi_cem_conv2_wgrad_bn(reduce)");
  return MMI_OK;
}

// ---- public: the stencil bank as an elementwise op -----------------------------------------------------------------------
extern "C" int mmi_sobel_add_fwd(const float* r, int ldr, const float* factor, const float* bias, float* chansum, float* t,
                                 int ldt, int N, int H, int W, int C, void* stream) {
  MMI_CHECK_ARG(r && factor && bias && chansum && t && N > 0 && H > 0 && W > 0, "mmi_sobel_add_fwd: bad arguments");
  MMI_CHECK_ARG(C == 24 && ldr % 4 == 0 && ldt % 4 == 0, "mmi_sobel_add_fwd: the stencil bank is the 24-channel CEM one");
  hipStream_t s = (hipStream_t)stream;
  const int64_t npix = (int64_t)N * H * W;
  hipLaunchKernelGGL(chansum_kernel<24>, dim3(cdiv(npix, 256)), dim3(256), 0, s, r, ldr, chansum, npix);
  MMI_CHECK_LAUNCH("mmi_sobel_add_fwd(chansum)");
  hipLaunchKernelGGL(sobel_add_fwd_kernel<24>, dim3(cdiv(npix, 256)), dim3(256), 0, s, r, ldr, (const float*)chansum, factor,
                     bias, t, ldt, N, H, W);
  MMI_CHECK_LAUNCH("mmi_sobel_add_fwd");
  return MMI_OK;
}

extern "C" size_t mmi_sobel_add_bwd_workspace(int N, int H, int W, int C) {
  const int64_t npix = (int64_t)N * H * W;
  return (size_t)(npix * 8 + (int64_t)cdiv(npix, 256) * 2 * C) * sizeof(float);
}

extern "C" int mmi_sobel_add_bwd(const float* dt, int ldd, const float* chansum, const float* factor, float* dr, int lddr,
                                 float* dfactor, float* dbias, void* workspace, int N, int H, int W, int C, void* stream) {
  MMI_CHECK_ARG(dt && chansum && factor && dr && dfactor && dbias && workspace && N > 0 && H > 0 && W > 0,
                "mmi_sobel_add_bwd: bad arguments");
  MMI_CHECK_ARG(C == 24 && ldd % 4 == 0 && lddr % 4 == 0, "mmi_sobel_add_bwd: the stencil bank is the 24-channel CEM one");
  hipStream_t s = (hipStream_t)stream;
  const int64_t npix = (int64_t)N * H * W;
  const int blocks = cdiv(npix, 256), blocks1 = cdiv(npix, 256 * SOBEL_PPT);
  float* D = (float*)workspace;
  float* part = D + npix * 8;
  hipLaunchKernelGGL(sobel_bwd1_kernel<24>, dim3(blocks1), dim3(256), 0, s, dt, ldd, chansum, factor, D, part, N, H, W);
  MMI_CHECK_LAUNCH("mmi_sobel_add_bwd(1)");
  hipLaunchKernelGGL(sobel_bwd2_kernel<24>, dim3(blocks), dim3(256), 0, s, dt, ldd, (const float*)D, dr, lddr, N, H, W);
  MMI_CHECK_LAUNCH("mmi_sobel_add_bwd(2)");
  return mmi_pair_colsum(part, blocks1, C, dbias, dfactor, stream);  // partials[block][2][C] -> dbias (slot 0), dfactor (slot 1)
}

// ---- fused CEM forward: host side ------------------------------------------------------------------------------------
// Statistics pre-pass of BN2: conv2 recomputed, nothing stored, partials[blocks][2][24] (blocks = mmi_cem_blocks).
extern "C" int mmi_cem_blocks(int N, int H, int W) { return N * cdiv(H, TS) * cdiv(W, TS); }

extern "C" int mmi_cem_conv2_stats(const float* x, int ldx, const float* w2, float* stat_partials, int N, int H, int W, void* stream) {
  MMI_CHECK_ARG(x && w2 && stat_partials && N > 0 && H > 0 && W > 0 && ldx >= 3, "mmi_cem_conv2_stats: bad arguments");
  const dim3 grid(cdiv(W, TS), cdiv(H, TS), N);
  hipLaunchKernelGGL((smallconv_kernel<3, 24, false>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, w2, (const float*)nullptr,
                     (float*)nullptr, 24, stat_partials, H, W);
  MMI_CHECK_LAUNCH("mmi_cem_conv2_stats");
  return MMI_OK;
}

// x -> y3 (+ its statistics partials [blocks][2][3]) with y2, t, chansum written for the backward when non-NULL.
extern "C" int mmi_cem_fused_fwd(const float* x, int ldx, const float* w2, const float* mean_invstd2, const float* gamma2,
                                 const float* beta2, const float* factor, const float* sobel_bias, const float* w3, float* y2,
                                 float* t, float* chansum, float* y3, float* stat_partials3, int N, int H, int W, void* stream) {
  MMI_CHECK_ARG(x && w2 && mean_invstd2 && gamma2 && beta2 && factor && sobel_bias && w3 && y3 && N > 0 && H > 0 && W > 0 && ldx >= 3,
                "mmi_cem_fused_fwd: bad arguments");
  MMI_CHECK_ARG(((uintptr_t)y2 & 15) == 0 && ((uintptr_t)t & 15) == 0, "mmi_cem_fused_fwd: y2 / t must be 16-byte aligned");
  const CemOut out{y2, t, chansum, y3, stat_partials3};
  const dim3 grid(cdiv(W, TS), cdiv(H, TS), N);
  static const int ob = [] {      // output channels of conv2 per scalar-weight fetch (A/B switch)
    const char* e = getenv("MMIDET_CEM_OB");
    const int v = e ? atoi(e) : 2;
    return (v == 2 || v == 3 || v == 4) ? v : 2;
  }();
  // 1 (default): OB = 4, y2 from registers, 3 workgroups per CU; 0: y2 staged through LDS, 2 workgroups per CU.  At 16 x 640 x 640
  // (profiles/r04_cem_forward_forms.txt): eval forward 0.71 vs 0.80-0.93 ms, training forward 1.09 vs 1.21-1.34, module 2.97 vs 3.08-3.19
  static const int form = [] {
    const char* e = getenv("MMIDET_CEM_FORM");
    return e ? atoi(e) : 1;
  }();
#define CEM_LAUNCH(OB_, Y2S_, WPE_) hipLaunchKernelGGL((cem_fused_fwd_kernel<OB_, Y2S_, WPE_, false>), grid, dim3(256), 0, (hipStream_t)stream, x, w2, \
                                           mean_invstd2, gamma2, beta2, factor, sobel_bias, w3, out, ldx, H, W)
  if (form == 1) CEM_LAUNCH(4, false, 3);
  else if (ob == 2) CEM_LAUNCH(2, true, 2);
  else if (ob == 3) CEM_LAUNCH(3, true, 2);
  else CEM_LAUNCH(4, true, 2);
#undef CEM_LAUNCH
  MMI_CHECK_LAUNCH("mmi_cem_fused_fwd");
  return MMI_OK;
}

// Training forward in two launches (the default since round 4): y2 = conv2(x) with BN2's statistics partials
// [mmi_cem_conv2_fwd_blocks][2][24], then (after mmi_bn_finalize) everything from y2 to y3 with r and t in LDS.
extern "C" int mmi_cem_conv2_fwd_blocks(int N, int H, int W) {
  const int tiles = N * cdiv(H, C2H) * cdiv(W, C2W);
  return cdiv(tiles, cdiv(tiles, 1024));      // <= 1024 workgroups with equal shares of the tiles (to within one)
}

extern "C" int mmi_cem_conv2_fwd(const float* x, int ldx, const float* w2, float* y2, float* stat_partials, int N, int H, int W,
                                 void* stream) {
  MMI_CHECK_ARG(x && w2 && y2 && stat_partials && N > 0 && H > 0 && W > 0 && ldx >= 3, "mmi_cem_conv2_fwd: bad arguments");
  MMI_CHECK_ARG(((uintptr_t)y2 & 15) == 0, "mmi_cem_conv2_fwd: y2 must be 16-byte aligned");
  static const bool stage = !(getenv("MMIDET_CEM_C2_STAGE") && atoi(getenv("MMIDET_CEM_C2_STAGE")) == 0);      // A/B switch
  if (stage)
    hipLaunchKernelGGL(cem_conv2_fwd_kernel<true>, dim3(mmi_cem_conv2_fwd_blocks(N, H, W)), dim3(256), 0, (hipStream_t)stream, x, w2, y2,
                       stat_partials, ldx, N, H, W);
  else
    hipLaunchKernelGGL(cem_conv2_fwd_kernel<false>, dim3(mmi_cem_conv2_fwd_blocks(N, H, W)), dim3(256), 0, (hipStream_t)stream, x, w2, y2,
                       stat_partials, ldx, N, H, W);
  MMI_CHECK_LAUNCH("mmi_cem_conv2_fwd");
  return MMI_OK;
}

extern "C" int mmi_cem_fwd_from_y2(const float* y2, const float* mean_invstd2, const float* gamma2, const float* beta2,
                                   const float* factor, const float* sobel_bias, const float* w3, float* t, float* chansum, float* y3,
                                   float* stat_partials3, int N, int H, int W, void* stream) {
  MMI_CHECK_ARG(y2 && mean_invstd2 && gamma2 && beta2 && factor && sobel_bias && w3 && y3 && N > 0 && H > 0 && W > 0,
                "mmi_cem_fwd_from_y2: bad arguments");
  MMI_CHECK_ARG(((uintptr_t)y2 & 15) == 0 && ((uintptr_t)t & 15) == 0, "mmi_cem_fwd_from_y2: y2 / t must be 16-byte aligned");
  const CemOut out{const_cast<float*>(y2), t, chansum, y3, stat_partials3};
  const dim3 grid(cdiv(W, TS), cdiv(H, TS), N);
  hipLaunchKernelGGL((cem_fused_fwd_kernel<4, false, 3, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)nullptr,
                     (const float*)nullptr, mean_invstd2, gamma2, beta2, factor, sobel_bias, w3, out, 0, H, W);
  MMI_CHECK_LAUNCH("mmi_cem_fwd_from_y2");
  return MMI_OK;
}

// ---- fused middle of the CEM backward: host side -----------------------------------------------------------------------
// Workgroups of the persistent grid: 512 = two per CU.  Module time at 16 x 640 x 640 on one box (profiles/r04_cem_bwd_mid_grid.txt):
// 2.47 ms at 1024 (rounds 2-3), 2.45 at 768, 2.44 at 512, 2.47 at 384, 2.63 at 256.  (Also tried in round 4 and dropped: a 28-float
// LDS pitch for dt, the staging loads unrolled, dr staged through LDS for contiguous stores -- each neutral or slower here.)
static int cem_mid_blocks_max() {
  static const int v = [] {
    const char* e = getenv("MMIDET_CEM_MID_BLOCKS");
    const int n = e ? atoi(e) : 512;
    return n >= 64 && n <= 4096 ? n : 512;
  }();
  return v;
}
#define CEM_MID_BLOCKS cem_mid_blocks_max()
extern "C" size_t mmi_cem_bwd_mid_workspace(int N, int H, int W) {
  const int tiles = N * cdiv(H, TS) * cdiv(W, TS);
  return (size_t)(tiles < CEM_MID_BLOCKS ? tiles : CEM_MID_BLOCKS) * 2 * 24 * sizeof(float);
}

extern "C" int mmi_cem_bwd_mid_blocks(int N, int H, int W) {
  const int tiles = N * cdiv(H, TS) * cdiv(W, TS);
  return tiles < CEM_MID_BLOCKS ? tiles : CEM_MID_BLOCKS;
}

// y2 .. bn_partials: all NULL, or all set -- then bn_partials[mmi_cem_bwd_mid_blocks][2][24] receives BatchNorm2's backward sums
// (feed them to mmi_bn_act_bwd_apply(y2, dr, ..., bn_partials, blocks, ...)).
extern "C" int mmi_cem_bwd_mid(const float* dy3, const float* w3, const float* chansum, const float* factor, float* dr, float* dfactor,
                               float* dbias, void* workspace, const float* y2, const float* mean_invstd2, const float* gamma2,
                               const float* beta2, float* bn_partials, int N, int H, int W, void* stream) {
  MMI_CHECK_ARG(dy3 && w3 && chansum && factor && dr && dfactor && dbias && workspace && N > 0 && H > 0 && W > 0,
                "mmi_cem_bwd_mid: bad arguments");
  const bool bn = y2 != nullptr;
  MMI_CHECK_ARG(bn == (mean_invstd2 != nullptr) && bn == (gamma2 != nullptr) && bn == (beta2 != nullptr) && bn == (bn_partials != nullptr),
                "mmi_cem_bwd_mid: the BatchNorm operands come all or none");
  MMI_CHECK_ARG(((uintptr_t)dr & 15) == 0 && ((uintptr_t)y2 & 15) == 0, "mmi_cem_bwd_mid: dr / y2 must be 16-byte aligned");
  const int blocks = mmi_cem_bwd_mid_blocks(N, H, W);
  if (bn)
    hipLaunchKernelGGL((cem_bwd_mid_kernel<2, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy3, w3, chansum, factor, dr,
                       (float*)workspace, y2, mean_invstd2, gamma2, beta2, bn_partials, N, H, W);
  else
    hipLaunchKernelGGL((cem_bwd_mid_kernel<2, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy3, w3, chansum, factor, dr,
                       (float*)workspace, y2, mean_invstd2, gamma2, beta2, bn_partials, N, H, W);
  MMI_CHECK_LAUNCH("mmi_cem_bwd_mid");
  return mmi_pair_colsum((float*)workspace, blocks, 24, dbias, dfactor, stream);  // (slot 0 -> dbias, slot 1 -> dfactor; consumes the partials)
}
