// Training-mode BatchNorm + activation (+ residual) over NHWC rows, forward and backward.  HBM-bound kernels:
// 16-byte lane accesses along the contiguous channel axis, per-channel reductions as (rows-chunk x channel) partials
// that a tiny second kernel folds in fp64 (deterministic, no atomics, no memset).
//
// Replaces nn.BatchNorm2d + SiLU/LeakyReLU(0.1) (+ Bottleneck/CEM residual add) of the reference:
// models/common.py:116-122, 613, 766-767, 774-775, 799; eps/momentum from utils/torch_utils.py:149-151.
#include "common.h"

namespace {

// Sum partial rows [blockIdx.y*chunk, +chunk) of partials[part][2][C] in fp64: one block = 16 channels x 16 part-lanes
// (64-byte channel segments stay coalesced, a long part list is walked 16-wide).  Result in red[s][0][cl] of warp 0.
__device__ __forceinline__ void fold_parts(const float* partials, int p0, int p1, int C, int c, int cl, int pl,
                                           double (*red)[16][16], double& s1, double& s2, int64_t pstride = 1) {
  s1 = s2 = 0.0;
  if (c < C) {
    constexpr int U = 16;  // loads of one round are independent: a 1024-part list costs 4 memory latencies, not 64
    for (int p = p0 + pl; p < p1; p += 16 * U) {
      float a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pp = p + 16 * u;
        const bool ok = pp < p1;
        a[u] = ok ? partials[((int64_t)pp * pstride * 2 + 0) * C + c] : 0.0f;
        b[u] = ok ? partials[((int64_t)pp * pstride * 2 + 1) * C + c] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s1 += (double)a[u];
        s2 += (double)b[u];
      }
    }
  }
  red[0][pl][cl] = s1;
  red[1][pl][cl] = s2;
  __syncthreads();
  if (pl == 0) {
    s1 = s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s1 += red[0][i][cl];
      s2 += red[1][i][cl];
    }
  }
}

// stage 1 for long part lists: grid (C/16, Y) -> folded[y][2][C]
__global__ void stat_fold_kernel(const float* __restrict__ partials, int nparts, int C, float* __restrict__ folded, int chunk) {
  __shared__ double red[2][16][16];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int p0 = blockIdx.y * chunk, p1 = min(p0 + chunk, nparts);
  double s1, s2;
  fold_parts(partials, p0, p1, C, c, cl, pl, red, s1, s2);
  if (pl == 0 && c < C) {
    folded[((int64_t)blockIdx.y * 2 + 0) * C + c] = (float)s1;
    folded[((int64_t)blockIdx.y * 2 + 1) * C + c] = (float)s2;
  }
}

// partials[part][2][C] (fp32) -> mean / invstd, running-stat update.
__global__ void bn_finalize_kernel(const float* __restrict__ partials, int nparts, double rows, int C, float eps,
                                   float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                   int64_t* __restrict__ nbt, float* __restrict__ mean_invstd) {
  __shared__ double red[2][16][16];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s1, s2;
  fold_parts(partials, 0, nparts, C, c, cl, pl, red, s1, s2);
  if (nbt != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
  if (pl == 0 && c < C) {
    const double mean = s1 / rows;
    double var = s2 / rows - mean * mean;  // biased (normalisation) variance
    if (var < 0.0) var = 0.0;
    mean_invstd[c] = (float)mean;
    mean_invstd[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
      const double unb = rows > 1.0 ? var * rows / (rows - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
  }
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, int C, float eps,
                                     float* __restrict__ mean_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    mean_invstd[c] = rm[c];
    mean_invstd[C + c] = 1.0f / sqrtf(rv[c] + eps);
  }
}

template <int V>
struct Vec {
  float v[V];
};
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
// bf16 storage: the streaming kernels are templated on the activation type T (float or __bf16); the
// arithmetic, the statistics and the parameters are fp32 either way
template <int V>
__device__ __forceinline__ Vec<V> ldv(const __bf16* p) {
  Vec<V> r;
  if (V == 4) {
    const bf16x4_t t = *reinterpret_cast<const bf16x4_t*>(p);
    r.v[0] = (float)t[0]; r.v[1 % V] = (float)t[1]; r.v[2 % V] = (float)t[2]; r.v[3 % V] = (float)t[3];
  } else {
    r.v[0] = (float)p[0];
  }
  return r;
}
template <int V>
__device__ __forceinline__ void stv(__bf16* p, const Vec<V>& r) {
  if (V == 4) {
    bf16x4_t t = {(__bf16)r.v[0], (__bf16)r.v[1 % V], (__bf16)r.v[2 % V], (__bf16)r.v[3 % V]};
    *reinterpret_cast<bf16x4_t*>(p) = t;
  } else {
    p[0] = (__bf16)r.v[0];
  }
}
template <int V>
__device__ __forceinline__ Vec<V> ldv(const float* p) {
  Vec<V> r;
  if (V == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    r.v[0] = t[0]; r.v[1 % V] = t[1]; r.v[2 % V] = t[2]; r.v[3 % V] = t[3];
  } else {
    r.v[0] = p[0];
  }
  return r;
}
template <int V>
__device__ __forceinline__ void stv(float* p, const Vec<V>& r) {
  if (V == 4) {
    f32x4 t = {r.v[0], r.v[1 % V], r.v[2 % V], r.v[3 % V]};
    *reinterpret_cast<f32x4*>(p) = t;
  } else {
    p[0] = r.v[0];
  }
}

// Streaming kernels: when the grid stride is a multiple of the C/V column groups (FIXED; every power-of-two C of the
// model), a thread keeps its column group, so the per-channel parameters are loaded once and no 64-bit division runs per
// element; otherwise the general index arithmetic is used.
// Channel maps (mmi_bn_map, include/mmidet_hip.h).  The C channels of a conv output may belong to up to four BatchNorm
// modules (parameter blocks of `blk` channels: C3's merged cv1|cv2 conv has two, the twin RGB|IR form of it four), and the
// activated output may be scattered over two tensors: with lane = c / period and r = c % period, channel c goes to
// t0 + lane * ls0 + r when r < split, else to t1 + lane * ls1 + (r - split).  A plain layer is period = split = C; C3's merged
// conv hands its first half to the bottleneck chain and writes the second half straight into the concat buffer (split = C/2);
// the twin forms repeat that per lane.  The backward reads the incoming gradient through the same map.
__device__ __forceinline__ const float* sel4(const float* const (&a)[4], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3])); }
__device__ __forceinline__ float* sel4(float* const (&a)[4], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3])); }
struct ChanLoc {
  bool hi;
  int off;   // element offset inside t0 / t1 (row offset excluded)
};
__device__ __forceinline__ ChanLoc chan_loc(const mmi_bn_map& m, int c) {
  const int lane = c / m.period, r = c - lane * m.period;
  ChanLoc l;
  l.hi = r >= m.split;
  l.off = l.hi ? lane * m.ls1 + (r - m.split) : lane * m.ls0 + r;
  return l;
}
template <typename T, int V, bool FIXED>
__global__ void bn_act_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ mi, mmi_bn_map mp,
                                  const T* __restrict__ res, int ldr, T* __restrict__ out, int ldo,
                                  T* __restrict__ out1, int ldo1, int64_t rows, int C, int act) {
  const int cv = C / V;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (FIXED) {
    const int64_t r0 = e0 / cv, dr = stride / cv;
    const int c = (int)(e0 - r0 * cv) * V;
    const int pb = c / mp.blk, pc = c - pb * mp.blk;
    const Vec<V> m = ldv<V>(mi + c), is = ldv<V>(mi + C + c), g = ldv<V>(sel4(mp.gamma, pb) + pc), b = ldv<V>(sel4(mp.beta, pb) + pc);
    const T* py = y + r0 * ldy + c;
    const T* pr = res != nullptr ? res + r0 * ldr + c : nullptr;
    const ChanLoc loc = chan_loc(mp, c);
    const int ldo_ = loc.hi ? ldo1 : ldo;
    T* po = (loc.hi ? out1 : out) + loc.off + r0 * ldo_;
    const int64_t sy = dr * ldy, sr = dr * ldr, so = dr * ldo_;
#pragma unroll 2
    for (int64_t r = r0; r < rows; r += dr, py += sy, po += so) {
      const Vec<V> yy = ldv<V>(py);
      Vec<V> o;
#pragma unroll
      for (int k = 0; k < V; ++k) o.v[k] = act_fwd((yy.v[k] - m.v[k]) * is.v[k] * g.v[k] + b.v[k], act);
      if (pr != nullptr) {
        const Vec<V> rr = ldv<V>(pr);
        pr += sr;
#pragma unroll
        for (int k = 0; k < V; ++k) o.v[k] += rr.v[k];
      }
      stv<V>(po, o);
    }
    return;
  }
  const int64_t total = rows * cv;
  for (int64_t e = e0; e < total; e += stride) {
    const int64_t r = e / cv;
    const int c = (int)(e - r * cv) * V;
    const int pb = c / mp.blk, pc = c - pb * mp.blk;
    const Vec<V> yy = ldv<V>(y + r * ldy + c), m = ldv<V>(mi + c), is = ldv<V>(mi + C + c), g = ldv<V>(sel4(mp.gamma, pb) + pc),
                 b = ldv<V>(sel4(mp.beta, pb) + pc);
    Vec<V> o;
#pragma unroll
    for (int k = 0; k < V; ++k) o.v[k] = act_fwd((yy.v[k] - m.v[k]) * is.v[k] * g.v[k] + b.v[k], act);
    if (res != nullptr) {
      const Vec<V> rr = ldv<V>(res + r * ldr + c);
#pragma unroll
      for (int k = 0; k < V; ++k) o.v[k] += rr.v[k];
    }
    const ChanLoc loc = chan_loc(mp, c);
    stv<V>(loc.hi ? out1 + r * ldo1 + loc.off : out + r * ldo + loc.off, o);
  }
}

// backward pass 1: block = (64-channel group, rows part); threads = (64/V channel lanes) x row lanes.  dout may come in two
// tensors split at channel `split` (see bn_act_fwd_kernel).  With fold.cnt != null the workgroup that arrives last
// (stat_arrive, common.h) also produces dbeta / dgamma, so the pass needs no "finalize" launch behind it.
template <typename T, int V>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dout, int ldd,
                                     const T* __restrict__ dout1, int ldd1, const float* __restrict__ mi, mmi_bn_map mp,
                                     float* __restrict__ partials, int64_t rows, int C,
                                     int act, int64_t rows_per_part, StatFold fold) {
  constexpr int CL = 64 / V, RL = 256 / CL;
  __shared__ float red[2][RL][64];
  __shared__ double redd[512];
  __shared__ int flag;
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int c = blockIdx.x * 64 + cl * V;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_part;
  const int64_t r1 = min(r0 + rows_per_part, rows);
  float s1[V], s2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) s1[k] = s2[k] = 0.f;
  if (c < C) {
    const int pb = c / mp.blk, pc = c - pb * mp.blk;
    const Vec<V> m = ldv<V>(mi + c), is = ldv<V>(mi + C + c), g = ldv<V>(sel4(mp.gamma, pb) + pc), b = ldv<V>(sel4(mp.beta, pb) + pc);
    const ChanLoc loc = chan_loc(mp, c);
    const T* pd = (loc.hi ? dout1 : dout) + loc.off;
    const int ldd_ = loc.hi ? ldd1 : ldd;
    for (int64_t r = r0 + rl; r < r1; r += RL) {
      const Vec<V> yy = ldv<V>(y + r * ldy + c), dd = ldv<V>(pd + r * ldd_);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float xh = (yy.v[k] - m.v[k]) * is.v[k];
        const float dz = dd.v[k] * act_grad(xh * g.v[k] + b.v[k], act);
        s1[k] += dz;
        s2[k] += dz * xh;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) {
    red[0][rl][cl * V + k] = s1[k];
    red[1][rl][cl * V + k] = s2[k];
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int s = threadIdx.x >> 6, cc = threadIdx.x & 63;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < RL; ++i) acc += red[s][i][cc];
    const int col = blockIdx.x * 64 + cc;
    if (col < C) st_agent(partials + ((int64_t)blockIdx.y * 2 + s) * C + col, acc);
  }
  if (fold.cnt == nullptr) return;  // uniform
  double t1, t2;
  if (!stat_arrive<64>(fold, blockIdx.y, blockIdx.x, blockIdx.x * 64, redd, &flag, t1, t2)) return;
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x < 64 && col < C) {
    const int pb = col / mp.blk, pc = col - pb * mp.blk;
    sel4(mp.dbeta, pb)[pc] = (float)t1;
    sel4(mp.dgamma, pb)[pc] = (float)t2;
  }
}

// backward pass 1 for very narrow maps (C <= 8: the CEM's 3-channel output): one thread per row, all channels in
// registers -- the 64-channel-lane layout above would leave 61 of 64 lanes idle
__global__ __launch_bounds__(256) void bn_bwd_reduce_narrow_kernel(const float* __restrict__ y, int ldy,
                                                                   const float* __restrict__ dout, int ldd,
                                                                   const float* __restrict__ mi, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ partials,
                                                                   int64_t rows, int C, int act, int64_t rows_per_part) {
  __shared__ float red[2][8][256];
  const int t = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_part, r1 = min(r0 + rows_per_part, rows);
  float s1[8], s2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) s1[c] = s2[c] = 0.f;
  for (int64_t r = r0 + t; r < r1; r += 256) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c < C) {
        const float xh = (y[r * ldy + c] - mi[c]) * mi[C + c];
        const float dz = dout[r * ldd + c] * act_grad(xh * gamma[c] + beta[c], act);
        s1[c] += dz;
        s2[c] += dz * xh;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    red[0][c][t] = s1[c];
    red[1][c][t] = s2[c];
  }
  __syncthreads();
  if (t < 2 * C) {                      // fixed-order fold: deterministic
    const int sidx = t / C, c = t - sidx * C;
    float acc = 0.f;
    for (int i = 0; i < 256; ++i) acc += red[sidx][c][i];
    partials[((int64_t)blockIdx.y * 2 + sidx) * C + c] = acc;
  }
}

// stage 1 for very long part lists, in place: block y folds rows [y*chunk, (y+1)*chunk) into row y*chunk (read by nobody else)
__global__ void pair_fold_inplace_kernel(float* partials, int nparts, int C, int chunk) {
  __shared__ double red[2][16][16];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int p0 = blockIdx.y * chunk, p1 = min(p0 + chunk, nparts);
  double s1, s2;
  fold_parts(partials, p0, p1, C, c, cl, pl, red, s1, s2);  // ends with every read done (barrier inside)
  if (pl == 0 && c < C && p0 < nparts) {
    partials[((int64_t)p0 * 2 + 0) * C + c] = (float)s1;
    partials[((int64_t)p0 * 2 + 1) * C + c] = (float)s2;
  }
}

// partials -> dbeta (= sum dz), dgamma (= sum dz*xhat); pstride: row stride of the list (after an in-place stage 1)
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nparts, int C, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int pstride) {
  __shared__ double red[2][16][16];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s1, s2;
  fold_parts(partials, 0, nparts, C, c, cl, pl, red, s1, s2, pstride);
  if (pl == 0 && c < C) {
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
  }
}

template <typename T, int V, bool FIXED>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dout, int ldd,
                                    const T* __restrict__ dout1, int ldd1, const float* __restrict__ mi, mmi_bn_map mp,
                                    T* __restrict__ dy, int lddy, int64_t rows, int C, int act, int frozen) {
  const int cv = C / V;
  const float inv_rows = 1.0f / (float)rows;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (FIXED) {
    const int64_t r0 = e0 / cv, dr = stride / cv;
    const int c = (int)(e0 - r0 * cv) * V;
    const int pb = c / mp.blk, pc = c - pb * mp.blk;
    const Vec<V> m = ldv<V>(mi + c), is = ldv<V>(mi + C + c), g = ldv<V>(sel4(mp.gamma, pb) + pc), b = ldv<V>(sel4(mp.beta, pb) + pc),
                 dg = ldv<V>(sel4(mp.dgamma, pb) + pc), db = ldv<V>(sel4(mp.dbeta, pb) + pc);
    const ChanLoc loc = chan_loc(mp, c);
    const int ldd_ = loc.hi ? ldd1 : ldd;
    const T* py = y + r0 * ldy + c;
    const T* pd = (loc.hi ? dout1 : dout) + loc.off + r0 * ldd_;
    T* po = dy + r0 * lddy + c;
    const int64_t sy = dr * ldy, sd = dr * ldd_, so = dr * lddy;
#pragma unroll 2
    for (int64_t r = r0; r < rows; r += dr, py += sy, pd += sd, po += so) {
      const Vec<V> yy = ldv<V>(py), dd = ldv<V>(pd);
      Vec<V> o;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float xh = (yy.v[k] - m.v[k]) * is.v[k];
        const float dz = dd.v[k] * act_grad(xh * g.v[k] + b.v[k], act);
        const float t = frozen ? dz : dz - db.v[k] * inv_rows - xh * dg.v[k] * inv_rows;
        o.v[k] = g.v[k] * is.v[k] * t;
      }
      stv<V>(po, o);
    }
    return;
  }
  const int64_t total = rows * cv;
  for (int64_t e = e0; e < total; e += stride) {
    const int64_t r = e / cv;
    const int c = (int)(e - r * cv) * V;
    const int pb = c / mp.blk, pc = c - pb * mp.blk;
    const ChanLoc loc = chan_loc(mp, c);
    const Vec<V> yy = ldv<V>(y + r * ldy + c), dd = ldv<V>(loc.hi ? dout1 + r * ldd1 + loc.off : dout + r * ldd + loc.off), m = ldv<V>(mi + c),
                 is = ldv<V>(mi + C + c), g = ldv<V>(sel4(mp.gamma, pb) + pc), b = ldv<V>(sel4(mp.beta, pb) + pc),
                 dg = ldv<V>(sel4(mp.dgamma, pb) + pc), db = ldv<V>(sel4(mp.dbeta, pb) + pc);
    Vec<V> o;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float xh = (yy.v[k] - m.v[k]) * is.v[k];
      const float dz = dd.v[k] * act_grad(xh * g.v[k] + b.v[k], act);
      const float t = frozen ? dz : dz - db.v[k] * inv_rows - xh * dg.v[k] * inv_rows;
      o.v[k] = g.v[k] * is.v[k] * t;
    }
    stv<V>(dy + r * lddy + c, o);
  }
}

// column sums: block = (64 channels, rows part); 64 channel lanes x 4 row lanes
__global__ void colsum_partial_kernel(const float* __restrict__ x, int ldx, int64_t rows, int C,
                                      float* __restrict__ partials, int64_t rows_per_part) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_part, r1 = min(r0 + rows_per_part, rows);
  float s = 0.f;
  if (c < C)
    for (int64_t r = r0 + rl; r < r1; r += 4) s += x[r * ldx + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) partials[(int64_t)blockIdx.y * C + c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}
__global__ void colsum_finalize_kernel(const float* __restrict__ partials, int nparts, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int p = 0; p < nparts; ++p) s += (double)partials[(int64_t)p * C + c];
  out[c] = (float)s;
}

__global__ void i64_increment_kernel(int64_t* c) { *c += 1; }

inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b));
}
// grid for a streaming kernel over rows x cv column groups; *fixed = the grid stride is a multiple of cv
inline int ew_grid(int64_t rows, int cv, bool* fixed) {
  int b = ew_blocks(rows * cv);
  int g = cv, r = 256;  // gcd(cv, 256)
  while (r) {
    const int t = g % r;
    g = r;
    r = t;
  }
  const int m = cv / g;  // the grid stride 256*b is a multiple of cv iff b is a multiple of m (e.g. C=24: cv=6, m=3)
  if (m <= 64) b = cdiv(b, m) * m;
  *fixed = ((int64_t)b * 256) % cv == 0;
  return b;
}
inline bool vec_ok(int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs, uintptr_t mask = 15) {
  if (C % 4) return false;
  for (int l : lds)
    if (l % 4) return false;
  for (const void* p : ptrs)
    if (p && ((uintptr_t)p & mask)) return false;
  return true;
}

}  // namespace

extern "C" int mmi_bn_finalize(const float* partials, int nparts, int64_t rows, int C, float eps, float momentum,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked,
                               float* mean_invstd, void* stream) {
  MMI_CHECK_ARG(partials && mean_invstd && nparts > 0 && rows > 0 && C > 0, "mmi_bn_finalize: bad arguments");
  MMI_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "mmi_bn_finalize: running stats must come in pairs");
  hipStream_t s = (hipStream_t)stream;
  if (nparts > 1024) {  // long part list (P1/P2-sized maps): fold it 64-wide first, into the slack rows behind the list
    float* folded = const_cast<float*>(partials) + (int64_t)nparts * 2 * C;
    const int Y = MMI_BN_FOLD_ROWS, chunk = cdiv(nparts, Y);
    hipLaunchKernelGGL(stat_fold_kernel, dim3(cdiv(C, 16), Y), dim3(256), 0, s, partials, nparts, C, folded, chunk);
    MMI_CHECK_LAUNCH("mmi_bn_finalize(fold)");
    partials = folded;
    nparts = cdiv(nparts, chunk);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, partials, nparts, (double)rows, C, eps,
                     momentum, running_mean, running_var, num_batches_tracked, mean_invstd);
  MMI_CHECK_LAUNCH("mmi_bn_finalize");
  return MMI_OK;
}

extern "C" int mmi_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps,
                                 float* mean_invstd, void* stream) {
  MMI_CHECK_ARG(running_mean && running_var && mean_invstd && C > 0, "mmi_bn_eval_stats: bad arguments");
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, running_mean,
                     running_var, C, eps, mean_invstd);
  MMI_CHECK_LAUNCH("mmi_bn_eval_stats");
  return MMI_OK;
}

namespace {
// map of the entry points that take plain parameter vectors: one block, or two blocks split at `split` (gradients in two tensors)
mmi_bn_map plain_map(const float* gamma, const float* beta, float* dgamma, float* dbeta, float* dgamma1, float* dbeta1, int split, int C) {
  mmi_bn_map m{};
  m.period = C; m.split = split; m.ls0 = 0; m.ls1 = 0;
  m.blk = split < C ? split : C;
  m.nblk = cdiv(C, m.blk);
  for (int k = 0; k < 4 && k < m.nblk; ++k) {
    m.gamma[k] = gamma + (size_t)k * m.blk;
    m.beta[k] = beta + (size_t)k * m.blk;
    m.dgamma[k] = k == 0 ? dgamma : (dgamma1 ? dgamma1 + (size_t)(k - 1) * m.blk : nullptr);
    m.dbeta[k] = k == 0 ? dbeta : (dbeta1 ? dbeta1 + (size_t)(k - 1) * m.blk : nullptr);
  }
  return m;
}
int check_map(const mmi_bn_map& m, int C, bool backward, const char* who) {
  MMI_CHECK_ARG(m.nblk >= 1 && m.nblk <= 4 && m.blk > 0 && (int64_t)m.nblk * m.blk >= C, "%s: parameter blocks (%d x %d) do not cover %d channels", who, m.nblk, m.blk, C);
  for (int k = 0; k < m.nblk; ++k)
    MMI_CHECK_ARG(m.gamma[k] && m.beta[k] && (!backward || (m.dgamma[k] && m.dbeta[k])), "%s: null parameter pointer in block %d", who, k);
  MMI_CHECK_ARG(m.period > 0 && m.split > 0 && m.split <= m.period && C % m.period == 0, "%s: bad channel map (period %d, split %d, C %d)", who, m.period, m.split, C);
  return MMI_OK;
}
bool map_vec_ok(const mmi_bn_map& m, bool backward) {
  if (m.blk % 4 || m.period % 4 || m.split % 4 || m.ls0 % 4 || m.ls1 % 4) return false;
  for (int k = 0; k < m.nblk; ++k) {
    if (((uintptr_t)m.gamma[k] | (uintptr_t)m.beta[k]) & 15) return false;
    if (backward && (((uintptr_t)m.dgamma[k] | (uintptr_t)m.dbeta[k]) & 15)) return false;
  }
  return true;
}
bool map_two(const mmi_bn_map& m) { return m.split < m.period; }

template <typename T>
int bn_act_fwd_impl(const T* y, int ldy, const float* mean_invstd, const mmi_bn_map& mp, const T* residual, int ldr,
                    T* out, int ldo, T* out1, int ldo1, int64_t rows, int C, int act, void* stream) {
  MMI_CHECK_ARG(y && mean_invstd && out && rows > 0 && C > 0, "mmi_bn_act_fwd: bad arguments");
  if (int e = check_map(mp, C, false, "mmi_bn_act_fwd")) return e;
  const bool two = map_two(mp);
  MMI_CHECK_ARG(!two || out1 != nullptr, "mmi_bn_act_fwd: the channel map needs a second output");
  MMI_CHECK_ARG(ldy >= C && (!residual || ldr >= C), "mmi_bn_act_fwd: row stride < C");
  hipStream_t s = (hipStream_t)stream;
  const uintptr_t am = 4 * sizeof(T) - 1;
  const bool vec = vec_ok(C, {ldy, ldo, residual ? ldr : 0, two ? ldo1 : 0}, {y, out, residual, two ? out1 : nullptr}, am) &&
                   vec_ok(C, {}, {mean_invstd}) && map_vec_ok(mp, false);
  bool fixed;
  const int blocks = ew_grid(rows, vec ? C / 4 : C, &fixed);
#define LAUNCH_FWD(V_, F_) \
  hipLaunchKernelGGL((bn_act_fwd_kernel<T, V_, F_>), dim3(blocks), dim3(256), 0, s, y, ldy, mean_invstd, mp, residual, \
                     ldr, out, ldo, out1, ldo1, rows, C, act)
  if (vec && fixed) LAUNCH_FWD(4, true);
  else if (vec) LAUNCH_FWD(4, false);
  else if (fixed) LAUNCH_FWD(1, true);
  else LAUNCH_FWD(1, false);
#undef LAUNCH_FWD
  MMI_CHECK_LAUNCH("mmi_bn_act_fwd");
  return MMI_OK;
}
int check_split(int split, int C, const void* second, int ld1, const char* who) {
  MMI_CHECK_ARG(split > 0 && split <= C && (split == C || (second && split % 4 == 0 && ld1 >= C - split)), "%s: bad channel split", who);
  return MMI_OK;
}
}  // namespace

extern "C" int mmi_bn_act_fwd_split(const float* y, int ldy, const float* mean_invstd, const float* gamma, const float* beta,
                                    const float* residual, int ldr, float* out, int ldo, float* out1, int ldo1, int split,
                                    int64_t rows, int C, int act, void* stream) {
  MMI_CHECK_ARG(gamma && beta && ldo >= split, "mmi_bn_act_fwd: bad arguments");
  if (int e = check_split(split, C, out1, ldo1, "mmi_bn_act_fwd")) return e;
  mmi_bn_map m = plain_map(gamma, beta, nullptr, nullptr, nullptr, nullptr, C, C);
  m.split = split;
  return bn_act_fwd_impl<float>(y, ldy, mean_invstd, m, residual, ldr, out, ldo, out1, ldo1, rows, C, act, stream);
}

extern "C" int mmi_bn_act_fwd_split_bf16(const void* y, int ldy, const float* mean_invstd, const float* gamma, const float* beta,
                                         const void* residual, int ldr, void* out, int ldo, void* out1, int ldo1, int split,
                                         int64_t rows, int C, int act, void* stream) {
  MMI_CHECK_ARG(gamma && beta && ldo >= split, "mmi_bn_act_fwd_bf16: bad arguments");
  if (int e = check_split(split, C, out1, ldo1, "mmi_bn_act_fwd_bf16")) return e;
  mmi_bn_map m = plain_map(gamma, beta, nullptr, nullptr, nullptr, nullptr, C, C);
  m.split = split;
  return bn_act_fwd_impl<__bf16>((const __bf16*)y, ldy, mean_invstd, m, (const __bf16*)residual, ldr, (__bf16*)out, ldo,
                                 (__bf16*)out1, ldo1, rows, C, act, stream);
}

extern "C" int mmi_bn_act_fwd(const float* y, int ldy, const float* mean_invstd, const float* gamma, const float* beta,
                              const float* residual, int ldr, float* out, int ldo, int64_t rows, int C, int act,
                              void* stream) {
  return mmi_bn_act_fwd_split(y, ldy, mean_invstd, gamma, beta, residual, ldr, out, ldo, nullptr, 0, C, rows, C, act, stream);
}

// the channel-map form (see bn_act_fwd_kernel): up to four parameter blocks, output scattered per lane over out / out1
extern "C" int mmi_bn_act_fwd_map(const float* y, int ldy, const float* mean_invstd, const mmi_bn_map* map, const float* residual,
                                  int ldr, float* out, int ldo, float* out1, int ldo1, int64_t rows, int C, int act, void* stream) {
  MMI_CHECK_ARG(map != nullptr, "mmi_bn_act_fwd_map: null map");
  return bn_act_fwd_impl<float>(y, ldy, mean_invstd, *map, residual, ldr, out, ldo, out1, ldo1, rows, C, act, stream);
}

extern "C" int mmi_bn_bwd_parts(int64_t rows) {
  int64_t p = (rows + 127) / 128;
  return (int)(p > 1024 ? 1024 : (p < 1 ? 1 : p));
}

extern "C" int mmi_bn_act_bwd_reduce(const float* y, int ldy, const float* dout, int ldd, const float* mean_invstd,
                                     const float* gamma, const float* beta, float* partials, int64_t rows, int C,
                                     int act, void* stream) {
  MMI_CHECK_ARG(y && dout && mean_invstd && gamma && beta && partials && rows > 0 && C > 0, "mmi_bn_act_bwd_reduce: bad arguments");
  const int nparts = mmi_bn_bwd_parts(rows);
  const int64_t rpp = (rows + nparts - 1) / nparts;
  const dim3 grid(cdiv(C, 64), nparts);
  hipStream_t s = (hipStream_t)stream;
  const StatFold nofold{};
  const mmi_bn_map mp = plain_map(gamma, beta, nullptr, nullptr, nullptr, nullptr, C, C);
  if (C <= 8)
    hipLaunchKernelGGL(bn_bwd_reduce_narrow_kernel, dim3(1, nparts), dim3(256), 0, s, y, ldy, dout, ldd, mean_invstd, gamma,
                       beta, partials, rows, C, act, rpp);
  else if (vec_ok(C, {ldy, ldd}, {y, dout, mean_invstd, gamma, beta}))
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 4>), grid, dim3(256), 0, s, y, ldy, dout, ldd, (const float*)nullptr, 0, mean_invstd, mp,
                       partials, rows, C, act, rpp, nofold);
  else
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, 1>), grid, dim3(256), 0, s, y, ldy, dout, ldd, (const float*)nullptr, 0, mean_invstd, mp,
                       partials, rows, C, act, rpp, nofold);
  MMI_CHECK_LAUNCH("mmi_bn_act_bwd_reduce");
  return MMI_OK;
}

namespace {
template <typename T>
int launch_apply(const T* y, int ldy, const T* dout, int ldd, const T* dout1, int ldd1, const float* mean_invstd, const mmi_bn_map& mp,
                 T* dy, int lddy, int64_t rows, int C, int act, int frozen, hipStream_t s) {
  const bool two = map_two(mp);
  const bool vec = vec_ok(C, {ldy, ldd, lddy, two ? ldd1 : 0}, {y, dout, dy, two ? dout1 : nullptr}, 4 * sizeof(T) - 1) &&
                   vec_ok(C, {}, {mean_invstd}) && map_vec_ok(mp, true);
  bool fixed;
  const int blocks = ew_grid(rows, vec ? C / 4 : C, &fixed);
#define LAUNCH_APPLY(V_, F_) \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<T, V_, F_>), dim3(blocks), dim3(256), 0, s, y, ldy, dout, ldd, dout1, ldd1, mean_invstd, mp, \
                     dy, lddy, rows, C, act, frozen)
  if (vec && fixed) LAUNCH_APPLY(4, true);
  else if (vec) LAUNCH_APPLY(4, false);
  else if (fixed) LAUNCH_APPLY(1, true);
  else LAUNCH_APPLY(1, false);
#undef LAUNCH_APPLY
  MMI_CHECK_LAUNCH("mmi_bn_act_bwd_apply");
  return MMI_OK;
}
}  // namespace

extern "C" int mmi_bn_act_bwd_apply(const float* y, int ldy, const float* dout, int ldd, const float* mean_invstd,
                                    const float* gamma, const float* beta, const float* partials, int nparts, float* dy,
                                    int lddy, float* dgamma, float* dbeta, int64_t rows, int C, int act, int frozen,
                                    void* stream) {
  MMI_CHECK_ARG(y && dout && mean_invstd && gamma && beta && partials && dgamma && dbeta && rows > 0 && C > 0,
                "mmi_bn_act_bwd_apply: bad arguments");      // (dy == NULL: the fold only)
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, partials, nparts, C, dgamma, dbeta, 1);
  MMI_CHECK_LAUNCH("mmi_bn_act_bwd_apply(finalize)");
  if (dy == nullptr) return MMI_OK;
  const mmi_bn_map mp = plain_map(gamma, beta, dgamma, dbeta, nullptr, nullptr, C, C);
  return launch_apply<float>(y, ldy, dout, ldd, nullptr, 0, mean_invstd, mp, dy, lddy, rows, C, act, frozen, s);
}

// One-call BatchNorm(+activation) backward: reduce (whose last-arriving workgroups write dgamma / dbeta) + apply = two
// launches instead of three.  Workspace layout: arrival counters | partials[nparts][2][C] | l1[ngroups][2][C].
extern "C" size_t mmi_bn_act_bwd_workspace(int64_t rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  const int nparts = mmi_bn_bwd_parts(rows);
  const int ngroups = cdiv(nparts, stat_group_size(nparts));
  return (size_t)MMI_STAT_MAX_COUNTERS * sizeof(int) + ((size_t)nparts + ngroups) * 2 * C * sizeof(float);
}

namespace {
template <typename T>
int bn_act_bwd_impl(const T* y, int ldy, const T* dout, int ldd, const T* dout1, int ldd1, const float* mean_invstd,
                    const mmi_bn_map& mp, void* workspace, size_t workspace_bytes, T* dy, int lddy, int64_t rows, int C, int act,
                    int frozen, void* stream) {
  MMI_CHECK_ARG(y && dout && mean_invstd && workspace && rows > 0 && C > 0, "mmi_bn_act_bwd: bad arguments");   // (dy == NULL: sums only)
  if (int e = check_map(mp, C, true, "mmi_bn_act_bwd")) return e;
  const bool two = map_two(mp);
  MMI_CHECK_ARG(!two || (dout1 != nullptr && C > 8), "mmi_bn_act_bwd: the channel map needs a second gradient tensor");
  MMI_CHECK_ARG(workspace_bytes >= mmi_bn_act_bwd_workspace(rows, C) && ((uintptr_t)workspace & 15) == 0, "mmi_bn_act_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int nparts = mmi_bn_bwd_parts(rows);
  const int64_t rpp = (rows + nparts - 1) / nparts;
  float* partials = (float*)((char*)workspace + (size_t)MMI_STAT_MAX_COUNTERS * sizeof(int));
  const int nct = cdiv(C, 64), G = stat_group_size(nparts), ngroups = cdiv(nparts, G);
  static const bool fold_off = getenv("MMIDET_BN_FOLD") != nullptr && atoi(getenv("MMIDET_BN_FOLD")) == 0;  // (A/B switch)
  const bool is_f32 = sizeof(T) == 4;
  const bool plain = mp.nblk == 1 && !two && mp.period == C;
  if (is_f32 && plain && (C <= 8 || fold_off || ngroups * nct + nct > MMI_STAT_MAX_COUNTERS)) {  // the CEM's 3-channel map: one thread per row, separate fold
    if (int e = mmi_bn_act_bwd_reduce((const float*)y, ldy, (const float*)dout, ldd, mean_invstd, mp.gamma[0], mp.beta[0], partials, rows, C, act, stream)) return e;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, (const float*)partials, nparts, C, mp.dgamma[0], mp.dbeta[0], 1);
    MMI_CHECK_LAUNCH("mmi_bn_act_bwd(finalize)");
  } else {
    MMI_CHECK_ARG(ngroups * nct + nct <= MMI_STAT_MAX_COUNTERS, "mmi_bn_act_bwd: too many column tiles");
    StatFold f{partials, partials + (size_t)nparts * 2 * C, (int*)workspace, nparts, C, nct, G};
    const dim3 grid(nct, nparts);
    if (vec_ok(C, {ldy, ldd, two ? ldd1 : 0}, {y, dout, two ? dout1 : nullptr}, 4 * sizeof(T) - 1) && vec_ok(C, {}, {mean_invstd}) && map_vec_ok(mp, false))
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 4>), grid, dim3(256), 0, s, y, ldy, dout, ldd, dout1, ldd1, mean_invstd, mp,
                         partials, rows, C, act, rpp, f);
    else
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 1>), grid, dim3(256), 0, s, y, ldy, dout, ldd, dout1, ldd1, mean_invstd, mp,
                         partials, rows, C, act, rpp, f);
    MMI_CHECK_LAUNCH("mmi_bn_act_bwd(reduce)");
  }
  if (dy == nullptr) return MMI_OK;      // the caller applies on its own way in (mmi_cem_conv2_wgrad_bn)
  return launch_apply<T>(y, ldy, dout, ldd, dout1, ldd1, mean_invstd, mp, dy, lddy, rows, C, act, frozen, s);
}
}  // namespace

extern "C" int mmi_bn_act_bwd(const float* y, int ldy, const float* dout, int ldd, const float* dout1, int ldd1, int split,
                              const float* mean_invstd, const float* gamma, const float* beta, void* workspace,
                              size_t workspace_bytes, float* dy, int lddy, float* dgamma, float* dbeta, float* dgamma1,
                              float* dbeta1, int64_t rows, int C, int act, int frozen, void* stream) {
  MMI_CHECK_ARG(gamma && beta && dgamma && dbeta, "mmi_bn_act_bwd: bad arguments");
  if (int e = check_split(split, C, dout1, ldd1, "mmi_bn_act_bwd")) return e;
  MMI_CHECK_ARG(split == C || (dgamma1 && dbeta1), "mmi_bn_act_bwd: bad channel split");
  return bn_act_bwd_impl<float>(y, ldy, dout, ldd, dout1, ldd1, mean_invstd, plain_map(gamma, beta, dgamma, dbeta, dgamma1, dbeta1, split, C),
                                workspace, workspace_bytes, dy, lddy, rows, C, act, frozen, stream);
}

extern "C" int mmi_bn_act_bwd_bf16(const void* y, int ldy, const void* dout, int ldd, const void* dout1, int ldd1, int split,
                                   const float* mean_invstd, const float* gamma, const float* beta, void* workspace,
                                   size_t workspace_bytes, void* dy, int lddy, float* dgamma, float* dbeta, float* dgamma1,
                                   float* dbeta1, int64_t rows, int C, int act, int frozen, void* stream) {
  MMI_CHECK_ARG(gamma && beta && dgamma && dbeta, "mmi_bn_act_bwd_bf16: bad arguments");
  if (int e = check_split(split, C, dout1, ldd1, "mmi_bn_act_bwd_bf16")) return e;
  MMI_CHECK_ARG(split == C || (dgamma1 && dbeta1), "mmi_bn_act_bwd_bf16: bad channel split");
  return bn_act_bwd_impl<__bf16>((const __bf16*)y, ldy, (const __bf16*)dout, ldd, (const __bf16*)dout1, ldd1, mean_invstd,
                                 plain_map(gamma, beta, dgamma, dbeta, dgamma1, dbeta1, split, C), workspace, workspace_bytes,
                                 (__bf16*)dy, lddy, rows, C, act, frozen, stream);
}

// the channel-map form: gradients of up to four parameter blocks, incoming gradient gathered per lane from dout / dout1
extern "C" int mmi_bn_act_bwd_map(const float* y, int ldy, const float* dout, int ldd, const float* dout1, int ldd1,
                                  const float* mean_invstd, const mmi_bn_map* map, void* workspace, size_t workspace_bytes, float* dy,
                                  int lddy, int64_t rows, int C, int act, int frozen, void* stream) {
  MMI_CHECK_ARG(map != nullptr, "mmi_bn_act_bwd_map: null map");
  return bn_act_bwd_impl<float>(y, ldy, dout, ldd, dout1, ldd1, mean_invstd, *map, workspace, workspace_bytes, dy, lddy, rows, C, act,
                                frozen, stream);
}

extern "C" int mmi_bn_act_bwd_apply_map(const float* y, int ldy, const float* dout, int ldd, const float* dout1, int ldd1,
                                        const float* mean_invstd, const mmi_bn_map* map, const float* const* partials, int nparts,
                                        float* dy, int lddy, int64_t rows, int C, int act, int frozen, void* stream) {
  MMI_CHECK_ARG(y && dout && mean_invstd && map && partials && dy && rows > 0 && C > 0 && nparts > 0, "mmi_bn_act_bwd_apply_map: bad arguments");
  if (int e = check_map(*map, C, true, "mmi_bn_act_bwd_apply_map")) return e;
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < map->nblk; ++i) {
    MMI_CHECK_ARG(partials[i] != nullptr, "mmi_bn_act_bwd_apply_map: null partial list (block %d)", i);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(map->blk, 16)), dim3(256), 0, s, partials[i], nparts, map->blk, map->dgamma[i],
                       map->dbeta[i], 1);
    MMI_CHECK_LAUNCH("mmi_bn_act_bwd_apply_map(finalize)");
  }
  return launch_apply<float>(y, ldy, dout, ldd, dout1, ldd1, mean_invstd, *map, dy, lddy, rows, C, act, frozen, s);
}

extern "C" int mmi_colsum(const float* x, int ldx, int64_t rows, int C, float* partials, float* out, void* stream) {
  MMI_CHECK_ARG(x && partials && out && rows > 0 && C > 0 && ldx >= C, "mmi_colsum: bad arguments");
  const int nparts = mmi_bn_bwd_parts(rows);
  const int64_t rpp = (rows + nparts - 1) / nparts;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(C, 64), nparts), dim3(256), 0, s, x, ldx, rows, C, partials, rpp);
  MMI_CHECK_LAUNCH("mmi_colsum");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, partials, nparts, C, out);
  MMI_CHECK_LAUNCH("mmi_colsum(finalize)");
  return MMI_OK;
}

int mmi_i64_increment(int64_t* counter, void* stream) {
  hipLaunchKernelGGL(i64_increment_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
  MMI_CHECK_LAUNCH("mmi_i64_increment");
  return MMI_OK;
}

// library-internal (common.h): partials[part][2][C] -> out0 = sum of slot 0, out1 = sum of slot 1, folded in fp64.  Lists
// longer than 1024 parts (the CEM stencil backward: one part per 256 full-resolution pixels) are first folded 128-wide IN
// PLACE, so the caller's partials are consumed.
int mmi_pair_colsum(float* partials, int nparts, int C, float* out0, float* out1, void* stream) {
  MMI_CHECK_ARG(partials && out0 && out1 && nparts > 0 && C > 0, "mmi_pair_colsum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  int pstride = 1;
  if (nparts > 1024) {
    const int chunk = cdiv(nparts, 128);
    hipLaunchKernelGGL(pair_fold_inplace_kernel, dim3(cdiv(C, 16), cdiv(nparts, chunk)), dim3(256), 0, s, partials, nparts, C, chunk);
    MMI_CHECK_LAUNCH("mmi_pair_colsum(fold)");
    pstride = chunk;
    nparts = cdiv(nparts, chunk);
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, (const float*)partials, nparts, C, out1, out0, pstride);
  MMI_CHECK_LAUNCH("mmi_pair_colsum");
  return MMI_OK;
}
