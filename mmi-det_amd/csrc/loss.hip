// Detection loss, forward value AND gradient w.r.t. the head outputs in one pass (the loss has no learnable state, so
// d loss / d pred only needs the upstream scalar, applied later with mmi_scale).
//
// Replaces ComputeLoss.__call__ (utils/loss.py:113-184) + bbox_iou(CIoU) (utils/general.py:403-447) of the reference:
// per matched record gather -> sigmoid decode -> CIoU -> cls BCE; objectness BCE over every grid cell with the sparse
// IoU targets scattered "last record wins" (index_put_ order on the reference's CPU path); gains, balance, batch-size
// scaling and the 0.1 * mean(CombineLoss) term.  Record counts are read from device memory (no host sync).
#include "common.h"

namespace {

constexpr int MAXL = 5;

struct LossP {
  const float* p[MAXL];    // head outputs (B,na,ny,nx,no)
  float* dp[MAXL];         // gradients, same shape
  int* owner[MAXL];        // per-cell winning record (-1 = none)
  int64_t cells[MAXL];     // B*na*ny*nx
  int ny[MAXL], nx[MAXL];
  float balance[MAXL];
  const int64_t* idx;      // (nl,4,cap)
  const int64_t* tcls;     // (nl,cap)
  const float* tbox;       // (nl,cap,4)
  const float* anch;       // (nl,cap,2)
  const int* counts;       // (nl)
  float* iou;              // (nl,cap) scratch
  double* acc;             // (nl, ACC_STRIDE) per-workgroup partial sums: [0,64) sum(1-iou), [64,128) sum cls bce, [128,..) sum obj bce
  int* cnt[MAXL];          // per-cell number of records
  int* cellid;             // (nl,cap) cell of every record
  float* grec;             // (nl,cap,no) gradient contribution of every record (slot 4 unused)
  int nl, na, no, nc, cap, bs;
  float hbox, hobj, hcls, gr, cp, cn;
  // nhwc = 1: p / dp are the head convolutions' own NHWC outputs (B,ny,nx,na*no) -- Detect's (B,na,ny,nx,no) tensor as a strided
  // view of them (models/yolo_test.py:54-55 of the reference without the permute copy); cells are still numbered anchor-major
  int nhwc;
};
// element offset of the outputs of the cell with anchor-major number `cell` = ((b*na + a)*ny + gj)*nx + gi
__device__ __forceinline__ int64_t pred_off(const LossP& P, int l, int64_t cell) {
  if (!P.nhwc) return cell * P.no;
  const int nx = P.nx[l], ny = P.ny[l];
  const int gi = (int)(cell % nx);
  int64_t t = cell / nx;
  const int gj = (int)(t % ny);
  t /= ny;
  const int a = (int)(t % P.na);
  const int64_t b = t / P.na;
  return (((b * ny + gj) * nx + gi) * P.na + a) * P.no;
}

// forward-mode dual number over the 4 predicted box parameters (x, y, w, h)
struct D4 {
  float v, d[4];
};
__device__ __forceinline__ D4 cst(float v) { return D4{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 var(float v, int i) {
  D4 r = cst(v);
  r.d[i] = 1.f;
  return r;
}
__device__ __forceinline__ D4 operator+(D4 a, D4 b) {
  D4 r; r.v = a.v + b.v;
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ __forceinline__ D4 operator-(D4 a, D4 b) {
  D4 r; r.v = a.v - b.v;
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ __forceinline__ D4 operator*(D4 a, D4 b) {
  D4 r; r.v = a.v * b.v;
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
  D4 r; r.v = a.v / b.v;
  const float inv = 1.0f / b.v;
  for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
__device__ __forceinline__ D4 dmin(D4 a, D4 b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ D4 dmax(D4 a, D4 b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ D4 clamp0(D4 a) { return a.v >= 0.f ? a : cst(0.f); }
__device__ __forceinline__ D4 datan(D4 a) {
  D4 r; r.v = atanf(a.v);
  const float g = 1.0f / (1.0f + a.v * a.v);
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * g;
  return r;
}
__device__ __forceinline__ D4 half_of(D4 a) {
  D4 r; r.v = a.v / 2.f;
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] / 2.f;
  return r;
}

// CIoU of box1 = predicted (x,y,w,h) [differentiated] vs box2 = target (x,y,w,h); general.py:403-447, x1y1x2y2=False
__device__ D4 ciou(float px, float py, float pw, float ph, const float* t) {
  const float eps = 1e-7f;
  const D4 X = var(px, 0), Y = var(py, 1), Wd = var(pw, 2), Hd = var(ph, 3);
  const D4 b1x1 = X - half_of(Wd), b1x2 = X + half_of(Wd), b1y1 = Y - half_of(Hd), b1y2 = Y + half_of(Hd);
  const D4 b2x1 = cst(t[0] - t[2] / 2.f), b2x2 = cst(t[0] + t[2] / 2.f), b2y1 = cst(t[1] - t[3] / 2.f),
           b2y2 = cst(t[1] + t[3] / 2.f);
  const D4 inter = clamp0(dmin(b1x2, b2x2) - dmax(b1x1, b2x1)) * clamp0(dmin(b1y2, b2y2) - dmax(b1y1, b2y1));
  const D4 w1 = b1x2 - b1x1, h1 = b1y2 - b1y1 + cst(eps);
  const D4 w2 = b2x2 - b2x1, h2 = b2y2 - b2y1 + cst(eps);
  const D4 uni = w1 * h1 + w2 * h2 - inter + cst(eps);
  const D4 iou = inter / uni;
  const D4 cw = dmax(b1x2, b2x2) - dmin(b1x1, b2x1), ch = dmax(b1y2, b2y2) - dmin(b1y1, b2y1);
  const D4 c2 = cw * cw + ch * ch + cst(eps);
  const D4 dx = b2x1 + b2x2 - b1x1 - b1x2, dy = b2y1 + b2y2 - b1y1 - b1y2;
  const D4 rho2 = (dx * dx + dy * dy) / cst(4.f);
  const D4 da = datan(w2 / h2) - datan(w1 / h1);
  const D4 v = cst(0.40528473456935109f) * (da * da);  // 4 / pi^2
  const float alpha = v.v / (v.v - iou.v + (1.f + eps));  // torch.no_grad()
  return iou - (rho2 / c2 + v * cst(alpha));
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float bce_logits(float x, float t) {  // nn.BCEWithLogitsLoss element (pos_weight = 1)
  return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

constexpr int REC_BLOCKS = 64, OBJ_BLOCKS = 1024, ACC_STRIDE = 2 * REC_BLOCKS + OBJ_BLOCKS;

// workgroup sum -> its own slot (no atomics: loss_finalize_kernel adds the slots in a fixed order, so the loss value is
// run-to-run bit-identical)
__device__ __forceinline__ void block_sum_to(double v, double* dst, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
    *dst = s;
  }
}

// grid (record blocks, nl)
__global__ __launch_bounds__(256) void loss_records_kernel(LossP P) {
  __shared__ double sh[4];
  const int l = blockIdx.y;
  const int n = P.counts[l];
  const int no = P.no;
  double s_box = 0.0, s_cls = 0.0;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
    const int64_t* ix = P.idx + (int64_t)l * 4 * P.cap;
    const int b = (int)ix[r], a = (int)ix[P.cap + r], gj = (int)ix[2 * (int64_t)P.cap + r], gi = (int)ix[3 * (int64_t)P.cap + r];
    const int64_t cell = (((int64_t)b * P.na + a) * P.ny[l] + gj) * P.nx[l] + gi;
    const float* ps = P.p[l] + pred_off(P, l, cell);
    const float* an = P.anch + ((int64_t)l * P.cap + r) * 2;
    const float s0 = sigm(ps[0]), s1 = sigm(ps[1]), s2 = sigm(ps[2]), s3 = sigm(ps[3]);
    const float px = s0 * 2.f - 0.5f, py = s1 * 2.f - 0.5f;                 // loss.py:128
    const float pw = (s2 * 2.f) * (s2 * 2.f) * an[0], ph = (s3 * 2.f) * (s3 * 2.f) * an[1];  // loss.py:129
    const D4 c = ciou(px, py, pw, ph, P.tbox + ((int64_t)l * P.cap + r) * 4);
    P.iou[(int64_t)l * P.cap + r] = c.v;
    s_box += (double)(1.0f - c.v);                                           // loss.py:132
    atomicMax(P.owner[l] + cell, r);                                         // last record wins (loss.py:135)
    atomicAdd(P.cnt[l] + cell, 1);                                           // (integer: order-independent)
    P.cellid[(int64_t)l * P.cap + r] = (int)cell;
    // Several records may land in one cell (two targets whose boxes fall into the same grid cell under the same anchor):
    // their gradients are summed by loss_scatter_kernel in record order, not with float atomics.
    float* grow = P.grec + ((int64_t)l * P.cap + r) * no;
    const float gb = -(float)P.bs * P.hbox / (float)n;                       // d total / d iou
    grow[0] = gb * c.d[0] * 2.f * s0 * (1.f - s0);
    grow[1] = gb * c.d[1] * 2.f * s1 * (1.f - s1);
    grow[2] = gb * c.d[2] * 8.f * s2 * s2 * (1.f - s2) * an[0];
    grow[3] = gb * c.d[3] * 8.f * s3 * s3 * (1.f - s3) * an[1];
    if (P.nc > 1) {                                                          // loss.py:138-141
      const int tc = (int)P.tcls[(int64_t)l * P.cap + r];
      const float gc = (float)P.bs * P.hcls / ((float)n * (float)P.nc);
      for (int k = 0; k < P.nc; ++k) {
        const float x = ps[5 + k], t = (k == tc) ? P.cp : P.cn;
        s_cls += (double)bce_logits(x, t);
        grow[5 + k] = gc * (sigm(x) - t);
      }
    } else {
      for (int k = 0; k < P.nc; ++k) grow[5 + k] = 0.f;
    }
  }
  block_sum_to(s_box, P.acc + (int64_t)l * ACC_STRIDE + blockIdx.x, sh);
  block_sum_to(s_cls, P.acc + (int64_t)l * ACC_STRIDE + REC_BLOCKS + blockIdx.x, sh);
}

// grid (record blocks, nl): the LAST record of a cell (its owner) writes the cell's box/class gradient = the sum over the
// cell's records in increasing record order (one term for nearly every cell)
__global__ __launch_bounds__(256) void loss_scatter_kernel(LossP P) {
  const int l = blockIdx.y;
  const int n = P.counts[l];
  const int no = P.no;
  const int* cid = P.cellid + (int64_t)l * P.cap;
  const float* g = P.grec + (int64_t)l * P.cap * no;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
    const int cell = cid[r];
    if (P.owner[l][cell] != r) continue;
    float* dps = P.dp[l] + pred_off(P, l, cell);
    if (P.cnt[l][cell] == 1) {
      for (int k = 0; k < no; ++k)
        if (k != 4) dps[k] = g[(int64_t)r * no + k];
      continue;
    }
    // a shared cell (rare): ONE scan over the earlier records -- the cell's gradient row is the accumulator, every column sums
    // its terms in increasing record order starting from 0 (the same sequence of additions as a per-column scan, which cost
    // `no` scans of up to n records per shared cell: 0.5 ms per step)
    for (int k = 0; k < no; ++k)
      if (k != 4) dps[k] = 0.f;
    int left = P.cnt[l][cell];
    for (int q = 0; q <= r && left > 0; ++q)
      if (cid[q] == cell) {
        --left;
        for (int k = 0; k < no; ++k)
          if (k != 4) dps[k] += g[(int64_t)q * no + k];
      }
  }
}

// grid (cell blocks, nl): objectness BCE over every cell, writes dp[...,4]
__global__ __launch_bounds__(256) void loss_obj_kernel(LossP P) {
  __shared__ double sh[4];
  const int l = blockIdx.y;
  const int no = P.no;
  const int64_t ncell = P.cells[l];
  const float g = (float)P.bs * P.hobj * P.balance[l] / (float)ncell;
  double s = 0.0;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < ncell; c += (int64_t)gridDim.x * 256) {
    const int64_t po = pred_off(P, l, c);
    const float x = P.p[l][po + 4];
    const int o = P.owner[l][c];
    float t = 0.f;
    if (o >= 0) t = (1.0f - P.gr) + P.gr * fmaxf(P.iou[(int64_t)l * P.cap + o], 0.f);   // loss.py:135
    s += (double)bce_logits(x, t);
    P.dp[l][po + 4] = g * (sigm(x) - t);
  }
  block_sum_to(s, P.acc + (int64_t)l * ACC_STRIDE + 2 * REC_BLOCKS + blockIdx.x, sh);
}

// out[0] = loss (scaled by bs), out[1..4] = lbox, lobj, lcls, Detectloss (loss.py:154-184)
__global__ void loss_finalize_kernel(LossP P, const float* __restrict__ combine, int ncombine, float alpha, int flag,
                                     float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float lbox = 0.f, lobj = 0.f, lcls = 0.f;
  for (int l = 0; l < P.nl; ++l) {
    const int n = P.counts[l];
    const double* a = P.acc + (int64_t)l * ACC_STRIDE;
    double sb = 0.0, sc = 0.0, so = 0.0;          // fixed-order sums of the per-workgroup slots (unused slots hold zero)
    for (int i = 0; i < REC_BLOCKS; ++i) sb += a[i], sc += a[REC_BLOCKS + i];
    for (int i = 0; i < OBJ_BLOCKS; ++i) so += a[2 * REC_BLOCKS + i];
    if (n > 0) {
      lbox += (float)(sb / (double)n);
      if (P.nc > 1) lcls += (float)(sc / ((double)n * P.nc));
    }
    lobj += (float)(so / (double)P.cells[l]) * P.balance[l];
  }
  lbox *= P.hbox;
  lobj *= P.hobj;
  lcls *= P.hcls;
  const float det = lbox + lobj + lcls;
  float loss = det;
  if (flag) {
    float avg = 0.f;
    if (ncombine > 0) {
      float s = 0.f;
      for (int i = 0; i < ncombine; ++i) s += combine[i];
      avg = s / (float)ncombine * alpha;
    }
    loss = avg + det;
  }
  out[0] = loss * (float)P.bs;
  out[1] = lbox;
  out[2] = lobj;
  out[3] = lcls;
  out[4] = det;
}

}  // namespace

// acc (nl*ACC_STRIDE doubles) | iou (nl*cap floats) | cellid (nl*cap ints) | grec (nl*cap*MAX_NO floats) | owner, cnt
// (total_cells ints each).  The entry point does not know nc here, so grec is sized for MAX_NO outputs per anchor.
constexpr int MAX_NO = 96;
extern "C" size_t mmi_detect_loss_workspace(int nl, int64_t total_cells, int64_t cap) {
  const size_t c = (size_t)(cap > 0 ? cap : 1);
  return ((size_t)nl * ACC_STRIDE * sizeof(double) + (size_t)nl * c * (sizeof(float) + sizeof(int) + MAX_NO * sizeof(float)) +
          2 * (size_t)total_cells * sizeof(int) + 15) & ~(size_t)15;
}

extern "C" int mmi_detect_loss(const float* const* preds, float* const* dpreds, const int32_t* grids_host, int nl, int bs,
                               int na, int nc, const int64_t* idx, const int64_t* tcls, const float* tbox, const float* anch,
                               const int32_t* counts, int64_t cap, const float* balance_host, float hbox, float hobj,
                               float hcls, float gr, float cp, float cn, const float* combine, int ncombine, float alpha,
                               int flag, void* workspace, size_t workspace_bytes, float* out5, void* stream) {
  MMI_CHECK_ARG(preds && dpreds && grids_host && balance_host && counts && workspace && out5, "mmi_detect_loss: null pointer");
  MMI_CHECK_ARG(nl > 0 && nl <= MAXL && bs > 0 && na > 0 && nc > 0 && nc + 5 <= MAX_NO && cap >= 0 && cap < (1LL << 30), "mmi_detect_loss: bad sizes");
  MMI_CHECK_ARG(cap == 0 || (idx && tcls && tbox && anch), "mmi_detect_loss: null target arrays");
  LossP P{};
  int64_t total = 0;
  for (int l = 0; l < nl; ++l) {
    P.ny[l] = grids_host[2 * l];
    P.nx[l] = grids_host[2 * l + 1];
    P.cells[l] = (int64_t)bs * na * P.ny[l] * P.nx[l];
    total += P.cells[l];
  }
  MMI_CHECK_ARG(workspace_bytes >= mmi_detect_loss_workspace(nl, total, cap), "mmi_detect_loss: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)workspace;
  const size_t capn = (size_t)(cap > 0 ? cap : 1);
  P.acc = (double*)w;
  w += (size_t)nl * ACC_STRIDE * sizeof(double);
  P.iou = (float*)w;
  w += (size_t)nl * capn * sizeof(float);
  P.cellid = (int*)w;
  w += (size_t)nl * capn * sizeof(int);
  P.grec = (float*)w;
  w += (size_t)nl * capn * MAX_NO * sizeof(float);
  int* owner = (int*)w;
  int* cnt = owner + total;
  // (fill KERNELS, not hipMemsetAsync: common.h::mmi_fill_bytes)
  if (int e = mmi_fill_bytes(P.acc, 0, (size_t)nl * ACC_STRIDE * sizeof(double), s)) return e;
  if (int e = mmi_fill_bytes(owner, 0xFF, (size_t)total * sizeof(int), s)) return e;
  if (int e = mmi_fill_bytes(cnt, 0, (size_t)total * sizeof(int), s)) return e;
  int64_t off = 0, maxcells = 0;
  for (int l = 0; l < nl; ++l) {
    P.p[l] = preds[l];
    P.dp[l] = dpreds[l];
    P.owner[l] = owner + off;
    P.cnt[l] = cnt + off;
    off += P.cells[l];
    P.balance[l] = balance_host[l];
    if (P.cells[l] > maxcells) maxcells = P.cells[l];
    if (int e = mmi_fill_bytes(dpreds[l], 0, (size_t)P.cells[l] * (nc + 5) * sizeof(float), s)) return e;
  }
  P.idx = idx; P.tcls = tcls; P.tbox = tbox; P.anch = anch; P.counts = counts;
  P.nl = nl; P.na = na; P.no = nc + 5; P.nc = nc; P.cap = (int)cap; P.bs = bs;
  P.hbox = hbox; P.hobj = hobj; P.hcls = hcls; P.gr = gr; P.cp = cp; P.cn = cn;
  P.nhwc = (flag & 2) ? 1 : 0;
  flag &= 1;
  if (cap > 0) {
    const dim3 rgrid(cdiv(cap, 256) > REC_BLOCKS ? REC_BLOCKS : cdiv(cap, 256), nl);
    hipLaunchKernelGGL(loss_records_kernel, rgrid, dim3(256), 0, s, P);
    MMI_CHECK_LAUNCH("mmi_detect_loss(records)");
    hipLaunchKernelGGL(loss_scatter_kernel, rgrid, dim3(256), 0, s, P);
    MMI_CHECK_LAUNCH("mmi_detect_loss(scatter)");
  }
  int ob = cdiv(maxcells, 256);
  if (ob > OBJ_BLOCKS) ob = OBJ_BLOCKS;
  hipLaunchKernelGGL(loss_obj_kernel, dim3(ob, nl), dim3(256), 0, s, P);
  MMI_CHECK_LAUNCH("mmi_detect_loss(obj)");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, P, combine, ncombine, alpha, flag, out5);
  MMI_CHECK_LAUNCH("mmi_detect_loss(finalize)");
  return MMI_OK;
}
