// Detection-loss target assignment as ONE integer kernel (one workgroup per detection level), bit-exact against
// ComputeLoss.build_targets of the reference (utils/loss.py:189-245).
//
// The reference builds the candidate list with ~60 tiny tensor ops per level and a host sync at every boolean mask.
// Here every candidate (offset o, anchor a, target t) evaluates the same IEEE fp32 expressions (this file is compiled
// with -ffp-contract=off; hipcc's fp32 divide is correctly rounded by default) and a block-wide stable prefix sum
// reproduces torch's boolean-mask order: offset-major, then anchor-major, then target order.  Outputs are
// fixed-capacity with a device-side count, so no host sync is needed to run the loss.
#include "common.h"

namespace {

constexpr int TB = 1024;

__device__ __forceinline__ float frac1(float x) {  // torch `x % 1.` for x >= 0 (fmod is exact)
  return fmodf(x, 1.0f);
}

__global__ __launch_bounds__(TB) void build_targets_kernel(const float* __restrict__ targets, int nt,
                                                           const float* __restrict__ anchors, int na,
                                                           const int* __restrict__ grids, float anchor_t,
                                                           int64_t* __restrict__ idx, int64_t* __restrict__ tcls,
                                                           float* __restrict__ tbox, float* __restrict__ anch,
                                                           int* __restrict__ counts) {
  __shared__ int wsum[TB / 64];
  __shared__ int base_s;
  const int l = blockIdx.x;
  const int ny = grids[2 * l], nx = grids[2 * l + 1];
  const float fnx = (float)nx, fny = (float)ny;
  const int64_t cap = (int64_t)5 * na * nt;
  int64_t* idx_l = idx + (int64_t)l * 4 * cap;
  int64_t* tcls_l = tcls + (int64_t)l * cap;
  float* tbox_l = tbox + (int64_t)l * cap * 4;
  float* anch_l = anch + (int64_t)l * cap * 2;
  const float* anc_l = anchors + (int64_t)l * na * 2;
  const int per_off = na * nt;
  const int total = 5 * per_off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();

  for (int c0 = 0; c0 < total; c0 += TB) {
    const int c = c0 + threadIdx.x;
    bool keep = false;
    int a = 0, ti = 0, o = 0;
    float tx = 0, ty = 0, tw = 0, th = 0, aw = 0, ah = 0, fimg = 0, fcls = 0;
    if (c < total) {
      o = c / per_off;
      const int r = c - o * per_off;
      a = r / nt;
      ti = r - a * nt;
      const float* t = targets + (int64_t)ti * 6;
      fimg = t[0];
      fcls = t[1];
      tx = t[2] * fnx;  // targets * gain, gain = (1,1,nx,ny,nx,ny,1)   loss.py:206-209
      ty = t[3] * fny;
      tw = t[4] * fnx;
      th = t[5] * fny;
      aw = anc_l[2 * a];
      ah = anc_l[2 * a + 1];
      const float rw = tw / aw, rh = th / ah;                                     // :212
      const float m = fmaxf(fmaxf(rw, 1.0f / rw), fmaxf(rh, 1.0f / rh));          // :213
      keep = (m < anchor_t) && (fimg >= 0.f);  // rows with a negative image index are padding (fixed-shape graph mode)
      if (keep && o > 0) {
        const float gx = (o == 1 || o == 2) ? tx : fnx - tx;  // j,k test gxy; l,m test gxi = gain - gxy   :219-221
        const float gy = (o == 1 || o == 2) ? ty : fny - ty;
        const float v = (o == 1 || o == 3) ? gx : gy;
        keep = (frac1(v) < 0.5f) && (v > 1.0f);
      }
    }
    // stable block-wide exclusive scan of the keep flags
    const unsigned long long mask = __ballot(keep);
    const int wpre = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(mask);
    __syncthreads();
    int pre = base_s;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    int tot = 0;
    for (int w = 0; w < TB / 64; ++w) tot += wsum[w];
    if (keep) {
      const int64_t pos = pre + wpre;
      const float offx = (o == 1) ? 0.5f : (o == 3 ? -0.5f : 0.0f);               // off * g   :198-201
      const float offy = (o == 2) ? 0.5f : (o == 4 ? -0.5f : 0.0f);
      long long gi = (long long)(tx - offx);                                      // .long() truncates   :233
      long long gj = (long long)(ty - offy);
      gi = gi < 0 ? 0 : (gi > nx - 1 ? nx - 1 : gi);                               // clamp_ (in place on gij) :239
      gj = gj < 0 ? 0 : (gj > ny - 1 ? ny - 1 : gj);
      idx_l[0 * cap + pos] = (long long)fimg;
      idx_l[1 * cap + pos] = a;
      idx_l[2 * cap + pos] = gj;
      idx_l[3 * cap + pos] = gi;
      tcls_l[pos] = (long long)fcls;
      tbox_l[pos * 4 + 0] = tx - (float)gi;                                       // gxy - gij (clamped)   :241
      tbox_l[pos * 4 + 1] = ty - (float)gj;
      tbox_l[pos * 4 + 2] = tw;
      tbox_l[pos * 4 + 3] = th;
      anch_l[pos * 2 + 0] = aw;
      anch_l[pos * 2 + 1] = ah;
    }
    __syncthreads();
    if (threadIdx.x == 0) base_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[l] = base_s;
}

}  // namespace

extern "C" int mmi_build_targets(const float* targets, int nt, const float* anchors, int nl, int na,
                                 const int32_t* grids_dev, float anchor_t, int64_t* idx, int64_t* tcls, float* tbox,
                                 float* anch, int32_t* counts, void* stream) {
  MMI_CHECK_ARG(nl > 0 && na > 0 && nt >= 0, "mmi_build_targets: bad sizes");
  MMI_CHECK_ARG(anchors && grids_dev && counts, "mmi_build_targets: null pointer");
  MMI_CHECK_ARG(nt == 0 || (targets && idx && tcls && tbox && anch), "mmi_build_targets: null pointer");
  MMI_CHECK_ARG((int64_t)5 * na * nt < (1 << 30), "mmi_build_targets: too many targets");
  hipLaunchKernelGGL(build_targets_kernel, dim3(nl), dim3(TB), 0, (hipStream_t)stream, targets, nt, anchors, na,
                     (const int*)grids_dev, anchor_t, idx, tcls, tbox, anch, counts);
  MMI_CHECK_LAUNCH("mmi_build_targets");
  return MMI_OK;
}
