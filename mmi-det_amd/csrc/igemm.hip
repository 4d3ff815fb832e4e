// fp32-MFMA implicit-GEMM convolution for gfx950: forward, data-gradient, weight-gradient.
//
// Replaces (SURVEY.md §8a rows 4,5,6,9,11,16) the ATen conv2d / linear forward+backward reached from
// models/common.py:114,764,772,333,337,1167-1170,1254,1257 and models/yolo_test.py:44 of the reference.
//
// Design (MI355X-first, not a cuDNN-style port):
//  * activations NHWC, weights OHWI  -> the GEMM K axis (tap, channel) is contiguous in both operands, every global
//    access is a 16-byte lane access over 128-byte row segments;
//  * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD); 256-thread workgroups = 4 waves in a 2x2 grid, each wave
//    owning (BM/2)x(BN/2) of the BMxBN tile as 32x32 accumulator tiles;
//  * A/B K-slabs of 32 staged through LDS, double buffered, one barrier per K-step; next slab's global loads are issued
//    before the MFMA block so HBM/L2 latency hides behind 4096 cycles of matrix work;
//  * LDS rows padded 32->36 floats so the ds_read_b128 operand fetches are bank-conflict-free; one b128 per lane
//    feeds FOUR k-steps (the K order inside a slab is permuted identically for A and B, which a GEMM sum allows);
//  * XCD-aware tile order: all output-channel tiles of a pixel tile run on one XCD (shared 4 MiB L2);
//  * epilogue fuses bias and the BatchNorm batch-statistics partial sums (no extra pass over y for mean/var).
#include "igemm_defs.h"

namespace mmi_ig {
// Invalid lanes of the branch-free tile loaders read this instead of being masked afterwards: no select on the loaded
// value, so hipcc does not have to wait for the load where it is issued (it would: `ok ? v : 0` forces vmcnt(0)).
// (The pointer travels as a kernel argument so that it stays in the global address space: selecting against the
// symbol itself degrades every tile load to a flat_load.)
__device__ f32x4 g_zero4 = {0.f, 0.f, 0.f, 0.f};

const float* zero_src() {
  static const float* ptr[64] = {nullptr};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (ptr[dev] == nullptr) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero4)) == hipSuccess) ptr[dev] = (const float*)q;
  }
  return ptr[dev];
}

// mmi_set_uniform_loaders / MMIDET_UNIFORM_LOADERS=0 (A/B switch): 1 = use the uniform-tap loaders where they apply
int g_uniform_loaders = getenv("MMIDET_UNIFORM_LOADERS") ? atoi(getenv("MMIDET_UNIFORM_LOADERS")) : 1;
int g_gemm_prec = 0;  // mmi_set_gemm_precision: 0 = exact fp32 MFMA, 1 = split-bf16 products for forward-layout GEMMs
int g_tile_bm = 0, g_tile_bn = 0;  // mmi_set_tile_override (tuning): force one tile variant, one workgroup per tile
int g_wgrad_force[3] = {0, 0, 0};  // mmi_set_wgrad_override (tuning): bm, bn, splits (0 = automatic)
int g_pf2 = getenv("MMIDET_PF2") ? atoi(getenv("MMIDET_PF2")) : 0;   // mmi_set_deep_prefetch (A/B switch)
int g_sk_slots = 0;  // mmi_set_streamk_slots: 0 = chip-sized, > 0 = this many workgroups, < 0 = schedule off

int device_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n = prop.multiProcessorCount;
    else
      n = 256;  // MI355X; also what a GPU-less host plans with
    (void)hipGetLastError();
  }
  return n;
}

int check_desc(const mmi_conv_desc* d, const char* who) {
  MMI_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  MMI_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "%s: non-positive dims", who);
  MMI_CHECK_ARG(d->KH == d->KW && (d->KH == 1 || d->KH == 3), "%s: kernel %dx%d unsupported (1x1, 3x3)", who, d->KH, d->KW);
  MMI_CHECK_ARG(d->stride == 1 || d->stride == 2, "%s: stride %d unsupported", who, d->stride);
  MMI_CHECK_ARG(d->pad == d->KH / 2, "%s: pad %d != k/2", who, d->pad);
  MMI_CHECK_ARG(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
                "%s: output dims (%d,%d) inconsistent", who, d->Ho, d->Wo);
  MMI_CHECK_ARG(d->ldx >= d->Cin && d->ldy >= d->Cout, "%s: row strides smaller than channel counts", who);
  MMI_CHECK_ARG((int64_t)d->N * d->H * d->W < (1LL << 31) && (int64_t)d->KH * d->KW * d->Cin < (1 << 24), "%s: size overflow", who);
  return MMI_OK;
}
int pick_exact_prec(const mmi_conv_desc* d, int dir) {
  if (d == nullptr || d->Cin % 32 != 0 || d->Cout % 32 != 0) return 0;     // (the nine-product kernels want the uniform-tap / pixel-table loaders)
  const int64_t kk = (int64_t)d->Cin * d->Cout;
  if (d->KH == 3) {
    if (d->Cin >= 128) return 3;                     // every direction of the 128..1024-channel 3x3 layers
    return dir == 1 ? 3 : 0;                         // 64-channel 3x3 at 160x160: the data gradient only
  }
  return kk >= (1 << 22) && dir != 2 ? 3 : 0;        // the large token-side projections (1024 x 4096 ...), forward and data gradient
}
PrecScope::PrecScope(const mmi_conv_desc* d, int dir) : saved(g_gemm_prec) {
  if (saved == 6) g_gemm_prec = pick_exact_prec(d, dir);
}
PrecScope::~PrecScope() { g_gemm_prec = saved; }
}  // namespace mmi_ig
using namespace mmi_ig;

namespace {
// nprob: problems of this shape that share the launch (twin launches: 2); the grid the rules below look at is theirs together
FwdPlan plan_tiles(int64_t M, int Ncol, int nprob = 1) {
  FwdPlan f;
  if (g_tile_bm > 0) {
    f.bm = g_tile_bm;
    f.bn = g_tile_bn;
    f.mtiles = cdiv(M, f.bm);
    f.ntiles = cdiv(Ncol, f.bn);
    f.sk_grid = 0;
    return f;
  }
  f.bn = Ncol > 64 ? 128 : 64;
  f.bm = 128;
  // small problems: shrink the tile until there are >= 2 workgroups per CU (256 CUs)
  if (f.bn == 128 && (int64_t)cdiv(M, 128) * cdiv(Ncol, 128) * nprob < 512) f.bn = 64;
  if ((int64_t)cdiv(M, 128) * cdiv(Ncol, f.bn) * nprob < 512) f.bm = 64, f.bn = 64;
  f.mtiles = cdiv(M, f.bm);
  f.ntiles = cdiv(Ncol, f.bn);
  f.sk_grid = 0;
  return f;
}

// Schedule choice for the vector kernels.  A kernel's time grows in steps of one workgroup per CU (769 tiles of 128x128
// cost as much as 1024), so:
//  * long K (>= 32 slabs per resident workgroup): stream-K over 128-wide tiles unless one workgroup per tile already
//    fills the chip evenly (>= 93 % of the last "layer" of 256);
//  * short K (1x1 convs, token projections; tools/sweep_tiles.py): one workgroup per tile, plan_tiles' shrink rule.
template <bool DGRAD>
FwdPlan plan_igemm(int64_t M, int Ncol, int Ktot, bool vec, bool allow_sk, int nprob = 1) {
  FwdPlan f = plan_tiles(M, Ncol, nprob);
  if (!vec && f.bn == 128) f.bn = 64, f.ntiles = cdiv(Ncol, 64);
  static const bool off = getenv("MMIDET_NO_STREAMK") != nullptr;
  const int nk = cdiv(Ktot, BK);
  if (!vec || !allow_sk || g_tile_bm > 0 || g_gemm_prec == 5) return f;   // (mode 5: short launches, one workgroup per tile)
  FwdPlan g;
  g.bm = 128;
  g.bn = Ncol > 64 ? 128 : 64;
  g.mtiles = cdiv(M, 128);
  g.ntiles = cdiv(Ncol, g.bn);
  g.sk_grid = 0;
  const int64_t tiles1 = (int64_t)g.mtiles * g.ntiles, tiles = tiles1 * nprob;   // per problem, in the launch
  const int cus = device_cus();
  const int slots = g_sk_slots > 0 ? g_sk_slots : cus * sk_occupancy<DGRAD>(g.bn);
  const double layers = (double)tiles / cus, dp_eff = layers / ceil(layers);
  const bool sk_ok = !off && g_sk_slots >= 0 && nk >= (g_sk_slots > 0 ? 2 : 16) && tiles1 <= SK_MAX_TILES &&
                     tiles * nk >= (int64_t)(g_sk_slots > 0 ? 1 : 32) * slots && tiles * nk < (1LL << 31);
  if (sk_ok) {
    if (g_sk_slots == 0 && tiles >= slots && dp_eff >= 0.93) return g;  // already even
    // every problem runs its own stream-K iteration space on an equal share of the resident slots (a multiple of 8, so that
    // workgroup b of either problem lands on XCD b % 8: xcd_remap)
    g.sk_grid = slots;
    if (nprob > 1) {
      g.sk_grid = slots / nprob;
      if (g_sk_slots == 0) g.sk_grid &= ~7;   // (a forced test grid keeps its size)
      if (g.sk_grid < 1) g.sk_grid = 1;
    }
    return g;
  }
  // short K on many rows (the 1x1 convs): with the uniform-tap loaders the 64x64 tile no longer pays more address
  // arithmetic per MFMA than the wide ones, and its finer grid wins 10-20 % stand-alone (tools/sweep_tiles.py)
  // (in the step: 126.7 -> 125.8 ms; MMIDET_SHORTK_TILE=0 restores plan_tiles' choice)
  static const int shortk_tile = getenv("MMIDET_SHORTK_TILE") ? atoi(getenv("MMIDET_SHORTK_TILE")) : 64;
  if (shortk_tile == 64 && g_gemm_prec == 0 && g_uniform_loaders && nk <= 64 && Ncol <= 1024 && M >= 8192) {
    f.bm = f.bn = 64;
    f.mtiles = cdiv(M, 64);
    f.ntiles = cdiv(Ncol, 64);
    return f;
  }
  // short K, or too little work per workgroup for stream-K: one workgroup per tile with the tile shrunk until the grid has
  // two workgroups per CU (plan_tiles).  Finer 64x64 tiles win 10-16 % on the 1x1 layers stand-alone
  // (profiles/r01_tile_sweep_short_k.txt) but nothing inside the step, where the other lane fills the tail; always taking the
  // 128-wide tile loses 4 %.
  return f;
}

bool fwd_vec(const mmi_conv_desc* d) { return d->Cin % 4 == 0 && d->ldx % 4 == 0; }
FwdPlan fwd_plan(const mmi_conv_desc* d, int nprob = 1) {
  return plan_igemm<false>((int64_t)d->N * d->Ho * d->Wo, d->Cout, d->KH * d->KW * d->Cin, fwd_vec(d), true, nprob);
}
bool dgrad_vec(const mmi_conv_desc* d) { return d->Cout % 4 == 0 && d->ldy % 4 == 0 && d->Cin % 4 == 0; }
bool dgrad_par(const mmi_conv_desc* d) { return dgrad_vec(d) && d->stride == 2 && d->KH == 3; }
FwdPlan dgrad_plan(const mmi_conv_desc* d, int nprob = 1) {
  // parity mode: the grid is sized for the largest class (ceil(H/2) x ceil(W/2) pixels per image); it keeps the
  // data-parallel schedule (four K extents in one launch)
  const bool par = dgrad_par(d);
  const int64_t mrows = par ? (int64_t)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2) : (int64_t)d->N * d->H * d->W;
  return plan_igemm<true>(mrows, d->Cin, d->KH * d->KW * d->Cout, dgrad_vec(d), !par, nprob);
}

}  // namespace

extern "C" int mmi_set_gemm_precision(int mode) {
  MMI_CHECK_ARG((mode >= 0 && mode <= 3) || mode == 5 || mode == 6,
                "mmi_set_gemm_precision: mode %d (0 = fp32 MFMA, 1 = bf16x3, 2 = bf16x6, 3 = bf16x9, 5 = bf16x1, 6 = exact: 0 or 3 per shape)", mode);
  g_gemm_prec = mode;
  return MMI_OK;
}

extern "C" int mmi_set_uniform_loaders(int on) {
  const int old = g_uniform_loaders;
  g_uniform_loaders = on ? 1 : 0;
  return old;
}

extern "C" int mmi_set_tile_override(int bm, int bn) {
  const bool ok = (bm == 0 && bn == 0) || (bm == 128 && (bn == 128 || bn == 64)) || (bm == 64 && bn == 64);
  MMI_CHECK_ARG(ok, "mmi_set_tile_override: (%d,%d) is not a kernel variant (128x128, 128x64, 64x64; 0,0 = automatic)", bm, bn);
  g_tile_bm = bm;
  g_tile_bn = bn;
  return MMI_OK;
}

extern "C" int mmi_set_wgrad_override(int bm, int bn, int splits) {
  const bool ok = (bm == 0 && bn == 0) || ((bm == 128 || bm == 64) && (bn == 128 || bn == 64));
  MMI_CHECK_ARG(ok && splits >= 0, "mmi_set_wgrad_override: (%d,%d,%d): tiles are 128/64 x 128/64 (0,0 = automatic), splits >= 0 (0 = off)", bm, bn, splits);
  g_wgrad_force[0] = bm; g_wgrad_force[1] = bn; g_wgrad_force[2] = splits;
  return MMI_OK;
}

namespace mmi_ig {
const void** t8_pending() {
  static thread_local const void* pend[4] = {nullptr, nullptr, nullptr, nullptr};
  return pend;
}
}  // namespace mmi_ig

extern "C" int mmi_gemm_operands_t8(const void* a_t8, const void* b_t8, const void* a_t8_twin, const void* b_t8_twin) {
  const void** pend = t8_pending();
  pend[0] = a_t8; pend[1] = b_t8; pend[2] = a_t8_twin; pend[3] = b_t8_twin;
  return MMI_OK;
}

extern "C" size_t mmi_workspace_header_bytes(int kind) {
  switch (kind) {
    case 0: return WS_HEADER_BYTES;                               // forward / dgrad (one problem)
    case 1: return 2 * WS_HEADER_BYTES;                           // forward / dgrad twin launch: [header 0 | header 1 | bodies]
    case 2: return WG_COUNTER_BYTES;                              // weight gradient (one problem)
    case 3: return 2 * WG_COUNTER_BYTES;                          // weight gradient twin launch
    case 4: return (size_t)MMI_STAT_MAX_COUNTERS * sizeof(int);   // BatchNorm backward (mmi_bn_act_bwd*)
    default: return 0;
  }
}

extern "C" int mmi_set_deep_prefetch(int on) {
  const int old = g_pf2;
  g_pf2 = on ? 1 : 0;
  return old;
}

extern "C" int mmi_set_streamk_slots(int slots) {
  const int old = g_sk_slots;
  g_sk_slots = slots;
  return old;
}

extern "C" int mmi_conv_fwd_row_blocks(const mmi_conv_desc* d) {
  const PrecScope prec_scope_(d, 0);
  if (check_desc(d, "mmi_conv_fwd_row_blocks") != MMI_OK) return MMI_ERR_ARG;
  if (mmi_smallconv_supported(d)) return mmi_smallconv_blocks(d);  // CEM layers: direct VALU conv (cem.hip)
  return fwd_plan(d).mtiles;
}

extern "C" size_t mmi_conv_fwd_workspace(const mmi_conv_desc* d) {
  const PrecScope prec_scope_(d, 0);
  if (check_desc(d, "mmi_conv_fwd_workspace") != MMI_OK || mmi_smallconv_supported(d)) return 0;
  return fwd_workspace_bytes(fwd_plan(d), d->Cout);
}

extern "C" size_t mmi_conv_dgrad_workspace(const mmi_conv_desc* d) {
  const PrecScope prec_scope_(d, 1);
  if (check_desc(d, "mmi_conv_dgrad_workspace") != MMI_OK || mmi_smallconv_dgrad_supported(d)) return 0;
  return sk_workspace_bytes(dgrad_plan(d));
}

namespace {
int conv_fwd_impl(const float* x, const float* w, const float* bias, const float* residual, int ldr, int act, float* y,
                  float* stat_partials, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream,
                  const char* who, const mmi_bn_stats* bn = nullptr) {
  const PrecScope prec_scope_(d, 0);
  if (int e = check_desc(d, who)) return e;
  MMI_CHECK_ARG(x && w && y, "%s: null pointer", who);
  MMI_CHECK_ARG(!(bias && stat_partials), "%s: bias and BN statistics are mutually exclusive", who);
  const bool plain = residual == nullptr && act == MMI_ACT_NONE;
  const int64_t rows = (int64_t)d->N * d->Ho * d->Wo;
  if (mmi_smallconv_supported(d) && plain) {
    MMI_CHECK_ARG(d->Cin != 24 || ((uintptr_t)x & 15) == 0, "%s: operands must be 16-byte aligned", who);
    if (int e = mmi_smallconv_fwd(x, w, bias, y, stat_partials, d, (hipStream_t)stream)) return e;
    if (bn == nullptr) return MMI_OK;   // the CEM's direct convolutions keep the separate fold
    if (int e = mmi_bn_finalize(stat_partials, mmi_smallconv_blocks(d), rows, d->Cout, bn->eps, bn->momentum, bn->running_mean,
                                bn->running_var, bn->num_batches_tracked, bn->mean_invstd, stream)) return e;
    if (bn->num_batches_tracked2 != nullptr) {
      mmi_set_error("%s: two num_batches_tracked counters are not supported on this path", who);
      return MMI_ERR_ARG;
    }
    return MMI_OK;
  }
  const bool vec = fwd_vec(d);
  MMI_CHECK_ARG(!vec || (((uintptr_t)x | (uintptr_t)w) & 15) == 0, "%s: operands must be 16-byte aligned", who);
  IgemmP p{};
  p.A = x; p.B = w; p.C = y; p.bias = bias; p.stat_part = stat_partials;
  p.res = residual; p.ldr = ldr; p.act = act;
  p.M = d->N * d->Ho * d->Wo; p.Ncol = d->Cout; p.Kc = d->Cin; p.KH = d->KH; p.KW = d->KW;
  p.P = d->Ho; p.Q = d->Wo; p.Hs = d->H; p.Ws = d->W; p.lda = d->ldx; p.ldc = d->ldy;
  p.stride = d->stride; p.pad = d->pad; p.Ktot = d->KH * d->KW * d->Cin; p.ldb = p.Ktot;
  const FwdPlan f = fwd_plan(d);
  static const bool bn_fold_off = getenv("MMIDET_BN_FOLD") != nullptr && atoi(getenv("MMIDET_BN_FOLD")) == 0;  // (A/B switch)
  const bool fold = bn != nullptr && !bn_fold_off && bn_fold_fits(f) &&
                    (bn->num_batches_tracked2 == nullptr || bn->num_batches_tracked2 == bn->num_batches_tracked + 1);
  if (fold) {
    MMI_CHECK_ARG(stat_partials && bn->mean_invstd, "%s: BN statistics need the partials buffer and mean_invstd", who);
    MMI_CHECK_ARG((bn->running_mean == nullptr) == (bn->running_var == nullptr), "%s: running stats must come in pairs", who);
    if (workspace == nullptr || workspace_bytes < fwd_workspace_bytes(f, d->Cout) || ((uintptr_t)workspace & 15)) {
      mmi_set_error("%s: needs a 16-byte aligned zero-initialised workspace of %zu bytes (got %zu)", who, fwd_workspace_bytes(f, d->Cout),
                    workspace_bytes);
      return MMI_ERR_WORKSPACE;
    }
    const int G = stat_group_size(f.mtiles);
    p.bn_fold = StatFold{stat_partials, (float*)((char*)workspace + WS_HEADER_BYTES), (int*)((char*)workspace + SK_COUNTER_BYTES),
                         f.mtiles, d->Cout, f.ntiles, G};
    p.bn_mi = bn->mean_invstd;
    p.bn_rmean = bn->running_mean; p.bn_rvar = bn->running_var;
    p.bn_nbt = bn->num_batches_tracked;
    p.bn_nnbt = bn->num_batches_tracked == nullptr ? 0 : (bn->num_batches_tracked2 != nullptr ? 2 : 1);
    p.bn_eps = bn->eps; p.bn_momentum = bn->momentum;
    p.bn_inv_rows = 1.0 / (double)rows; p.bn_unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
  }
  if (int e = launch_igemm<false, false>(p, f, vec, workspace, workspace_bytes, (hipStream_t)stream, WS_HEADER_BYTES + bn_l1_bytes(f, d->Cout)))
    return e;
  if (bn != nullptr && !fold) {  // (a list too long for the counter block, or counters that are not adjacent: separate fold)
    if (int e = mmi_bn_finalize(stat_partials, f.mtiles, rows, d->Cout, bn->eps, bn->momentum, bn->running_mean, bn->running_var,
                                bn->num_batches_tracked, bn->mean_invstd, stream)) return e;
    if (bn->num_batches_tracked2 != nullptr) {   // (rare path: the second counter takes a launch of its own)
      if (int e = mmi_i64_increment(bn->num_batches_tracked2, stream)) return e;
    }
  }
  return MMI_OK;
}
}  // namespace

extern "C" int mmi_conv_bn_fwd(const float* x, const float* w, float* y, float* stat_partials, const mmi_bn_stats* bn,
                               void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream) {
  MMI_CHECK_ARG(bn != nullptr && stat_partials != nullptr, "mmi_conv_bn_fwd: null BN arguments");
  return conv_fwd_impl(x, w, nullptr, nullptr, 0, MMI_ACT_NONE, y, stat_partials, workspace, workspace_bytes, d, stream,
                       "mmi_conv_bn_fwd", bn);
}

namespace {
int fill_epilogue(IgemmP& p, const mmi_conv_desc* d, const mmi_linear_epilogue* e, bool dgrad, const float* out, const char* who) {
  MMI_CHECK_ARG(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0, "%s: 1x1 descriptors only", who);
  if (e == nullptr || e->kind == MMI_EPI_NONE) return MMI_OK;
  const int cols = dgrad ? d->Cin : d->Cout;
  const bool fwd_kind = e->kind == MMI_EPI_DROPOUT_RESIDUAL || e->kind == MMI_EPI_GELU;
  const bool bwd_kind = e->kind == MMI_EPI_GELU_GRAD || e->kind == MMI_EPI_ACCUMULATE;
  MMI_CHECK_ARG(dgrad ? bwd_kind : fwd_kind, "%s: epilogue kind %d does not belong to this direction", who, e->kind);
  if (e->kind == MMI_EPI_GELU) {
    MMI_CHECK_ARG(e->aux_out != nullptr && e->ldaux_out >= cols, "%s: GELU epilogue needs aux_out with ldaux_out >= %d", who, cols);
  } else {
    MMI_CHECK_ARG(e->aux != nullptr && e->ldaux >= cols, "%s: epilogue needs aux with ldaux >= %d", who, cols);
    // only the accumulate form may read what it writes (element by element, same thread)
    MMI_CHECK_ARG(e->kind == MMI_EPI_ACCUMULATE || e->aux != out, "%s: aux aliases the output", who);
  }
  if (e->kind == MMI_EPI_DROPOUT_RESIDUAL)
    MMI_CHECK_ARG(e->p_drop >= 0.f && e->p_drop < 1.f, "%s: dropout probability %g", who, (double)e->p_drop);
  p.epi = e->kind; p.aux = e->aux; p.ldaux = e->ldaux; p.aux_out = e->aux_out; p.ldaux_out = e->ldaux_out;
  p.seed = e->seed; p.seed_dev = e->seed_dev;
  p.drop_thresh = e->kind == MMI_EPI_DROPOUT_RESIDUAL ? drop_thresh(e->p_drop) : 0u;
  p.inv_keep = e->kind == MMI_EPI_DROPOUT_RESIDUAL ? 1.0f / (1.0f - e->p_drop) : 1.0f;
  return MMI_OK;
}
}  // namespace

extern "C" int mmi_linear_fwd_fused(const float* x, const float* w, const float* bias, float* y, void* workspace,
                                    size_t workspace_bytes, const mmi_conv_desc* d, const mmi_linear_epilogue* e,
                                    void* stream) {
  const PrecScope prec_scope_(d, 0);
  if (int err = check_desc(d, "mmi_linear_fwd_fused")) return err;
  MMI_CHECK_ARG(x && w && y, "mmi_linear_fwd_fused: null pointer");
  const bool vec = fwd_vec(d);
  MMI_CHECK_ARG(!vec || (((uintptr_t)x | (uintptr_t)w) & 15) == 0, "mmi_linear_fwd_fused: operands must be 16-byte aligned");
  IgemmP p{};
  if (int err = fill_epilogue(p, d, e, false, y, "mmi_linear_fwd_fused")) return err;
  p.A = x; p.B = w; p.C = y; p.bias = bias;
  p.M = d->N * d->Ho * d->Wo; p.Ncol = d->Cout; p.Kc = d->Cin; p.KH = 1; p.KW = 1;
  p.P = d->Ho; p.Q = d->Wo; p.Hs = d->H; p.Ws = d->W; p.lda = d->ldx; p.ldc = d->ldy;
  p.stride = 1; p.pad = 0; p.Ktot = d->Cin; p.ldb = p.Ktot;
  if (p.epi == MMI_EPI_NONE)  // a plain Linear: the convolution instantiation (buffer-store epilogue, fewer registers)
    return launch_igemm<false, false>(p, fwd_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
  return launch_igemm<false, true>(p, fwd_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int mmi_linear_dgrad_fused(const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                                      const mmi_conv_desc* d, const mmi_linear_epilogue* e, void* stream) {
  const PrecScope prec_scope_(d, 1);
  if (int err = check_desc(d, "mmi_linear_dgrad_fused")) return err;
  MMI_CHECK_ARG(dy && w && dx, "mmi_linear_dgrad_fused: null pointer");
  const bool vec = dgrad_vec(d);
  MMI_CHECK_ARG(!vec || (((uintptr_t)dy | (uintptr_t)w) & 15) == 0, "mmi_linear_dgrad_fused: operands must be 16-byte aligned");
  IgemmP p{};
  if (int err = fill_epilogue(p, d, e, true, dx, "mmi_linear_dgrad_fused")) return err;
  p.A = dy; p.B = w; p.C = dx;
  p.M = d->N * d->H * d->W; p.Ncol = d->Cin; p.Kc = d->Cout; p.KH = 1; p.KW = 1;
  p.P = d->H; p.Q = d->W; p.Hs = d->Ho; p.Ws = d->Wo; p.lda = d->ldy; p.ldc = d->ldx;
  p.stride = 1; p.pad = 0; p.Ktot = d->Cout; p.ldb = d->Cin;
  if (p.epi == MMI_EPI_NONE)
    return launch_igemm<true, false>(p, dgrad_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
  return launch_igemm<true, true>(p, dgrad_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int mmi_conv_fwd(const float* x, const float* w, const float* bias, float* y, float* stat_partials,
                            void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream) {
  return conv_fwd_impl(x, w, bias, nullptr, 0, MMI_ACT_NONE, y, stat_partials, workspace, workspace_bytes, d, stream,
                       "mmi_conv_fwd");
}

extern "C" int mmi_conv_bias_act_fwd(const float* x, const float* w, const float* bias, const float* residual, int ldr,
                                     int act, float* y, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d,
                                     void* stream) {
  MMI_CHECK_ARG(act == MMI_ACT_NONE || act == MMI_ACT_SILU || act == MMI_ACT_LEAKY, "mmi_conv_bias_act_fwd: unknown activation %d", act);
  MMI_CHECK_ARG(residual == nullptr || (d && ldr >= d->Cout), "mmi_conv_bias_act_fwd: residual row stride < Cout");
  return conv_fwd_impl(x, w, bias, residual, ldr, act, y, nullptr, workspace, workspace_bytes, d, stream,
                       "mmi_conv_bias_act_fwd");
}

extern "C" int mmi_conv_dgrad(const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                              const mmi_conv_desc* d, void* stream) {
  const PrecScope prec_scope_(d, 1);
  if (int e = check_desc(d, "mmi_conv_dgrad")) return e;
  MMI_CHECK_ARG(dy && w && dx, "mmi_conv_dgrad: null pointer");
  if (mmi_smallconv_dgrad_supported(d)) return mmi_smallconv_dgrad(dy, w, dx, d, (hipStream_t)stream);
  // A = dy (channels Cout), output columns = Cin
  const bool vec = dgrad_vec(d);
  MMI_CHECK_ARG(!vec || (((uintptr_t)dy | (uintptr_t)w) & 15) == 0, "mmi_conv_dgrad: operands must be 16-byte aligned");
  IgemmP p{};
  p.A = dy; p.B = w; p.C = dx; p.bias = nullptr; p.stat_part = nullptr;
  p.M = d->N * d->H * d->W; p.Ncol = d->Cin; p.Kc = d->Cout; p.KH = d->KH; p.KW = d->KW;
  p.P = d->H; p.Q = d->W; p.Hs = d->Ho; p.Ws = d->Wo; p.lda = d->ldy; p.ldc = d->ldx;
  p.stride = d->stride; p.pad = d->pad; p.Ktot = d->KH * d->KW * d->Cout; p.ldb = d->KH * d->KW * d->Cin;
  p.par = dgrad_par(d) ? 1 : 0;
  return launch_igemm<true, false>(p, dgrad_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- input gradient with the BatchNorm backward reduction of the layer below in its epilogue (IgemmP::bnr_y) ----------------
namespace {
void fill_dgrad(IgemmP& p, const float* dy, const float* w, float* dx, const mmi_conv_desc* d) {
  p.A = dy; p.B = w; p.C = dx;
  p.M = d->N * d->H * d->W; p.Ncol = d->Cin; p.Kc = d->Cout; p.KH = d->KH; p.KW = d->KW;
  p.P = d->H; p.Q = d->W; p.Hs = d->Ho; p.Ws = d->Wo; p.lda = d->ldy; p.ldc = d->ldx;
  p.stride = d->stride; p.pad = d->pad; p.Ktot = d->KH * d->KW * d->Cout; p.ldb = d->KH * d->KW * d->Cin;
  p.par = dgrad_par(d) ? 1 : 0;
}
int set_bn_hook(IgemmP& p, const mmi_bn_reduce_hook* h, const mmi_conv_desc* d, const char* who) {
  MMI_CHECK_ARG(h->y && h->mean_invstd && h->gamma && h->beta && h->partials, "%s: null pointer in the BatchNorm hook", who);
  MMI_CHECK_ARG(h->ldy >= d->Cin && h->mi_stride >= d->Cin, "%s: hook strides < Cin", who);
  MMI_CHECK_ARG(h->act == MMI_ACT_NONE || h->act == MMI_ACT_SILU || h->act == MMI_ACT_LEAKY, "%s: unknown activation %d", who, h->act);
  MMI_CHECK_ARG(!dgrad_par(d) && !mmi_smallconv_dgrad_supported(d), "%s: stride-1 MFMA layers only", who);
  p.bnr_y = h->y; p.bnr_ldy = h->ldy; p.bnr_g = h->gamma; p.bnr_b = h->beta; p.bnr_act = h->act;
  p.bn_mi = const_cast<float*>(h->mean_invstd); p.mi_stride = h->mi_stride;
  p.stat_part = h->partials;
  return MMI_OK;
}
}  // namespace

extern "C" int mmi_conv_dgrad_row_blocks_n(const mmi_conv_desc* d, int nprob) {
  const PrecScope prec_scope_(d, 1);
  if (check_desc(d, "mmi_conv_dgrad_row_blocks_n") != MMI_OK || nprob < 1 || nprob > 2) return MMI_ERR_ARG;
  return dgrad_plan(d, nprob).mtiles;
}

extern "C" int mmi_conv_dgrad_bnred(const float* dy, const float* w, float* dx, const float* skip, int ldskip,
                                    const mmi_bn_reduce_hook* hook, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d,
                                    void* stream) {
  const PrecScope prec_scope_(d, 1);
  const char* who = "mmi_conv_dgrad_bnred";
  if (int e = check_desc(d, who)) return e;
  MMI_CHECK_ARG(dy && w && dx && hook, "%s: null pointer", who);
  const bool vec = dgrad_vec(d);
  MMI_CHECK_ARG(!vec || (((uintptr_t)dy | (uintptr_t)w) & 15) == 0, "%s: operands must be 16-byte aligned", who);
  MMI_CHECK_ARG(skip == nullptr || (vec && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && ldskip >= d->Cin && ldskip % 4 == 0),
                "%s: the skip accumulation exists for vector-shaped 1x1 stride-1 layers", who);
  IgemmP p{};
  fill_dgrad(p, dy, w, dx, d);
  if (int e = set_bn_hook(p, hook, d, who)) return e;
  if (skip != nullptr) {
    p.epi = MMI_EPI_ACCUMULATE; p.aux = skip; p.ldaux = ldskip; p.inv_keep = 1.0f;
    return launch_igemm<true, true>(p, dgrad_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
  }
  return launch_igemm<true, false>(p, dgrad_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- twin launches ----------------------------------------------------------------------------------------------------
// The RGB and IR backbones of the two-stream model (models/yolo_test.py:162-273 of the reference walks them layer by layer) run
// the same layer shapes on different weights.  A twin entry point carries both problems in ONE launch (blockIdx.z = problem):
// twice the tiles per launch (wave quantisation, ramp and drain are paid once), half the launches, and no second HIP stream
// racing for the same CUs.  The two problems' activations are usually the two channel halves of one NHWC buffer (row strides
// ldx / ldy span both), so everything per-channel around the GEMM (BatchNorm, activation, pooling) runs once over the buffer.
// Workspace of a twin launch: [header of problem 0 | header of problem 1 | body 0 | body 1].  The headers (arrival counters, kept
// at zero between launches by the kernels themselves) sit at FIXED offsets whatever the shape, because launches of different
// shapes share one buffer per stream: a counter block at a shape-dependent offset would be another shape's scratch.
namespace {
size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
size_t twin_body_bytes(const FwdPlan& f, int Ncol_for_bn) { return align256((Ncol_for_bn ? bn_l1_bytes(f, Ncol_for_bn) : 0) + sk_slot_bytes(f)); }
char* twin_hdr(void* ws, int g) { return (char*)ws + (size_t)g * WS_HEADER_BYTES; }
char* twin_body(void* ws, int g, size_t body) { return (char*)ws + 2 * WS_HEADER_BYTES + (size_t)g * body; }
void fill_fwd(IgemmP& p, const float* x, const float* w, float* y, float* part, const mmi_conv_desc* d) {
  p.A = x; p.B = w; p.C = y; p.bias = nullptr; p.stat_part = part;
  p.M = d->N * d->Ho * d->Wo; p.Ncol = d->Cout; p.Kc = d->Cin; p.KH = d->KH; p.KW = d->KW;
  p.P = d->Ho; p.Q = d->Wo; p.Hs = d->H; p.Ws = d->W; p.lda = d->ldx; p.ldc = d->ldy;
  p.stride = d->stride; p.pad = d->pad; p.Ktot = d->KH * d->KW * d->Cin; p.ldb = p.Ktot;
}
bool twin_shape_ok(const mmi_conv_desc* d) { return !mmi_smallconv_supported(d) && !mmi_smallconv_dgrad_supported(d); }
}  // namespace

extern "C" int mmi_conv_fwd_row_blocks_n(const mmi_conv_desc* d, int nprob) {
  const PrecScope prec_scope_(d, 0);
  if (check_desc(d, "mmi_conv_fwd_row_blocks_n") != MMI_OK || nprob < 1 || nprob > 2) return MMI_ERR_ARG;
  return fwd_plan(d, nprob).mtiles;
}

extern "C" size_t mmi_conv_fwd_workspace_n(const mmi_conv_desc* d, int nprob) {
  const PrecScope prec_scope_(d, 0);
  if (check_desc(d, "mmi_conv_fwd_workspace_n") != MMI_OK || nprob < 1 || nprob > 2 || !twin_shape_ok(d)) return 0;
  if (nprob == 1) return fwd_workspace_bytes(fwd_plan(d, 1), d->Cout);
  return 2 * WS_HEADER_BYTES + 2 * twin_body_bytes(fwd_plan(d, 2), d->Cout);
}

extern "C" size_t mmi_conv_dgrad_workspace_n(const mmi_conv_desc* d, int nprob) {
  const PrecScope prec_scope_(d, 1);
  if (check_desc(d, "mmi_conv_dgrad_workspace_n") != MMI_OK || nprob < 1 || nprob > 2 || !twin_shape_ok(d)) return 0;
  if (nprob == 1) return sk_workspace_bytes(dgrad_plan(d, 1));
  const FwdPlan f = dgrad_plan(d, 2);
  return f.sk_grid > 0 ? 2 * WS_HEADER_BYTES + 2 * twin_body_bytes(f, 0) : 0;
}

extern "C" int mmi_conv_bn_fwd2(const float* const* x, const float* const* w, float* const* y, float* const* stat_partials,
                                const mmi_bn_stats* bn, int mi_stride, void* workspace, size_t workspace_bytes,
                                const mmi_conv_desc* d, void* stream) {
  const PrecScope prec_scope_(d, 0);
  const char* who = "mmi_conv_bn_fwd2";
  if (int e = check_desc(d, who)) return e;
  MMI_CHECK_ARG(x && w && y && stat_partials && bn, "%s: null argument arrays", who);
  MMI_CHECK_ARG(twin_shape_ok(d), "%s: the CEM's direct convolutions have no twin form", who);
  const bool vec = fwd_vec(d);
  const FwdPlan f = fwd_plan(d, 2);
  MMI_CHECK_ARG(bn_fold_fits(f), "%s: statistics row list too long for the in-launch fold", who);
  MMI_CHECK_ARG(mi_stride >= d->Cout, "%s: mi_stride %d < Cout", who, mi_stride);
  const size_t body = twin_body_bytes(f, d->Cout), need = 2 * WS_HEADER_BYTES + 2 * body;
  if (workspace == nullptr || workspace_bytes < need || ((uintptr_t)workspace & 255)) {
    mmi_set_error("%s: needs a 256-byte aligned zero-initialised workspace of %zu bytes (got %zu)", who, need, workspace_bytes);
    return MMI_ERR_WORKSPACE;
  }
  const int64_t rows = (int64_t)d->N * d->Ho * d->Wo;
  IgemmP p[2];
  for (int g = 0; g < 2; ++g) {
    MMI_CHECK_ARG(x[g] && w[g] && y[g] && stat_partials[g] && bn[g].mean_invstd, "%s: null pointer (problem %d)", who, g);
    MMI_CHECK_ARG(!vec || (((uintptr_t)x[g] | (uintptr_t)w[g]) & 15) == 0, "%s: operands must be 16-byte aligned", who);
    MMI_CHECK_ARG((bn[g].running_mean == nullptr) == (bn[g].running_var == nullptr), "%s: running stats must come in pairs", who);
    MMI_CHECK_ARG(bn[g].num_batches_tracked2 == nullptr || bn[g].num_batches_tracked2 == bn[g].num_batches_tracked + 1,
                  "%s: the two num_batches_tracked counters of a problem must be adjacent", who);
    p[g] = IgemmP{};
    fill_fwd(p[g], x[g], w[g], y[g], stat_partials[g], d);
    p[g].sk_count = (int*)twin_hdr(workspace, g);
    p[g].sk_slots = (float*)(twin_body(workspace, g, body) + bn_l1_bytes(f, d->Cout));
    p[g].bn_fold = StatFold{stat_partials[g], (float*)twin_body(workspace, g, body), (int*)(twin_hdr(workspace, g) + SK_COUNTER_BYTES), f.mtiles,
                            d->Cout, f.ntiles, stat_group_size(f.mtiles)};
    p[g].bn_mi = bn[g].mean_invstd; p[g].mi_stride = mi_stride;
    p[g].bn_rmean = bn[g].running_mean; p[g].bn_rvar = bn[g].running_var;
    p[g].bn_nbt = bn[g].num_batches_tracked;
    p[g].bn_nnbt = bn[g].num_batches_tracked == nullptr ? 0 : (bn[g].num_batches_tracked2 != nullptr ? 2 : 1);
    p[g].bn_eps = bn[g].eps; p[g].bn_momentum = bn[g].momentum;
    p[g].bn_inv_rows = 1.0 / (double)rows; p[g].bn_unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
  }
  return launch_igemm<false, false>(p[0], f, vec, workspace, workspace_bytes, (hipStream_t)stream, 0, &p[1]);
}

// dx[g] = conv_transpose(dy[g], w[g]) [+ skip[g]]; skip (row stride ldskip) only for 1x1 stride-1 layers (GEMM epilogue), else NULL
namespace {
int conv_dgrad2_impl(const float* const* dy, const float* const* w, float* const* dx, const float* const* skip, int ldskip,
                     const mmi_bn_reduce_hook* hooks, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream,
                     const char* who) {
  const PrecScope prec_scope_(d, 1);
  if (int e = check_desc(d, who)) return e;
  MMI_CHECK_ARG(dy && w && dx, "%s: null argument arrays", who);
  MMI_CHECK_ARG(twin_shape_ok(d), "%s: the CEM's direct convolutions have no twin form", who);
  const bool vec = dgrad_vec(d);
  const bool acc = skip != nullptr && skip[0] != nullptr;
  MMI_CHECK_ARG(!acc || (skip[1] != nullptr && vec && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && ldskip >= d->Cin && ldskip % 4 == 0),
                "%s: the skip accumulation exists for vector-shaped 1x1 stride-1 layers (both problems)", who);
  const FwdPlan f = dgrad_plan(d, 2);
  const size_t body = twin_body_bytes(f, 0), need = f.sk_grid > 0 ? 2 * WS_HEADER_BYTES + 2 * body : 0;
  if (need != 0 && (workspace == nullptr || workspace_bytes < need || ((uintptr_t)workspace & 255))) {
    mmi_set_error("%s: needs a 256-byte aligned zero-initialised workspace of %zu bytes (got %zu)", who, need, workspace_bytes);
    return MMI_ERR_WORKSPACE;
  }
  IgemmP p[2];
  for (int g = 0; g < 2; ++g) {
    MMI_CHECK_ARG(dy[g] && w[g] && dx[g], "%s: null pointer (problem %d)", who, g);
    MMI_CHECK_ARG(!vec || (((uintptr_t)dy[g] | (uintptr_t)w[g]) & 15) == 0, "%s: operands must be 16-byte aligned", who);
    p[g] = IgemmP{};
    fill_dgrad(p[g], dy[g], w[g], dx[g], d);
    if (need != 0) {
      p[g].sk_count = (int*)twin_hdr(workspace, g);
      p[g].sk_slots = (float*)twin_body(workspace, g, body);
    }
    if (acc) {
      p[g].epi = MMI_EPI_ACCUMULATE; p[g].aux = skip[g]; p[g].ldaux = ldskip; p[g].inv_keep = 1.0f;
    }
    if (hooks != nullptr)
      if (int e = set_bn_hook(p[g], hooks + g, d, who)) return e;
  }
  if (acc) return launch_igemm<true, true>(p[0], f, vec, workspace, workspace_bytes, (hipStream_t)stream, 0, &p[1]);
  return launch_igemm<true, false>(p[0], f, vec, workspace, workspace_bytes, (hipStream_t)stream, 0, &p[1]);
}
}  // namespace

extern "C" int mmi_conv_dgrad2(const float* const* dy, const float* const* w, float* const* dx, const float* const* skip, int ldskip,
                               void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream) {
  return conv_dgrad2_impl(dy, w, dx, skip, ldskip, nullptr, workspace, workspace_bytes, d, stream, "mmi_conv_dgrad2");
}

// ... with each problem's BatchNorm backward reduction in the epilogue (hooks[2]; partial rows: mmi_conv_dgrad_row_blocks_n(d, 2))
extern "C" int mmi_conv_dgrad2_bnred(const float* const* dy, const float* const* w, float* const* dx, const float* const* skip, int ldskip,
                                     const mmi_bn_reduce_hook* hooks, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d,
                                     void* stream) {
  MMI_CHECK_ARG(hooks != nullptr, "mmi_conv_dgrad2_bnred: null hooks");
  return conv_dgrad2_impl(dy, w, dx, skip, ldskip, hooks, workspace, workspace_bytes, d, stream, "mmi_conv_dgrad2_bnred");
}

namespace {
FwdPlan fwd_plan_bf16(const mmi_conv_desc* d) { return plan_tiles((int64_t)d->N * d->Ho * d->Wo, d->Cout); }
FwdPlan dgrad_plan_bf16(const mmi_conv_desc* d) {
  const bool par = dgrad_par(d);
  const int64_t mrows = par ? (int64_t)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2) : (int64_t)d->N * d->H * d->W;
  return plan_tiles(mrows, d->Cin);
}
}  // namespace

extern "C" int mmi_conv_fwd_row_blocks_bf16(const mmi_conv_desc* d) {
  if (check_desc(d, "mmi_conv_fwd_row_blocks_bf16") != MMI_OK) return MMI_ERR_ARG;
  return fwd_plan_bf16(d).mtiles;
}

extern "C" size_t mmi_conv_fwd_workspace_bf16(const mmi_conv_desc* d) {
  if (check_desc(d, "mmi_conv_fwd_workspace_bf16") != MMI_OK) return 0;
  return WS_HEADER_BYTES + bn_l1_bytes(fwd_plan_bf16(d), d->Cout);
}

extern "C" int mmi_conv_fwd_bf16(const void* x, const float* w, const float* bias, void* y, float* stat_partials,
                                 const mmi_bn_stats* bn, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d,
                                 void* stream) {
  const char* who = "mmi_conv_fwd_bf16";
  if (int e = check_desc(d, who)) return e;
  MMI_CHECK_ARG(x && w && y, "%s: null pointer", who);
  MMI_CHECK_ARG(!(bias && stat_partials), "%s: bias and BN statistics are mutually exclusive", who);
  MMI_CHECK_ARG(fwd_vec(d) && d->ldy % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 7) == 0 && ((uintptr_t)w & 15) == 0,
                "%s: channel counts / row strides must be multiples of 4 and the operands aligned", who);
  IgemmP p{};
  p.A = (const float*)x; p.B = w; p.C = (float*)y; p.bias = bias; p.stat_part = stat_partials;
  p.M = d->N * d->Ho * d->Wo; p.Ncol = d->Cout; p.Kc = d->Cin; p.KH = d->KH; p.KW = d->KW;
  p.P = d->Ho; p.Q = d->Wo; p.Hs = d->H; p.Ws = d->W; p.lda = d->ldx; p.ldc = d->ldy;
  p.stride = d->stride; p.pad = d->pad; p.Ktot = d->KH * d->KW * d->Cin; p.ldb = p.Ktot;
  const FwdPlan f = fwd_plan_bf16(d);
  const int64_t rows = (int64_t)d->N * d->Ho * d->Wo;
  bool fold = false;
  if (bn != nullptr) {
    MMI_CHECK_ARG(stat_partials && bn->mean_invstd, "%s: BN statistics need the partials buffer and mean_invstd", who);
    fold = bn_fold_fits(f) && (bn->num_batches_tracked2 == nullptr || bn->num_batches_tracked2 == bn->num_batches_tracked + 1);
    if (fold) {
      if (workspace == nullptr || workspace_bytes < WS_HEADER_BYTES + bn_l1_bytes(f, d->Cout) || ((uintptr_t)workspace & 15)) {
        mmi_set_error("%s: needs a 16-byte aligned zero-initialised workspace of %zu bytes (got %zu)", who,
                      WS_HEADER_BYTES + bn_l1_bytes(f, d->Cout), workspace_bytes);
        return MMI_ERR_WORKSPACE;
      }
      p.bn_fold = StatFold{stat_partials, (float*)((char*)workspace + WS_HEADER_BYTES), (int*)((char*)workspace + SK_COUNTER_BYTES),
                           f.mtiles, d->Cout, f.ntiles, stat_group_size(f.mtiles)};
      p.bn_mi = bn->mean_invstd;
      p.bn_rmean = bn->running_mean; p.bn_rvar = bn->running_var;
      p.bn_nbt = bn->num_batches_tracked;
      p.bn_nnbt = bn->num_batches_tracked == nullptr ? 0 : (bn->num_batches_tracked2 != nullptr ? 2 : 1);
      p.bn_eps = bn->eps; p.bn_momentum = bn->momentum;
      p.bn_inv_rows = 1.0 / (double)rows; p.bn_unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
    }
  }
  if (int e = launch_igemm_bf16<false, false>(p, f, (hipStream_t)stream, who)) return e;
  if (bn != nullptr && !fold) {
    if (int e = mmi_bn_finalize(stat_partials, f.mtiles, rows, d->Cout, bn->eps, bn->momentum, bn->running_mean, bn->running_var,
                                bn->num_batches_tracked, bn->mean_invstd, stream)) return e;
    if (bn->num_batches_tracked2 != nullptr)
      if (int e = mmi_i64_increment(bn->num_batches_tracked2, stream)) return e;
  }
  return MMI_OK;
}

// dx = conv_transpose(dy, w) [+ skip]: dy, dx, skip bf16 (skip: 1x1 stride-1 layers only, row stride ldskip; may be NULL)
extern "C" int mmi_conv_dgrad_bf16(const void* dy, const float* w, void* dx, const void* skip, int ldskip, const mmi_conv_desc* d,
                                   void* stream) {
  const char* who = "mmi_conv_dgrad_bf16";
  if (int e = check_desc(d, who)) return e;
  MMI_CHECK_ARG(dy && w && dx, "%s: null pointer", who);
  MMI_CHECK_ARG(dgrad_vec(d) && d->ldx % 4 == 0 && (((uintptr_t)dy | (uintptr_t)dx) & 7) == 0 && ((uintptr_t)w & 15) == 0,
                "%s: channel counts / row strides must be multiples of 4 and the operands aligned", who);
  IgemmP p{};
  p.A = (const float*)dy; p.B = w; p.C = (float*)dx;
  p.M = d->N * d->H * d->W; p.Ncol = d->Cin; p.Kc = d->Cout; p.KH = d->KH; p.KW = d->KW;
  p.P = d->H; p.Q = d->W; p.Hs = d->Ho; p.Ws = d->Wo; p.lda = d->ldy; p.ldc = d->ldx;
  p.stride = d->stride; p.pad = d->pad; p.Ktot = d->KH * d->KW * d->Cout; p.ldb = d->KH * d->KW * d->Cin;
  p.par = dgrad_par(d) ? 1 : 0;
  if (skip != nullptr) {
    MMI_CHECK_ARG(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && ldskip >= d->Cin && skip != dx,
                  "%s: the skip accumulation exists for 1x1 stride-1 layers", who);
    p.epi = MMI_EPI_ACCUMULATE; p.aux = (const float*)skip; p.ldaux = ldskip;
    return launch_igemm_bf16<true, true>(p, dgrad_plan_bf16(d), (hipStream_t)stream, who);
  }
  return launch_igemm_bf16<true, false>(p, dgrad_plan_bf16(d), (hipStream_t)stream, who);
}

