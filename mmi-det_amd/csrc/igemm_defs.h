// Shared declarations of the implicit-GEMM translation units (igemm.hip: planners + C ABI; igemm_fwd*.hip / igemm_dgrad*.hip:
// the forward / data-gradient kernel instantiations; igemm_wgrad.hip: the weight-gradient kernels).  The kernels used to live
// in one file that took 2.5 minutes to compile; split by template arguments the pieces build in parallel.
#pragma once
#include "common.h"

namespace mmi_ig {

#ifndef MMI_BK
#define MMI_BK 32
#endif
constexpr int BK = MMI_BK;            // K-slab depth (32; 64 is an experiment: half the barriers per MFMA, two workgroups per CU)
constexpr int KT = BK / 4;            // loader threads per tile row (one float4 each)
constexpr int RPP = 256 / KT;         // tile rows covered by one pass of the 256 loader threads
#ifndef MMI_IGEMM_STAGES
#define MMI_IGEMM_STAGES 1  // LDS stages of the fwd/dgrad kernel: 1 = single buffer + register prefetch (3 waves/SIMD,
                            // measured +3 % over the double-buffered 2-waves/SIMD form); 2 = double buffer
#endif
constexpr int LDS_PAD = BK + 4;  // floats per [row][k] LDS row
// which MFMA group (0..3) of the current slab issues prefetch load number i of the next slab
#ifndef MMI_LOAD_SPREAD
#define MMI_LOAD_SPREAD(i, g) ((i) % 3 == (g))
#endif

#define ZERO_SRC p.zero

struct IgemmP {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  float* stat_part;
  const float* zero;  // 16 zero bytes in global memory (source of masked lanes)
  int M, Ncol, Kc, KH, KW, P, Q, Hs, Ws, lda, ldc, stride, pad, Ktot, ldb, mtiles, ntiles;
  int par;  // dgrad of a stride-2 conv: blockIdx.y = output-pixel parity class, which only sees its own taps
  const float* res;  // inference epilogue (Model.fuse()): y = act(acc + bias) + res[row * ldr + col]; null = no residual
  int ldr, act;
  float* sk_slots;  // stream-K: 2 partial-tile slots of BM*BN floats per workgroup
  int* sk_count;    // stream-K: per-tile arrival counters (zero before and after every launch)
  // token-side Linear epilogues (mmi_linear_epilogue): MMI_EPI_*
  int epi, ldaux, ldaux_out;
  const float* aux;
  float* aux_out;
  uint64_t seed;
  const uint64_t* seed_dev;
  uint32_t drop_thresh;
  float inv_keep;
  // uniform-tap loaders: byte extents of the A tensor (incl. the margin in front of it) and of the weights; of the output
  // (0: too large for 31-bit offsets, the epilogue keeps its pointer stores)
  uint32_t a_bytes, b_bytes, c_bytes;
  // BatchNorm statistics finished inside this launch (mmi_conv_bn_fwd): the workgroups that arrive last fold the partial
  // rows (stat_arrive, common.h) and write mean / 1/sqrt(var + eps), the running statistics and num_batches_tracked, so no
  // "finalize" launch follows the convolution.  bn_mi == null: the partial rows are all there is (mmi_conv_fwd).
  StatFold bn_fold;
  float* bn_mi;   // mean at [col], 1/std at [mi_stride + col]
  int mi_stride;  // Ncol, or the channel count of a wider mean_invstd vector this layer owns a column block of (twin launches)
  float* bn_rmean;
  float* bn_rvar;
  int64_t* bn_nbt;
  int bn_nnbt;
  float bn_eps, bn_momentum;
  double bn_inv_rows, bn_unbias;  // 1 / rows, rows / (rows - 1)
  // pre-split operands (igemm_kernel<..., T8>, t8.hip): the T8 images of A and / or B (same element indexing, 6 bytes per element)
  const void* A8;
  const void* B8;
  // dgrad only (mmi_conv_dgrad_bnred*, round 4): the output dx IS the incoming gradient of the BatchNorm + activation whose output the
  // convolution read, so the epilogue -- which holds the finished dx tile -- takes that BatchNorm's backward REDUCTION along:
  // dz = dx * act'(xhat * gamma + beta), partial column sums of dz and dz * xhat per row block into stat_part[mtile][2][Ncol]
  // (the layout mmi_bn_act_bwd_apply folds).  bnr_y: the BatchNorm's input (that layer's raw conv output), row stride bnr_ldy;
  // mean / invstd through bn_mi / mi_stride; null = off.
  const float* bnr_y;
  const float* bnr_g;
  const float* bnr_b;
  int bnr_ldy, bnr_act;
};

// ------------------------------------------------------------------------------------------------------------------
// weight gradient: C[co][(tap,ci)] = sum_pix dy[pix][co] * x[gather(pix,tap)][ci]; both operands are K(pixel)-strided,
// tiles live in LDS as [k][m] / [k][n] and the MFMA operands are conflict-free ds_read_b32.
struct WgradP {
  const float* DY;
  const float* X;
  float* OUT;   // dw, or slab base when splits > 1
  float* OUTB;  // bias gradient (column sums of dy) of split 0, or null; split z writes OUTB + z * slab_stride
  const float* zero;
  int Mpix, Cout, Cin, KH, KW, Ho, Wo, H, W, stride, pad, ldx, ldy, Ntot, chunk, mtiles, ntiles, splits;
  int64_t slab_stride;
  uint32_t x_bytes;  // TAB loaders: byte extent of x including the margin in front of it
  // split-K fold inside the launch: per-tile arrival counters (zero before and after); the workgroup that completes a
  // tile's last split sums the splits' partial tiles in split order (deterministic) into DW (and DB): no reduce launch
  int* cnt;
  int cnt_per_tile;  // counters of one tile: sum over the tree's levels of ceil(nodes / 4)
  float* DW;
  float* DB;
  // TAB loaders: the per-pixel {source offset, invalid-tap mask} table of the layer's geometry, precomputed once
  // (mmi_conv_wgrad_table_build: it depends on shapes and strides only, not on data); null = built in the kernel, slab by slab
  const uint2* tab;
  // pre-split operands (wgrad_kernel<..., T8>): the T8 images of dy and x
  const void* DY8;
  const void* X8;
};

// Twin launches (two problems of one shape in one grid, blockIdx.z = problem): the kernels take ONE parameter block by value, as
// they always did, plus the byte distances from problem 0's operands to problem 1's.  Problem 1's workgroups add them to their
// copy of the pointers -- a handful of scalar adds behind a uniform branch.  (Selecting between two whole parameter blocks by
// reference looked the same in the source but cost 2-4x the VALU instructions in the K loops -- the compiler no longer kept
// the loaders' address parts scalar -- and with them 10-20 % of the dgrad / wgrad kernels' speed: profiles/r03_twin_param_select.txt.)
struct IgemmDelta {
  int64_t A, B, C, stat_part, sk_slots, sk_count, aux, aux_out, fold_part, fold_l1, fold_cnt, bn_mi, bn_rmean, bn_rvar, bn_nbt, A8, B8;
  int64_t bnr_y, bnr_g, bnr_b;
};
struct WgradDelta {
  int64_t DY, X, OUT, OUTB, cnt, DW, DB, DY8, X8;
};
template <typename T>
__device__ __forceinline__ T* shift_ptr(T* p, int64_t bytes) {
  return p == nullptr ? p : reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (uintptr_t)bytes);
}
inline int64_t ptr_delta(const void* to, const void* from) { return (to == nullptr || from == nullptr) ? 0 : (int64_t)((const char*)to - (const char*)from); }

struct FwdPlan {
  int bm, bn, mtiles, ntiles;   // per problem
  int sk_grid;  // > 0: stream-K schedule over this many workgroups (needs the workspace), 0: one workgroup per tile
};
struct WgPlan {
  int bm, bn, mtiles, ntiles, splits, chunk;
  bool vec;
};

// Mode 6 of mmi_set_gemm_precision ("exact, fastest kernel per shape"): every entry point that plans or launches a GEMM holds a
// PrecScope, which replaces the global mode by 0 (fp32 MFMA) or 3 (nine bf16 products) for the duration of the call -- both forms
// make every fp32 product exactly and accumulate in fp32 -- by a rule on the layer's shape and direction (0 fwd, 1 dgrad, 2 wgrad)
// taken from the per-shape in-step timings of the two modes (profiles/r04_gemm_by_shape_fp32_vs_bf16x9.txt).
int pick_exact_prec(const mmi_conv_desc* d, int dir);
struct PrecScope {
  int saved;
  PrecScope(const mmi_conv_desc* d, int dir);
  ~PrecScope();
};

// planner / tuning state (igemm.hip)
extern int g_uniform_loaders, g_gemm_prec, g_tile_bm, g_tile_bn, g_wgrad_force[3], g_sk_slots;
const float* zero_src();   // 16 zero bytes in device memory: the source of masked lanes (see the loaders)
int device_cus();
int check_desc(const mmi_conv_desc* d, const char* who);

constexpr int SK_MAX_TILES = 65536;                                  // arrival counters at the head of the workspace
constexpr size_t SK_COUNTER_BYTES = (size_t)SK_MAX_TILES * sizeof(int);
// Workspace of a forward / dgrad launch (zero-filled when first handed over, self-cleaning afterwards):
//   [0, SK_COUNTER_BYTES)            stream-K arrival counters
//   [.., + BN_COUNTER_BYTES)         arrival counters of the in-launch BatchNorm statistics fold (forward)
//   [WS_HEADER_BYTES, ..)            no zero-fill needed: level-1 statistics partials (forward), then the stream-K slots
// The header is the same for every shape and direction, because launches on one stream share one buffer.
constexpr size_t BN_COUNTER_BYTES = (size_t)MMI_STAT_MAX_COUNTERS * sizeof(int);
constexpr size_t WS_HEADER_BYTES = SK_COUNTER_BYTES + BN_COUNTER_BYTES;
constexpr int WG_MAX_TILES = 4096;
constexpr size_t WG_COUNTER_BYTES = (size_t)WG_MAX_TILES * sizeof(int);
inline size_t bn_l1_bytes(const FwdPlan& f, int Ncol) {
  const int G = stat_group_size(f.mtiles);
  return (((size_t)cdiv(f.mtiles, G) * 2 * Ncol * sizeof(float)) + 15) & ~(size_t)15;
}
inline bool bn_fold_fits(const FwdPlan& f) {
  const int ngroups = cdiv(f.mtiles, stat_group_size(f.mtiles));
  return (int64_t)ngroups * f.ntiles + f.ntiles <= MMI_STAT_MAX_COUNTERS;
}
inline size_t sk_slot_bytes(const FwdPlan& f) { return f.sk_grid > 0 ? (size_t)f.sk_grid * 2 * f.bm * f.bn * sizeof(float) : 0; }
inline size_t sk_workspace_bytes(const FwdPlan& f) { return f.sk_grid > 0 ? WS_HEADER_BYTES + sk_slot_bytes(f) : 0; }
inline size_t fwd_workspace_bytes(const FwdPlan& f, int Ncol) { return WS_HEADER_BYTES + bn_l1_bytes(f, Ncol) + sk_slot_bytes(f); }

// kernel launchers: defined in igemm_launch.h, instantiated once per (DGRAD, EPI) in igemm_fwd*.hip / igemm_dgrad*.hip
template <bool DGRAD, bool EPI>
int launch_igemm(const IgemmP& p0, const FwdPlan& f, bool vec, void* workspace, size_t workspace_bytes, hipStream_t s,
                 size_t slot_offset = WS_HEADER_BYTES, const IgemmP* twin = nullptr);   // twin != null: both problems carry their own sk_count / sk_slots
template <bool DGRAD, bool EPI>
int launch_igemm_bf16(IgemmP p, const FwdPlan& f, hipStream_t s, const char* who);
template <bool DGRAD>
int sk_occupancy(int bn);
// the pre-split-operand variants (igemm_kernel<..., T8>), instantiated in igemm_fwd_t8.hip / igemm_dgrad_t8.hip
template <bool DGRAD>
int launch_igemm_t8(const IgemmP& p, const IgemmDelta& q, const FwdPlan& f, dim3 grid, int t8, hipStream_t s);
// the deep-prefetch variants (igemm_kernel<..., PF2>), instantiated in igemm_fwd_pf2.hip / igemm_dgrad_pf2.hip
template <bool DGRAD>
int launch_igemm_pf2(const IgemmP& p, const IgemmDelta& q, const FwdPlan& f, dim3 grid, int t8, hipStream_t s);
extern int g_pf2;   // mmi_set_deep_prefetch / MMIDET_PF2: 1 = take the deep-prefetch variants where they exist
// T8 images announced for the next GEMM launch of this thread (mmi_gemm_operands_t8): [0] = A, [1] = B, [2], [3] = the twin problem's
const void** t8_pending();

}  // namespace mmi_ig
