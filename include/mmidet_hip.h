/* mmidet_hip.h — C ABI of libmmidet_hip.so (gfx950 / MI355X).
 *
 * The reference (joewybean/MMI-Det) is pure Python on PyTorch and has NO FFI of its own: every
 * device op it runs is whatever ATen dispatches.  This header is therefore the build-defined boundary underneath the
 * preserved Python surface (models.yolo_test.Model / utils.loss.ComputeLoss).  Each entry point names the reference
 * call site(s) (file:line under /root/reference) whose ATen ops it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless said otherwise; fp32 unless said otherwise
 *   - activations are NHWC ("rows" = N*H*W pixels, channels contiguous) with an explicit row stride `ld*` in elements,
 *     so a tensor may be a channel slice of a wider buffer
 *   - conv weights are OHWI = [Cout][KH][KW][Cin] (the physical layout of a torch channels_last (Cout,Cin,KH,KW) tensor)
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work (no host sync, no allocation)
 *   - return 0 on success, negative on error; mmi_last_error() returns a static description for the calling thread
 */
#ifndef MMIDET_HIP_H
#define MMIDET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMI_OK 0
#define MMI_ERR_ARG (-1)
#define MMI_ERR_LAUNCH (-2)
#define MMI_ERR_WORKSPACE (-3)

#define MMI_ACT_NONE 0
#define MMI_ACT_SILU 1   /* models/common.py:117 nn.SiLU */
#define MMI_ACT_LEAKY 2  /* models/common.py:767,775 nn.LeakyReLU(0.1) */

int mmi_version(void);
const char* mmi_last_error(void);
/* Bytes at the head of a zero-initialised workspace that hold arrival counters and are zero again after every launch (the
 * kernels that elect a last arriver reset what they used).  kind: 0 = forward / dgrad workspace, 1 = the twin-launch form of it,
 * 2 = weight-gradient workspace, 3 = its twin-launch form, 4 = BatchNorm backward workspace.  Test hook: a debug run checks
 * that the region is all zero between launches (mmidet_hip/ops.py::check_counters). */
size_t mmi_workspace_header_bytes(int kind);

/* ---- pre-split operands ("T8") for the three-term bf16 GEMM modes (mmi_set_gemm_precision 2 / 3) ------------------------------
 * In those modes every fp32 operand value is the sum of three bf16 terms and a product is 6 / 9 bf16 MFMA products; by default the
 * kernels split each tile when they stage it -- a weight tile once per row tile of the launch, an activation row once per column
 * tile and tap.  A T8 image holds the terms of a tensor, split ONCE by its producer: per 8 consecutive channels of a row 48 bytes =
 * [term 0: 8 bf16 | term 1 | term 2], same element indexing as the tensor (element (row, c) with row stride ld lives in the group at
 * byte (row * ld + (c & ~7)) * 6; channel counts, slice offsets and row strides multiples of 8, 16-byte aligned).  The terms are the
 * ones the kernels' own split produces, so a GEMM on images is bit-identical to the same GEMM on the fp32 tensors.
 * mmi_split_t8: the stand-alone converter (src row stride ld, image row stride ld8, both in elements).
 * mmi_gemm_operands_t8: announces images for the NEXT GEMM launch issued by the calling thread (mmi_conv_fwd / _bn_fwd / _dgrad /
 * mmi_linear_* without epilogue / the twin forms / mmi_conv_wgrad*), which takes and clears them whatever path it then runs:
 *   forward: a = image of x, b = image of w;  dgrad: a = image of dy, b = image of w;  wgrad: a = image of dy, b = image of x;
 *   *_twin: the second problem of a twin launch.  NULL = that operand is split in the kernel (forward / dgrad: weights alone may be
 *   pre-split; wgrad needs both).  Ignored in the other precision modes and by shapes that take the general loaders. */
int mmi_split_t8(const float* src, int ld, void* dst, int ld8, int64_t rows, int C, void* stream);
int mmi_gemm_operands_t8(const void* a_t8, const void* b_t8, const void* a_t8_twin, const void* b_t8_twin);

/* ---- conv / linear as fp32-MFMA implicit GEMM -------------------------------------------------------------------
 * Replaces nn.Conv2d forward/backward at models/common.py:114 (Conv), 764,772 (CEM convs), 333,337 (FFM 1x1),
 * models/yolo_test.py:44 (Detect) and nn.Linear at models/common.py:1167-1170,1254,1257 (a Linear is the 1x1 case
 * with H=W=1, N=rows).  Descriptor: x is (N,H,W,Cin) row stride ldx; y is (N,Ho,Wo,Cout) row stride ldy. */
typedef struct {
  int32_t N, H, W, Cin;
  int32_t Ho, Wo, Cout;
  int32_t KH, KW, stride, pad;
  int32_t ldx, ldy;
} mmi_conv_desc;

/* y = conv(x, w) [+ bias].  If stat_partials != NULL the epilogue also writes per-row-block column sums of y and y*y
 * (BatchNorm batch statistics, models/common.py:116) to stat_partials[rb][2][Cout], rb < mmi_conv_fwd_row_blocks().
 *
 * Workspace (forward and dgrad): shapes whose tile count is a poor multiple of the chip run a stream-K schedule (equal
 * K-slab shares per resident workgroup, partial tiles folded in K order by the last contributor: deterministic) and need
 * mmi_conv_{fwd,dgrad}_workspace(d) bytes of device memory, 16-byte aligned; 0 means none (NULL is then fine; the
 * forward query also covers mmi_conv_bn_fwd's statistics fold and is never 0 for an implicit-GEMM shape).  The
 * buffer must be ZERO-FILLED when first handed over; every launch leaves it ready for the next one, of any shape, as
 * long as launches sharing a buffer are ordered on one stream. */
int mmi_conv_fwd_row_blocks(const mmi_conv_desc* d);
/* Epilogue forms of the token-side Linear layers, so that a transformer block (models/common.py:1237-1267,
 * SelfAttention 1147-1235) is 9 dependent kernels forward and 11 backward instead of 14 and 20:
 *   MMI_EPI_DROPOUT_RESIDUAL  y = aux + dropout(x w^T + bias)      out_proj / mlp[2] + nn.Dropout + the residual add
 *                             (common.py:1173-1174,1233,1258,1263-1266); mask index = row * Cout + col, as mmi_dropout
 *                             over the contiguous (rows, Cout) tensor
 *   MMI_EPI_GELU              aux_out = x w^T + bias; y = GELU(aux_out)          mlp[0] + nn.GELU (common.py:1254-1256)
 *   MMI_EPI_GELU_GRAD         (dgrad) dx = (dy w) * GELU'(aux)                   backward of the same pair
 *   MMI_EPI_ACCUMULATE        (dgrad) dx = aux + dy w; aux may alias dx          sum of the q/k/v input gradients */
#define MMI_EPI_NONE 0
#define MMI_EPI_DROPOUT_RESIDUAL 1
#define MMI_EPI_GELU 2
#define MMI_EPI_GELU_GRAD 3
#define MMI_EPI_ACCUMULATE 4
typedef struct {
  int32_t kind;
  int32_t ldaux, ldaux_out;
  float p_drop;
  const float* aux;
  float* aux_out;
  uint64_t seed;
  const uint64_t* seed_dev;
} mmi_linear_epilogue;
/* 1x1 descriptors only (a Linear); workspace as for mmi_conv_fwd / mmi_conv_dgrad with the same descriptor. */
int mmi_linear_fwd_fused(const float* x, const float* w, const float* bias, float* y, void* workspace,
                         size_t workspace_bytes, const mmi_conv_desc* d, const mmi_linear_epilogue* e, void* stream);
int mmi_linear_dgrad_fused(const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                           const mmi_conv_desc* d, const mmi_linear_epilogue* e, void* stream);
/* Tuning/testing knob of the stream-K planner: 0 = size the grid to the chip (default), n > 0 = n workgroups for every
 * shape with >= 2 K slabs (lets small test shapes take the schedule), n < 0 = schedule off.  Returns the old value.
 * Changes what the *_workspace() and row_blocks() queries answer: set it before planning a call, not between. */
int mmi_set_streamk_slots(int slots);
/* Arithmetic of the conv / linear GEMMs (forward, dgrad, wgrad).  0 (default): exact fp32 products on
 * v_mfma_f32_32x32x2_f32.  Opt-in split forms on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, each fp32 operand split
 * into bf16 terms when its tile is staged into LDS: 1 = two terms, three products (relative product error <= 2^-16);
 * 2 = three terms, the six products of total order <= 2 (dropped terms <= 2^-24 per product); 3 = three terms, all
 * nine products (each fp32 product exact).  Modes 2 and 3 are indistinguishable from mode 0 on a single GEMM (3e-7..1e-6) and on
 * full-depth training gradients (profiles/r01_gemm_modes_full_size_gradients.txt); mode 1 is not (gradients 1e-2).
 * A process-wide switch; takes effect at the next launch. */
int mmi_set_gemm_precision(int mode);
/* A/B switch of the forward/dgrad tile loaders: 1 (default) = uniform-tap buffer-load loaders wherever the channel count is a
 * multiple of 32 and the tensors are below 2 GiB (same arithmetic, same results bit for bit; ~10 instead of ~100 address
 * instructions per K slab), 0 = the general cursor-based loaders everywhere.  Returns the old value. */
int mmi_set_uniform_loaders(int on);
/* A/B switch of the forward/dgrad K loop: 1 = deep prefetch (double-buffered LDS tile, global loads two K slabs ahead in two
 * register sets, one barrier per slab, two workgroups per CU) wherever the uniform-tap loaders apply, 0 = single LDS stage with the
 * next slab prefetched into registers (three workgroups per CU).  Same arithmetic and summation order: results are bit-identical.
 * Initial value from MMIDET_PF2 (default: see DESIGN.md).  Returns the old value. */
int mmi_set_deep_prefetch(int on);
/* Tuning knob: force the forward/dgrad tile variant (128x128, 128x64 or 64x64; one workgroup per tile, stream-K off);
 * (0,0) restores the planner.  Used by tools/sweep_tiles.py to calibrate the planner's cost model. */
int mmi_set_tile_override(int bm, int bn);
/* tuning hook of the weight-gradient planner (tools/sweep_wgrad.py): force the tile variant (128/64 x 128/64; 0,0 = the planner's)
 * and the split-K count of every following mmi_conv_wgrad* call; splits = 0 switches the override off. */
int mmi_set_wgrad_override(int bm, int bn, int splits);
size_t mmi_conv_fwd_workspace(const mmi_conv_desc* d);
int mmi_conv_fwd(const float* x, const float* w, const float* bias, float* y, float* stat_partials, void* workspace,
                 size_t workspace_bytes, const mmi_conv_desc* d, void* stream);
/* Training-mode Conv (models/common.py:116 `self.bn(self.conv(x))`): y = conv(x, w) AND the BatchNorm batch statistics
 * finished in the same launch.  The epilogue writes the per-row-block partials as mmi_conv_fwd does; the workgroups that
 * arrive last fold them (fixed order, fp64: run-to-run bit-identical) and write mean_invstd[0..C) = batch mean,
 * mean_invstd[C..2C) = 1/sqrt(biased var + eps), update running_mean/var (torch semantics: unbiased variance, momentum)
 * and increment *num_batches_tracked -- what mmi_bn_finalize would do in a launch of its own.  A layer that merges two
 * convolutions of the reference over one input (C3's cv1 | cv2, models/common.py:645-650) passes both counters; they must
 * be adjacent int64 words.  Workspace: mmi_conv_fwd_workspace(d) bytes, zero-filled when first handed over (the arrival
 * counters live in it); always needed. */
typedef struct {
  float eps, momentum;
  float* running_mean;            /* (C) or NULL (then running_var too) */
  float* running_var;
  int64_t* num_batches_tracked;   /* or NULL */
  int64_t* num_batches_tracked2;  /* NULL, or num_batches_tracked + 1 */
  float* mean_invstd;             /* out, (2C) */
} mmi_bn_stats;
int mmi_conv_bn_fwd(const float* x, const float* w, float* y, float* stat_partials, const mmi_bn_stats* bn, void* workspace,
                    size_t workspace_bytes, const mmi_conv_desc* d, void* stream);
/* Inference form of Conv after Model.fuse() (models/common.py:124-125 fuseforward, utils/torch_utils.py:181-201):
 * y = act(conv(x, w) + bias) [+ residual]; act is an MMI_ACT_* code, residual (row stride ldr) may be NULL.  Same
 * workspace rule as mmi_conv_fwd. */
int mmi_conv_bias_act_fwd(const float* x, const float* w, const float* bias, const float* residual, int ldr, int act,
                          float* y, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream);
/* dx = conv_transpose(dy, w): gradient w.r.t. the input (autograd of the call sites above). dx has row stride ldx. */
size_t mmi_conv_dgrad_workspace(const mmi_conv_desc* d);
int mmi_conv_dgrad(const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                   const mmi_conv_desc* d, void* stream);
/* dw (OHWI) = sum over pixels dy^T x; dbias (nullable, Cout floats) = column sums of dy, taken from the same dy tiles
 * (the bias gradient of Detect / Linear costs no extra pass).  The workspace (query its size first; 16-byte aligned) holds
 * per-tile arrival counters at its head -- ZERO-FILLED when first handed over, self-cleaning afterwards, launches sharing it
 * ordered on one stream -- and the split-K slabs behind them: the workgroup that completes a tile's last split sums the
 * splits in split order (deterministic), so no reduce launch follows. */
size_t mmi_conv_wgrad_workspace(const mmi_conv_desc* d);
/* Optional: the layer's pixel table (for every output pixel the byte offset of its top-left source position and the mask
 * of taps that leave the image).  It depends on the descriptor only, not on data: build it ONCE per geometry
 * (mmi_conv_wgrad_table_bytes(d) bytes, 8-byte aligned; 0 = this shape uses none) and hand it to mmi_conv_wgrad_tab, whose
 * kernel then copies 32 entries per K slab instead of computing them next to its MFMA stream (~100 VALU instructions per
 * slab).  mmi_conv_wgrad is the same call with table = NULL. */
size_t mmi_conv_wgrad_table_bytes(const mmi_conv_desc* d);
int mmi_conv_wgrad_table_build(void* table, const mmi_conv_desc* d, void* stream);
int mmi_conv_wgrad_tab(const float* dy, const float* x, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                       const void* table, const mmi_conv_desc* d, void* stream);
int mmi_conv_wgrad(const float* dy, const float* x, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                   const mmi_conv_desc* d, void* stream);

/* ---- BatchNorm (training statistics) + activation (+ residual) --------------------------------------------------
 * Replaces nn.BatchNorm2d + nn.SiLU / nn.LeakyReLU at models/common.py:116-122, 766-767, 774-775 and the residual adds
 * at common.py:613, 799.  eps/momentum per utils/torch_utils.py:149-151. */
/* partials[rb][2][C] -> mean_invstd[2][C]; updates running_mean/var (unbiased var) and increments
 * *num_batches_tracked (int64) when non-NULL.  The partials buffer must have MMI_BN_FOLD_ROWS spare rows (of 2*C floats)
 * behind the nparts rows: long lists are folded there first. */
#define MMI_BN_FOLD_ROWS 64
int mmi_bn_finalize(const float* partials, int nparts, int64_t rows, int C, float eps, float momentum,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean_invstd,
                    void* stream);
/* eval mode: mean_invstd from running stats. */
int mmi_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean_invstd,
                      void* stream);
/* out = act(gamma*(y-mean)*invstd+beta) [+ residual] */
int mmi_bn_act_fwd(const float* y, int ldy, const float* mean_invstd, const float* gamma, const float* beta,
                   const float* residual, int ldr, float* out, int ldo, int64_t rows, int C, int act, void* stream);
/* The same with the output channels split over two tensors: [0, split) -> out (row stride ldo), [split, C) -> out1 (ldo1).
 * C3's merged cv1 | cv2 convolution hands its first half to the bottleneck chain and writes the second straight into the
 * buffer cv3 reads (models/common.py:650 `torch.cat` without the copy).  split = C: one output; split % 4 == 0. */
int mmi_bn_act_fwd_split(const float* y, int ldy, const float* mean_invstd, const float* gamma, const float* beta,
                         const float* residual, int ldr, float* out, int ldo, float* out1, int ldo1, int split, int64_t rows,
                         int C, int act, void* stream);
/* backward, pass 1: partial column sums of dz and dz*xhat, dz = dout*act'(z): partials[pb][2][C], pb < nparts
 * (nparts = mmi_bn_bwd_parts(rows)). */
int mmi_bn_bwd_parts(int64_t rows);
int mmi_bn_act_bwd_reduce(const float* y, int ldy, const float* dout, int ldd, const float* mean_invstd,
                          const float* gamma, const float* beta, float* partials, int64_t rows, int C, int act,
                          void* stream);
/* backward, pass 2: sums partials -> dgamma,dbeta (C each) and writes dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)).
 * In eval-stat mode (frozen=1) dy = gamma*invstd*dz.  dy = NULL: the fold only (dgamma, dbeta). */
int mmi_bn_act_bwd_apply(const float* y, int ldy, const float* dout, int ldd, const float* mean_invstd,
                         const float* gamma, const float* beta, const float* partials, int nparts, float* dy, int lddy,
                         float* dgamma, float* dbeta, int64_t rows, int C, int act, int frozen, void* stream);
/* Both passes in one call, two launches: the reduce pass's last-arriving workgroups write dgamma / dbeta themselves (no
 * fold launch in between; fixed fold order, fp64).  dout may come in two tensors split at channel `split` (gradient of the
 * two outputs of mmi_bn_act_fwd_split), dgamma/dbeta then go to two parameter pairs; split = C: one of each (the *1
 * arguments may be NULL).  Workspace: mmi_bn_act_bwd_workspace(rows, C) bytes, 16-byte aligned, zero-filled when first
 * handed over (arrival counters at its head; self-cleaning; launches sharing it must be ordered on one stream).
 * dy = NULL: the sums only (dgamma / dbeta are finished when the call's launches are); the apply pass is then the caller's, on
 * its own way into a consumer (mmi_cem_conv2_wgrad_bn). */
size_t mmi_bn_act_bwd_workspace(int64_t rows, int C);
int mmi_bn_act_bwd(const float* y, int ldy, const float* dout, int ldd, const float* dout1, int ldd1, int split,
                   const float* mean_invstd, const float* gamma, const float* beta, void* workspace, size_t workspace_bytes,
                   float* dy, int lddy, float* dgamma, float* dbeta, float* dgamma1, float* dbeta1, int64_t rows, int C,
                   int act, int frozen, void* stream);

/* ---- channel maps and twin launches ---------------------------------------------------------------------------------------
 * The reference walks its RGB and IR backbones as two sequences of identical layers (models/yolo_test.py:162-273; the default
 * YAML's rows 0-2 / 3-5, 9-10 / 11-12, ...).  Here the two lanes' activations are the two channel halves of ONE NHWC buffer
 * (lane stride = C, row stride = 2C) and a "twin" entry point carries both lanes' GEMMs in one launch (gridDim.z = 2): twice the
 * tiles per launch, half the launches, no second stream racing for the CUs; everything per-channel around the GEMM (BatchNorm,
 * activation, residual, pooling) runs once over the 2C-channel buffer.
 *
 * mmi_bn_map describes such a buffer to the BatchNorm kernels.  Parameter blocks: channel c takes gamma/beta (and writes
 * dgamma/dbeta) of block c / blk, element c % blk -- one block for a plain layer, two for C3's merged cv1|cv2 conv, two for a twin
 * layer, four for a twin cv1|cv2.  Scatter: with lane = c / period and r = c % period, channel c of the activated output (of the
 * incoming gradient, in backward) lives at  t0 + lane * ls0 + r  when r < split, else at  t1 + lane * ls1 + (r - split).  Plain:
 * period = split = C.  All of blk, period, split, ls0, ls1 multiples of 4 keeps the 16-byte lane accesses. */
typedef struct {
  const float* gamma[4];
  const float* beta[4];
  float* dgamma[4];               /* backward only */
  float* dbeta[4];
  int32_t nblk, blk;
  int32_t period, split, ls0, ls1;
} mmi_bn_map;
int mmi_bn_act_fwd_map(const float* y, int ldy, const float* mean_invstd, const mmi_bn_map* map, const float* residual, int ldr,
                       float* out, int ldo, float* out1, int ldo1, int64_t rows, int C, int act, void* stream);
int mmi_bn_act_bwd_map(const float* y, int ldy, const float* dout, int ldd, const float* dout1, int ldd1, const float* mean_invstd,
                       const mmi_bn_map* map, void* workspace, size_t workspace_bytes, float* dy, int lddy, int64_t rows, int C,
                       int act, int frozen, void* stream);
/* Planner queries for a launch that carries `nprob` (1 or 2) problems of shape d: the tile plan looks at all the tiles of the
 * launch, so row-block counts and workspaces differ from the single-problem answers.  Workspaces cover ALL problems
 * (256-byte aligned, zero-filled when first handed over, self-cleaning). */
int mmi_conv_fwd_row_blocks_n(const mmi_conv_desc* d, int nprob);
size_t mmi_conv_fwd_workspace_n(const mmi_conv_desc* d, int nprob);
size_t mmi_conv_dgrad_workspace_n(const mmi_conv_desc* d, int nprob);
size_t mmi_conv_wgrad_workspace_n(const mmi_conv_desc* d, int nprob);
/* Twin training-mode Conv: y[g] = conv(x[g], w[g]) for g = 0, 1 with the BatchNorm batch statistics of both finished in the
 * launch (as mmi_conv_bn_fwd).  bn = array of two; problem g writes mean to bn[g].mean_invstd[0..Cout) and 1/std to
 * bn[g].mean_invstd[mi_stride ..) -- pass mean_invstd = mi + g * Cout and mi_stride = 2 * Cout for one (2, 2*Cout) vector that
 * the map kernels read.  stat_partials[g]: (row_blocks_n + 64) * 2 * Cout floats each. */
int mmi_conv_bn_fwd2(const float* const* x, const float* const* w, float* const* y, float* const* stat_partials,
                     const mmi_bn_stats* bn, int mi_stride, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d,
                     void* stream);
/* dx[g] = conv_transpose(dy[g], w[g]) [+ skip[g]]; skip (row stride ldskip; may alias dx) only for 1x1 stride-1 layers, else NULL */
int mmi_conv_dgrad2(const float* const* dy, const float* const* w, float* const* dx, const float* const* skip, int ldskip,
                    void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream);
/* Input gradient with the BatchNorm backward REDUCTION of the layer below in its epilogue (round 4).  dx of a convolution is the
 * incoming gradient of the BatchNorm + activation whose output the convolution read (models/common.py:108-125: Conv = conv -> bn ->
 * act; when that output has no other reader).  The epilogue holds the finished dx tile, so it forms dz = dx * act'(xhat * gamma + beta)
 * and writes the partial column sums of dz and dz * xhat per row block: hook->partials[row_block][2][Cin], row blocks =
 * mmi_conv_dgrad_row_blocks_n(d, nprob) -- the list mmi_bn_act_bwd_apply / _apply_map fold.  That BatchNorm's own backward then skips
 * its pass over (dx, y): one read of each per layer and one launch less.  hook->y: the BatchNorm's input (the layer's raw conv output,
 * row stride ldy), mean at mean_invstd[c], 1/std at mean_invstd[mi_stride + c].  Stride-1 MFMA layers only; skip / ldskip as
 * mmi_conv_dgrad2 (dx = dgrad + skip, 1x1 stride-1 layers; NULL otherwise).  fp32 storage. */
typedef struct mmi_bn_reduce_hook {
  const float* y;
  int ldy;
  const float* mean_invstd;
  int mi_stride;
  const float* gamma;
  const float* beta;
  int act;
  float* partials;
} mmi_bn_reduce_hook;
int mmi_conv_dgrad_row_blocks_n(const mmi_conv_desc* d, int nprob);
int mmi_conv_dgrad_bnred(const float* dy, const float* w, float* dx, const float* skip, int ldskip, const mmi_bn_reduce_hook* hook,
                         void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream);
int mmi_conv_dgrad2_bnred(const float* const* dy, const float* const* w, float* const* dx, const float* const* skip, int ldskip,
                          const mmi_bn_reduce_hook* hooks, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d,
                          void* stream);
/* The apply pass of a mapped BatchNorm backward whose reduction came from elsewhere: parameter block i's sums are folded from
 * partials[i][nparts][2][map->blk] into map->dgamma[i] / dbeta[i], then dy as mmi_bn_act_bwd_map writes it. */
int mmi_bn_act_bwd_apply_map(const float* y, int ldy, const float* dout, int ldd, const float* dout1, int ldd1, const float* mean_invstd,
                             const mmi_bn_map* map, const float* const* partials, int nparts, float* dy, int lddy, int64_t rows, int C,
                             int act, int frozen, void* stream);
/* dw[g] = dy[g]^T x[g] (and dbias[g] = column sums of dy[g] when dbias != NULL); `table` as for mmi_conv_wgrad_tab (shapes only:
 * one table serves both problems) or NULL */
int mmi_conv_wgrad2(const float* const* dy, const float* const* x, float* const* dw, float* const* dbias, void* workspace,
                    size_t workspace_bytes, const void* table, const mmi_conv_desc* d, void* stream);

/* ---- Contour Enhancement Module special forms (models/common.py:751-911) --------------------------------------------
 * The 3->24 and 24->3 convs of AdaptiveModule3 are served by mmi_conv_fwd/dgrad/wgrad themselves (direct VALU kernels,
 * selected by the descriptor).  The 24->24 EnhanceConv2d never runs as a convolution:
 *   t[pix][o] = r[pix][o] + factor[o] * stencil_{o%8}(sum_c r[pix][c]) + bias[o]      (common.py:795, 899-909)
 * chansum (N*H*W floats) is written by the forward and is all the backward needs besides dt. */
int mmi_sobel_add_fwd(const float* r, int ldr, const float* factor, const float* bias, float* chansum, float* t, int ldt,
                      int N, int H, int W, int C, void* stream);
size_t mmi_sobel_add_bwd_workspace(int N, int H, int W, int C);
int mmi_sobel_add_bwd(const float* dt, int ldd, const float* chansum, const float* factor, float* dr, int lddr,
                      float* dfactor, float* dbias, void* workspace, int N, int H, int W, int C, void* stream);

/* out[c] = sum_r x[r,c] (bias gradients of Detect / Linear).  partials: workspace of mmi_bn_bwd_parts(rows)*C floats. */
int mmi_colsum(const float* x, int ldx, int64_t rows, int C, float* partials, float* out, void* stream);

/* ---- layout / elementwise / pooling ------------------------------------------------------------------------------*/
/* NCHW (arbitrary element strides sn,sc,sh,sw) -> NHWC contiguous.  train.py:743-745 hands the model strided views. */
int mmi_nchw_to_nhwc(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* y, int N, int C, int H,
                     int W, void* stream);
int mmi_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream);
/* Input wire format (utils/datasets.py:1565: uint8 (N,6,H,W), RGB = channels 0-2, IR = 3-5) -> the two fp32 NHWC
 * images x/255 that train.py:743-745 computes with .float()/255 and two channel slices. */
int mmi_u8_pair_to_nhwc(const uint8_t* in, float* rgb, float* ir, int N, int H, int W, void* stream);
/* Focus space-to-depth (models/common.py:708): in (N,H,W,C) -> out (N,H/2,W/2,4C), channel = q*C+c,
 * q: (dy,dx)=(0,0),(1,0),(0,1),(1,1).  inverse=1: in is a (N,H/2,W/2,4C) gradient, out the (N,H,W,C) gradient. */
int mmi_space_to_depth(const float* in, float* out, int N, int H, int W, int C, int inverse, void* stream);
/* the same with a row stride on the depth-side tensor (a channel block of a wider buffer: the twin-lane Focus input) */
int mmi_space_to_depth_ld(const float* in, float* out, int N, int H, int W, int C, int ldd, int inverse, void* stream);
/* Detect view/permute (models/yolo_test.py:54-55): in (B,P=ny*nx,na*no) -> out (B,na,P,no); inverse=1 maps a
 * (B,na,P,no) gradient back to (B,P,na*no). */
int mmi_head_permute(const float* in, float* out, int B, int na, int no, int P, int inverse, void* stream);
/* out[r, :C] = a[r, :C] + b[r, :C] with row strides (Add/Add2: common.py:914-935) */
int mmi_add(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int64_t rows, int C, void* stream);
/* strided 2-D copy out[r,:C] = in[r,:C] (Concat common.py:740-748 and its backward split) */
int mmi_copy2d(const float* in, int ldi, float* out, int ldo, int64_t rows, int C, void* stream);
/* nearest x2 upsample (nn.Upsample in the YAML head) and its backward (sum of the 4 children) */
int mmi_upsample2x(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* the same on rows with strides: x and / or y may be channel slices of wider buffers (the neck's concat buffers) */
int mmi_upsample2x_ld(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, void* stream);
int mmi_upsample2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* the same with dy read through a row stride (a channel slice of a Concat gradient) and, when skip != NULL, the gradient of
 * the map's OTHER consumer added in the same pass: dx = sum4(dy) + skip.  Replaces the autograd engine's fan-out accumulation
 * (an ATen add) where a head Conv output feeds both nn.Upsample and a later Concat (yolov5 PANet, YAML head). */
int mmi_upsample2x_bwd_acc(const float* dy, int lddy, const float* skip, int ldskip, float* dx, int N, int H, int W, int C,
                           void* stream);
/* SPP (models/common.py:681-693): x (N,H,W,C) row stride ldx -> out[..,0:C]=x, [C:2C]=mp5, [2C:3C]=mp9, [3C:4C]=mp13
 * (row stride ldo >= 4C), computed as cascaded 5x5 stride-1 max-pools. */
int mmi_spp_pool_fwd(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int C, void* stream);
/* gradient w.r.t. x given dcat (N,H,W,4C): each pooled gradient is routed to the arg-max position (first max in
 * row-major window order, as ATen max_pool2d), added to dcat[...,0:C]. */
int mmi_spp_pool_bwd(const float* x, int ldx, const float* dcat, int ldd, float* dx, int lddx, int N, int H, int W,
                     int C, void* stream);

/* ---- fusion transformers, token side (models/common.py:1147-1267, 476-503, 529) ---------------------------------- */
/* out = (a [+ b broadcast with period bmod]) * keep/(1-p); keep = hash(seed, index) >= p*2^32 (p = 0: plain add).
 * nn.Dropout at common.py:326,1173-1174,1258 and the positional-embedding add at common.py:529,1347.  The same call
 * with b=NULL applied to the incoming gradient is the backward. */
int mmi_dropout(const float* a, const float* b, int64_t bmod, float* out, int64_t n, float p, uint64_t seed,
                const uint64_t* seed_dev, void* stream);
/* Every dropout mask is hash(seed + (seed_dev ? *seed_dev : 0), element index): `seed` is a per-call-site constant, the
 * device word is advanced once per training step, so a captured hipGraph draws fresh masks on every replay. */
int mmi_seed_advance(uint64_t* seed_dev, void* stream);
int mmi_gelu_fwd(const float* x, float* y, int64_t n, void* stream);                       /* nn.GELU (erf), common.py:1256 */
int mmi_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
int mmi_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream);                    /* common.py:335,477,480 */
int mmi_sigmoid_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);
int mmi_mul(const float* a, const float* b, float* out, int64_t n, void* stream);          /* common.py:502-503 */
int mmi_scale(const float* a, const float* scalar_dev, float* out, int64_t n, void* stream); /* out = a * scalar_dev[0] */
/* nn.LayerNorm over the last dim (common.py:1250-1251, 323, 1294); stats[rows][2] = (mean, rstd) */
int mmi_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, int rows, int C,
                      float eps, void* stream);
int mmi_layernorm_bwd_parts(int rows);
int mmi_layernorm_bwd(const float* x, const float* gamma, const float* stats, const float* dy, float* dx, float* partials,
                      float* dgamma, float* dbeta, int rows, int C, void* stream); /* partials: parts*2*C floats */
/* The two halves of mmi_layernorm_bwd for a transformer block's backward (common.py:1263-1266: x = x + sa(ln(x)),
 * x = x + mlp(ln(x))).  _input sits on the dependency chain: dx = LayerNorm'(dy) [+ dresidual, the gradient arriving over
 * the skip connection], and optionally dx_dropped = dx * dropout-mask/(1-p) of the branch that produced this block input
 * (what the Dropout at common.py:1174 / 1258 hands to its Linear in the backward).  _params (dgamma, dbeta) is off the
 * chain and may run on another stream. */
int mmi_layernorm_bwd_input(const float* x, const float* gamma, const float* stats, const float* dy, const float* dresidual,
                            float* dx, float* dx_dropped, float p_drop, uint64_t seed, const uint64_t* seed_dev, int rows,
                            int C, void* stream);
int mmi_layernorm_bwd_params(const float* x, const float* stats, const float* dy, float* partials, float* dgamma,
                             float* dbeta, int rows, int C, void* stream);
/* SelfAttention core over T=128 tokens (common.py:1206-1231): q,k,v,out are (B,128,heads*dk) with row stride ld; head h
 * owns channels [h*dk,(h+1)*dk).  probs (B,heads,128,128) receives the softmax (saved for backward); attention dropout
 * is regenerated from (seed) in the backward. */
int mmi_attention_fwd(const float* q, const float* k, const float* v, float* out, float* probs, int B, int heads, int dk,
                      int ld, float p_drop, uint64_t seed, const uint64_t* seed_dev, void* stream);
int mmi_attention_bwd(const float* q, const float* k, const float* v, const float* probs, const float* dout, float* dq,
                      float* dk_, float* dv, int B, int heads, int dk, int ld, float p_drop, uint64_t seed,
                      const uint64_t* seed_dev, void* stream);
/* The same with two row strides: ld for q, k, v (and dq, dk, dv), ldo for out (and dout).  q, k, v may then be the three
 * column blocks of one (B*128, 3*heads*dk) buffer written by a single packed projection GEMM. */
int mmi_attention_fwd_strided(const float* q, const float* k, const float* v, float* out, float* probs, int B, int heads,
                              int dk, int ld, int ldo, float p_drop, uint64_t seed, const uint64_t* seed_dev, void* stream);
int mmi_attention_bwd_strided(const float* q, const float* k, const float* v, const float* probs, const float* dout,
                              float* dq, float* dk_, float* dv, int B, int heads, int dk, int ld, int ldo, float p_drop,
                              uint64_t seed, const uint64_t* seed_dev, void* stream);

/* ---- fusion stack, spatial side ------------------------------------------------------------------------------------ */
/* nn.AdaptiveAvgPool2d((8,8)) (common.py:395-396, 1331-1332) written straight into the token layout:
 * out[n*out_batch_stride + (oi*8+oj)*out_ld + c]. */
int mmi_avgpool8_fwd(const float* x, int ldx, int N, int H, int W, int C, float* out, int64_t out_batch_stride, int out_ld,
                     void* stream);
int mmi_avgpool8_bwd(const float* dpool, int64_t d_batch_stride, int d_ld, float* dx, int lddx, int N, int H, int W, int C,
                     void* stream);
/* the same plus the gradient of the map's other consumer (Add2, models/common.py:924-935) when skip != NULL: dx = pool
 * gradient + skip in one pass instead of a pool-gradient pass and an ATen accumulation. */
int mmi_avgpool8_bwd_acc(const float* dpool, int64_t d_batch_stride, int d_ld, const float* skip, int ldskip, float* dx, int lddx,
                         int N, int H, int W, int C, void* stream);
/* out = x + F.interpolate(tok 8x8 -> HxW, bilinear, align_corners=False) (common.py:548-550, 1364-1366 fused with Add2
 * common.py:924-935); x may be NULL.  tok[n*tok_batch_stride + (i*8+j)*tok_ld + c]. */
int mmi_upsample_add_fwd(const float* x, int ldx, const float* tok, int64_t tok_batch_stride, int tok_ld, float* out,
                         int ldo, int N, int H, int W, int C, void* stream);
int mmi_upsample_add_bwd(const float* dout, int ldd, float* dtok, int64_t tok_batch_stride, int tok_ld, int N, int H, int W,
                         int C, void* stream);
/* extract_frequency2 (common.py:37-69) on pooled (B,64,C) token-layout maps + torch.mul(high, pooled) (common.py:440-441):
 * 8x8 DFT per (b,c) plane, keep the unshifted bins whose bit (u*8+v) is set in keep_mask, inverse, real part -> fp16. */
int mmi_ffm_highpass(const float* pooled, float* out, int B, int C, uint64_t keep_mask, void* stream);
/* Seperation_loss (common.py:128-139, 487-494) over four (B,64,8) gate maps -> out[0] */
int mmi_separation_loss(const float* m_rgb, const float* m_ir, const float* m_rgb_hi, const float* m_ir_hi, int B, float* out,
                        void* stream);
/* One pass over the FFM inputs (N,H,W,C): out3 = (SSIMloss, Entropy_loss, ContrastiveValue) of models/yolo_test.py:338-486;
 * the fused map (mean of the two up-sampled FFM outputs) is recomputed from tokens (N,128,C). */
size_t mmi_fusion_stats_workspace(void);
int mmi_fusion_stats(const float* in_rgb, int lda, const float* in_ir, int ldb, const float* tokens, int N, int H, int W,
                     int C, void* workspace, float* out3, void* stream);

/* ---- detection-loss target assignment (integer kernel, bit-exact) ----------------------------------------------
 * Replaces ComputeLoss.build_targets, utils/loss.py:189-245.  targets: (nt,6) fp32 [img,cls,x,y,w,h]; anchors (nl,na,2)
 * in grid units; grids[nl][2] = (ny,nx) int32 on the DEVICE.  Outputs per level l (capacity cap = 5*na*nt records each):
 * idx[l][4][cap] int64 rows (b,a,gj,gi), tcls[l][cap] int64, tbox[l][cap][4] fp32, anch[l][cap][2] fp32, counts[l] int32.
 * Record order equals the reference's boolean-mask order (offset-major, then anchor-major, then target order). */
int mmi_build_targets(const float* targets, int nt, const float* anchors, int nl, int na, const int32_t* grids_dev,
                      float anchor_t, int64_t* idx, int64_t* tcls, float* tbox, float* anch, int32_t* counts,
                      void* stream);

/* ---- detection loss: value and gradient w.r.t. the head outputs in one pass ---------------------------------------
 * Replaces ComputeLoss.__call__ (utils/loss.py:113-184) + bbox_iou CIoU (utils/general.py:403-447).
 * preds/dpreds: HOST arrays of nl device pointers to (bs,na,ny,nx,nc+5) tensors; grids_host[nl][2]=(ny,nx) and
 * balance_host[nl] are host arrays; idx/tcls/tbox/anch/counts/cap are mmi_build_targets outputs (device).  combine:
 * device vector of ncombine CombineLoss values (may be NULL when ncombine=0).  out5 = (loss*bs, lbox, lobj, lcls,
 * Detectloss).  dpreds receive d(out5[0])/d(preds).  flag: bit 0 = add the CombineLoss term (the reference's `Flag`); bit 1 = the
 * tensors are the head convolutions' NHWC outputs (bs,ny,nx,na*(nc+5)), i.e. Detect's (bs,na,ny,nx,nc+5) result as a strided
 * view of them (models/yolo_test.py:54-55 without the permute copy); dpreds take the same layout. */
size_t mmi_detect_loss_workspace(int nl, int64_t total_cells, int64_t cap);
int mmi_detect_loss(const float* const* preds, float* const* dpreds, const int32_t* grids_host, int nl, int bs, int na,
                    int nc, const int64_t* idx, const int64_t* tcls, const float* tbox, const float* anch,
                    const int32_t* counts, int64_t cap, const float* balance_host, float hbox, float hobj, float hcls,
                    float gr, float cp, float cn, const float* combine, int ncombine, float alpha, int flag,
                    void* workspace, size_t workspace_bytes, float* out5, void* stream);

/* ---- fused multi-tensor SGD(nesterov) + ModelEMA -------------------------------------------------
 * Replaces optimizer.step() + ema.update() of train.py:799-804 (torch.optim.SGD nesterov over the 3 groups of
 * train.py:572-589; utils/torch_utils.py:269-299).  recs_dev: array of 48-byte records
 * {float* p; const float* g; float* buf; float* ema; int64 n; int32 group; int32 flags(1=SGD,2=EMA,4=16B-aligned)};
 * chunks_dev: array of {int32 rec; int32 chunk} covering every tensor in MMI_OPT_CHUNK-element pieces;
 * hyper_dev: 9 floats = lr[3], weight_decay[3], momentum, ema_decay, first_step_flag (all read on the device, so the
 * launch can live in a captured graph while the schedule changes). */
#define MMI_OPT_CHUNK 65536
int mmi_sgd_ema_step(const void* recs_dev, const void* chunks_dev, int nchunks, const float* hyper_dev, void* stream);

/* ---- evaluation path ------------------------------------------------------------------------
 * Detect in eval mode (models/yolo_test.py:57-66): x is one level's permuted head output (B,na,ny,nx,no); writes the
 * decoded rows [sigmoid -> (xy*2-0.5+grid)*stride, (wh*2)^2*anchor, conf...] into z (B,total_rows,no) at row_offset
 * (levels are concatenated along rows, yolo_test.py:68).  anchor_px: the level's na x 2 anchors in pixels (anchor_grid). */
int mmi_detect_decode(const float* x, float* z, int B, int na, int ny, int nx, int no, int64_t total_rows,
                      int64_t row_offset, float stride, const float* anchor_px, void* stream);
/* non_max_suppression (utils/general.py:486-580; NMS proper = torchvision.ops.nms semantics): pred (B,R,nc+5) decoded
 * rows; class_allow: device array of nc bytes, non-zero = class kept (general.py:549-550 `classes`), NULL = no filter;
 * max_wh 4096 is the
 * reference's class offset; out (B,max_det,6) = [x1,y1,x2,y2,conf,cls] by decreasing confidence, nout[B] rows valid.
 * Every candidate takes part (the reference first cuts to its 30 000 best). */
size_t mmi_nms_workspace(int B, int64_t R, int nc, int multi_label);
int mmi_nms(const float* pred, int B, int64_t R, int nc, float conf_thres, float iou_thres, const uint8_t* class_allow, int agnostic,
            int multi_label, int max_det, float max_wh, void* workspace, size_t workspace_bytes, float* out, int* nout,
            void* stream);

/* ---- fused Contour Enhancement forward (models/common.py:751-803 AdaptiveModule3 + 806-911
 * EnhanceConv2d with the reference's frozen stencil bank): x (N,H,W,3) -> y3 = conv3(t), t = r + factor * stencil(sum_c r_c) +
 * bias, r = LeakyReLU(BN2(conv2(x))), one 16x16 tile per workgroup with r and t in LDS (x staged with a halo of 3).  BN2's
 * batch statistics come from a recompute pre-pass (mmi_cem_conv2_stats -> partials[mmi_cem_blocks][2][24] -> mmi_bn_finalize);
 * y3's statistics partials [mmi_cem_blocks][2][3] come out of the fused kernel, the output of the module is then
 * mmi_bn_act_fwd(y3, ..., residual = x).  y2, t (24 channels) and the channel-sum map are written for the backward when
 * non-NULL (training); inference passes NULL and moves 1 read of x + 1 write of y3.  w2 = [24][9][3], w3 = [3][9][24] (OHWI). */
/* Middle of the CEM backward in one kernel: dy3 (N,H,W,3; gradient of conv3's output) -> dt = conv3^T(dy3) -> stencil-bank backward ->
 * dr (N,H,W,24; gradient of r), dfactor[24], dbias[24] (EnhanceConv2d's parameters), with dt and the eight D maps in LDS.
 * Replaces mmi_conv_dgrad(conv3) + mmi_sobel_add_bwd.  w3 = [3][9][24] (OHWI), chansum = the forward's channel-sum map. */
size_t mmi_cem_bwd_mid_workspace(int N, int H, int W);
/* y2 .. bn_partials all NULL, or all set: then BatchNorm2's backward REDUCTION rides along -- with y2 (conv2's raw output) and BN2's
 * mean_invstd / gamma / beta the kernel also sums dz = dr * LeakyReLU'(z) and dz * xhat per channel into
 * bn_partials[mmi_cem_bwd_mid_blocks(N,H,W)][2][24], the list mmi_bn_act_bwd_apply(y2, dr, ..., bn_partials, blocks, ...) folds. */
int mmi_cem_bwd_mid_blocks(int N, int H, int W);
int mmi_cem_bwd_mid(const float* dy3, const float* w3, const float* chansum, const float* factor, float* dr, float* dfactor,
                    float* dbias, void* workspace, const float* y2, const float* mean_invstd2, const float* gamma2,
                    const float* beta2, float* bn_partials, int N, int H, int W, void* stream);
int mmi_cem_blocks(int N, int H, int W);
int mmi_cem_conv2_stats(const float* x, int ldx, const float* w2, float* stat_partials, int N, int H, int W, void* stream);
/* Training forward in two launches (round 4; conv2 is evaluated once instead of 2.56 times per pixel): y2 = conv2(x) stored, with
 * BN2's statistics partials [mmi_cem_conv2_fwd_blocks][2][24] -> mmi_bn_finalize; then y2 -> r -> t -> y3 (mmi_cem_fwd_from_y2:
 * the fused kernel reading y2 instead of recomputing it; t / chansum written when non-NULL, y3 statistics partials
 * [mmi_cem_blocks][2][3]).  y2, and for the same BN2 statistics t / chansum / y3 and its partials, are bit-identical to
 * mmi_cem_fused_fwd's; the BN2 statistics themselves are sums of differently grouped partials (equal to rounding). */
/* conv2's weight gradient with BatchNorm2 + LeakyReLU's backward applied in the loader (the image takes no gradient, so dy2 has no
 * other reader): dw2 = wgrad(x, dy2(dr, y2)), dy2 never in HBM.  dgamma2 / dbeta2: finished by mmi_bn_act_bwd(..., dy = NULL, ...)
 * earlier on the same stream.  Workspace: mmi_cem_conv2_wgrad_bn_workspace bytes (plain scratch, no counters). */
size_t mmi_cem_conv2_wgrad_bn_workspace(int N, int H, int W);
int mmi_cem_conv2_wgrad_bn(const float* dr, const float* y2, const float* x, int ldx, const float* mean_invstd2, const float* gamma2,
                           const float* beta2, const float* dgamma2, const float* dbeta2, int frozen, float* dw2, void* workspace,
                           size_t workspace_bytes, int N, int H, int W, void* stream);
int mmi_cem_conv2_fwd_blocks(int N, int H, int W);
int mmi_cem_conv2_fwd(const float* x, int ldx, const float* w2, float* y2, float* stat_partials, int N, int H, int W, void* stream);
int mmi_cem_fwd_from_y2(const float* y2, const float* mean_invstd2, const float* gamma2, const float* beta2, const float* factor,
                        const float* sobel_bias, const float* w3, float* t, float* chansum, float* y3, float* stat_partials3, int N,
                        int H, int W, void* stream);
int mmi_cem_fused_fwd(const float* x, int ldx, const float* w2, const float* mean_invstd2, const float* gamma2, const float* beta2,
                      const float* factor, const float* sobel_bias, const float* w3, float* y2, float* t, float* chansum, float* y3,
                      float* stat_partials3, int N, int H, int W, void* stream);

/* ---- bf16 storage (the reference trains under torch.cuda.amp, train.py:706,784,796-801) ---------
 * Opt-in second numeric mode of the conv family and its glue: activations and activation gradients are bf16 in HBM (NHWC,
 * same row-stride convention, counted in ELEMENTS), weights / weight gradients / BatchNorm parameters and statistics stay
 * fp32; every product is one bf16 MFMA with fp32 accumulation, statistics are taken from the fp32 accumulators.  bf16 has
 * fp32's exponent range, so no loss scaling is involved (GradScaler exists for fp16).  Same semantics and workspace rules
 * as the fp32 entry points of the same name; channel counts and row strides must be multiples of 4. */
int mmi_conv_fwd_row_blocks_bf16(const mmi_conv_desc* d);
size_t mmi_conv_fwd_workspace_bf16(const mmi_conv_desc* d);
/* bn != NULL: training Conv, statistics finished in the launch (as mmi_conv_bn_fwd); bias != NULL: Detect-style bias */
int mmi_conv_fwd_bf16(const void* x, const float* w, const float* bias, void* y, float* stat_partials, const mmi_bn_stats* bn,
                      void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream);
/* dx = conv_transpose(dy, w) [+ skip] (skip: 1x1 stride-1 layers, the Bottleneck shortcut gradient) */
int mmi_conv_dgrad_bf16(const void* dy, const float* w, void* dx, const void* skip, int ldskip, const mmi_conv_desc* d, void* stream);
/* dw, dbias fp32; workspace as mmi_conv_wgrad */
int mmi_conv_wgrad_bf16(const void* dy, const void* x, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                        const mmi_conv_desc* d, void* stream);
int mmi_bn_act_fwd_split_bf16(const void* y, int ldy, const float* mean_invstd, const float* gamma, const float* beta,
                              const void* residual, int ldr, void* out, int ldo, void* out1, int ldo1, int split, int64_t rows,
                              int C, int act, void* stream);
int mmi_bn_act_bwd_bf16(const void* y, int ldy, const void* dout, int ldd, const void* dout1, int ldd1, int split,
                        const float* mean_invstd, const float* gamma, const float* beta, void* workspace, size_t workspace_bytes,
                        void* dy, int lddy, float* dgamma, float* dbeta, float* dgamma1, float* dbeta1, int64_t rows, int C,
                        int act, int frozen, void* stream);
int mmi_cast_f32_bf16(const float* in, int ldi, void* out, int ldo, int64_t rows, int C, void* stream);   /* round to nearest even */
int mmi_cast_bf16_f32(const void* in, int ldi, float* out, int ldo, int64_t rows, int C, void* stream);
int mmi_add_bf16(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int64_t rows, int C, void* stream);
int mmi_copy2d_bf16(const void* in, int ldi, void* out, int ldo, int64_t rows, int C, void* stream);
int mmi_upsample2x_bf16(const void* x, void* y, int N, int H, int W, int C, void* stream);
int mmi_upsample2x_bwd_bf16(const void* dy, void* dx, int N, int H, int W, int C, void* stream);

/* ---- data-parallel communication (one process per GPU, RCCL over xGMI) ---------------------------------------------
 * Replaces DistributedDataParallel's gradient all-reduce and initial parameter broadcast (train.py:683-686 of the
 * reference; SURVEY.md §8e: pure data parallelism, one all-reduce of all trainable gradients per optimizer step).  RCCL is
 * bound at run time and the copy already loaded by PyTorch-ROCm is reused; without RCCL these calls fail with a message
 * and everything else still works.  One communicator per process.  The collectives are only enqueued on the given stream
 * (no watchdog thread, no host synchronisation), so they overlap with backward on a side stream and may be captured into a
 * hipGraph with the kernels around them. */
int mmi_comm_available(void);                       /* 1 when RCCL could be bound */
int mmi_comm_unique_id(void* id128_host);           /* rank 0: 128 bytes of HOST memory to hand to every rank */
int mmi_comm_init(int rank, int world, const void* id128_host);   /* collective; after hipSetDevice() */
int mmi_comm_world(void);
int mmi_comm_rank(void);
int mmi_allreduce_bucket(float* bucket, int64_t count, int average, void* stream);   /* in place; average: mean over ranks */
int mmi_broadcast_bytes(void* buf, int64_t bytes, int root, void* stream);           /* in place */
int mmi_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif
