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
 * (BatchNorm batch statistics, models/common.py:116) to stat_partials[rb][2][Cout], rb < mmi_conv_fwd_row_blocks(). */
int mmi_conv_fwd_row_blocks(const mmi_conv_desc* d);
int mmi_conv_fwd(const float* x, const float* w, const float* bias, float* y, float* stat_partials,
                 const mmi_conv_desc* d, void* stream);
/* dx = conv_transpose(dy, w): gradient w.r.t. the input (autograd of the call sites above). dx has row stride ldx. */
int mmi_conv_dgrad(const float* dy, const float* w, float* dx, const mmi_conv_desc* d, void* stream);
/* dw (OHWI) = sum over pixels dy^T x.  workspace holds split-K slabs; query its size first. */
size_t mmi_conv_wgrad_workspace(const mmi_conv_desc* d);
int mmi_conv_wgrad(const float* dy, const float* x, float* dw, void* workspace, size_t workspace_bytes,
                   const mmi_conv_desc* d, void* stream);

/* ---- BatchNorm (training statistics) + activation (+ residual) --------------------------------------------------
 * Replaces nn.BatchNorm2d + nn.SiLU / nn.LeakyReLU at models/common.py:116-122, 766-767, 774-775 and the residual adds
 * at common.py:613, 799.  eps/momentum per utils/torch_utils.py:149-151. */
/* partials[rb][2][C] -> mean_invstd[2][C]; updates running_mean/var (unbiased var) and increments
 * *num_batches_tracked (int64) when non-NULL. */
int mmi_bn_finalize(const float* partials, int nparts, int64_t rows, int C, float eps, float momentum,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean_invstd,
                    void* stream);
/* eval mode: mean_invstd from running stats. */
int mmi_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean_invstd,
                      void* stream);
/* out = act(gamma*(y-mean)*invstd+beta) [+ residual] */
int mmi_bn_act_fwd(const float* y, int ldy, const float* mean_invstd, const float* gamma, const float* beta,
                   const float* residual, int ldr, float* out, int ldo, int64_t rows, int C, int act, void* stream);
/* backward, pass 1: partial column sums of dz and dz*xhat, dz = dout*act'(z): partials[pb][2][C], pb < nparts
 * (nparts = mmi_bn_bwd_parts(rows)). */
int mmi_bn_bwd_parts(int64_t rows);
int mmi_bn_act_bwd_reduce(const float* y, int ldy, const float* dout, int ldd, const float* mean_invstd,
                          const float* gamma, const float* beta, float* partials, int64_t rows, int C, int act,
                          void* stream);
/* backward, pass 2: sums partials -> dgamma,dbeta (C each) and writes dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)).
 * In eval-stat mode (frozen=1) dy = gamma*invstd*dz. */
int mmi_bn_act_bwd_apply(const float* y, int ldy, const float* dout, int ldd, const float* mean_invstd,
                         const float* gamma, const float* beta, const float* partials, int nparts, float* dy, int lddy,
                         float* dgamma, float* dbeta, int64_t rows, int C, int act, int frozen, void* stream);

/* out[c] = sum_r x[r,c] (bias gradients of Detect / Linear).  partials: workspace of mmi_bn_bwd_parts(rows)*C floats. */
int mmi_colsum(const float* x, int ldx, int64_t rows, int C, float* partials, float* out, void* stream);

/* ---- layout / elementwise / pooling ------------------------------------------------------------------------------*/
/* NCHW (arbitrary element strides sn,sc,sh,sw) -> NHWC contiguous.  train.py:743-745 hands the model strided views. */
int mmi_nchw_to_nhwc(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* y, int N, int C, int H,
                     int W, void* stream);
int mmi_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream);
/* Focus space-to-depth (models/common.py:708): in (N,H,W,C) -> out (N,H/2,W/2,4C), channel = q*C+c,
 * q: (dy,dx)=(0,0),(1,0),(0,1),(1,1).  inverse=1: in is a (N,H/2,W/2,4C) gradient, out the (N,H,W,C) gradient. */
int mmi_space_to_depth(const float* in, float* out, int N, int H, int W, int C, int inverse, void* stream);
/* Detect view/permute (models/yolo_test.py:54-55): in (B,P=ny*nx,na*no) -> out (B,na,P,no); inverse=1 maps a
 * (B,na,P,no) gradient back to (B,P,na*no). */
int mmi_head_permute(const float* in, float* out, int B, int na, int no, int P, int inverse, void* stream);
/* out[r, :C] = a[r, :C] + b[r, :C] with row strides (Add/Add2: common.py:914-935) */
int mmi_add(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int64_t rows, int C, void* stream);
/* strided 2-D copy out[r,:C] = in[r,:C] (Concat common.py:740-748 and its backward split) */
int mmi_copy2d(const float* in, int ldi, float* out, int ldo, int64_t rows, int C, void* stream);
/* nearest x2 upsample (nn.Upsample in the YAML head) and its backward (sum of the 4 children) */
int mmi_upsample2x(const float* x, float* y, int N, int H, int W, int C, void* stream);
int mmi_upsample2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* SPP (models/common.py:681-693): x (N,H,W,C) row stride ldx -> out[..,0:C]=x, [C:2C]=mp5, [2C:3C]=mp9, [3C:4C]=mp13
 * (row stride ldo >= 4C), computed as cascaded 5x5 stride-1 max-pools. */
int mmi_spp_pool_fwd(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int C, void* stream);
/* gradient w.r.t. x given dcat (N,H,W,4C): each pooled gradient is routed to the arg-max position (first max in
 * row-major window order, as ATen max_pool2d), added to dcat[...,0:C]. */
int mmi_spp_pool_bwd(const float* x, int ldx, const float* dcat, int ldd, float* dx, int lddx, int N, int H, int W,
                     int C, void* stream);

/* ---- detection-loss target assignment (integer kernel, bit-exact) ----------------------------------------------
 * Replaces ComputeLoss.build_targets, utils/loss.py:189-245.  targets: (nt,6) fp32 [img,cls,x,y,w,h]; anchors (nl,na,2)
 * in grid units; grids[nl][2] = (ny,nx) int32 on the DEVICE.  Outputs per level l (capacity cap = 5*na*nt records each):
 * idx[l][4][cap] int64 rows (b,a,gj,gi), tcls[l][cap] int64, tbox[l][cap][4] fp32, anch[l][cap][2] fp32, counts[l] int32.
 * Record order equals the reference's boolean-mask order (offset-major, then anchor-major, then target order). */
int mmi_build_targets(const float* targets, int nt, const float* anchors, int nl, int na, const int32_t* grids_dev,
                      float anchor_t, int64_t* idx, int64_t* tcls, float* tbox, float* anch, int32_t* counts,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif
