// Launchers of igemm_kernel: one explicit instantiation per (DGRAD, EPI) pair and translation unit.
#pragma once
#include "igemm_kernel.h"

namespace mmi_ig {

// Resident workgroups per CU of a stream-K kernel variant (registers and LDS decide; 3 by the launch bound).
template <bool DGRAD>
int sk_occupancy(int bn) {
  static int cache[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  int& c = cache[g_gemm_prec][bn == 128 ? 1 : 0];
  if (c == 0) {
    int n = 0;
    hipError_t e;
#define OCC(P_) (bn == 128 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 128, DGRAD, true, true, P_>, 256, 0) \
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 64, DGRAD, true, true, P_>, 256, 0))
    // (fp32: the uniform-tap variant is what nearly every stream-K shape runs; the few others fit its grid as well)
    if (g_gemm_prec == 2 && g_uniform_loaders)
      e = bn == 128 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 128, DGRAD, true, true, 2, false, true>, 256, 0)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 64, DGRAD, true, true, 2, false, true>, 256, 0);
    else if (g_gemm_prec == 3 && g_uniform_loaders)
      e = bn == 128 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 128, DGRAD, true, true, 3, false, true>, 256, 0)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 64, DGRAD, true, true, 3, false, true>, 256, 0);
    else if (g_gemm_prec == 0 && g_uniform_loaders)
      e = bn == 128 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 128, DGRAD, true, true, 0, false, true>, 256, 0)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 64, DGRAD, true, true, 0, false, true>, 256, 0);
    else
      e = g_gemm_prec == 0 ? OCC(0) : (g_gemm_prec == 1 ? OCC(1) : (g_gemm_prec == 2 ? OCC(2) : OCC(3)));
#undef OCC
    c = (e == hipSuccess && n > 0) ? n : (g_gemm_prec >= 2 ? 2 : 3);
    (void)hipGetLastError();
  }
  return c;
}

template <bool DGRAD, bool EPI>
int launch_igemm(const IgemmP& p0, const FwdPlan& f, bool vec, void* workspace, size_t workspace_bytes, hipStream_t s,
                 size_t slot_offset, const IgemmP* twin0) {
  const char* who = DGRAD ? "mmi_conv_dgrad" : "mmi_conv_fwd";
  IgemmP p = p0;
  // T8 images announced for this launch (mmi_gemm_operands_t8): taken and cleared, whatever path the launch then takes
  IgemmP twin_copy;
  const IgemmP* twin = twin0;
  {
    const void** pend = t8_pending();
    p.A8 = pend[0]; p.B8 = pend[1];
    if (twin0 != nullptr) {
      twin_copy = *twin0;
      twin_copy.A8 = pend[2]; twin_copy.B8 = pend[3];
      twin = &twin_copy;
    }
    pend[0] = pend[1] = pend[2] = pend[3] = nullptr;
  }
  p.zero = zero_src();
  if (p.zero == nullptr) {
    mmi_set_error("igemm: cannot resolve the zero-source symbol");
    return MMI_ERR_LAUNCH;
  }
  p.mtiles = f.mtiles;
  p.ntiles = f.ntiles;
  if (p.mi_stride == 0) p.mi_stride = p.Ncol;
  // uniform-tap loaders (igemm_kernel<..., UNI>): whole slabs inside one tap, tap table in 32 bits, 31-bit byte offsets
  bool uni = false;
  // (fp32 MFMA and -- round 3 -- the six-product split, the mode reported beside the headline; the other split forms keep the
  //  cursor loaders: every (PREC, UNI) pair is another set of kernel variants to build)
  if (vec && g_uniform_loaders && (g_gemm_prec == 0 || g_gemm_prec == 2 || g_gemm_prec == 3) && p.Kc % BK == 0 && p.KH * p.KW <= 32 && !(DGRAD && p.stride == 2 && !p.par)) {
    const int64_t margin = ((int64_t)p.KH * p.Ws + p.KW) * p.lda;
    const int64_t npix = (int64_t)(p.M / ((int64_t)p.P * p.Q)) * p.Hs * p.Ws;
    const int64_t a_bytes = (margin + (npix - 1) * p.lda + p.Kc) * 4;
    const int64_t b_bytes = DGRAD ? (int64_t)p.Kc * p.ldb * 4 : (int64_t)p.Ncol * p.ldb * 4;
    if (a_bytes < (1LL << 31) && b_bytes < (1LL << 31)) {
      uni = true;
      p.a_bytes = (uint32_t)a_bytes;
      p.b_bytes = (uint32_t)b_bytes;
      const int64_t c_bytes = ((int64_t)(p.M - 1) * p.ldc + p.Ncol) * 4;
      p.c_bytes = c_bytes < (1LL << 31) ? (uint32_t)c_bytes : 0u;
    }
  }
  // pre-split operands (T8 images handed over with mmi_gemm_operands_t8): bit 0 = weights, bit 1 = activations as well.  The byte
  // extents of the buffer resources grow by 6 / 4; a shape whose extents no longer fit 31 bits keeps the in-kernel split.
  int t8 = 0;
  if (uni && !EPI && (g_gemm_prec == 2 || g_gemm_prec == 3) && p.B8 != nullptr && p.ldb % 8 == 0 && (!DGRAD || p.Ncol % 8 == 0) &&
      ((uintptr_t)p.B8 & 15) == 0 && (twin == nullptr || twin->B8 != nullptr)) {
    const int64_t b8 = (int64_t)p.b_bytes / 4 * 6;
    if (b8 < (1LL << 31)) {
      t8 = 1;
      if (p.A8 != nullptr && p.lda % 8 == 0 && ((uintptr_t)p.A8 & 15) == 0 && (twin == nullptr || twin->A8 != nullptr)) {
        const int64_t a8 = (int64_t)p.a_bytes / 4 * 6;
        if (a8 < (1LL << 31)) {
          t8 = 3;
          p.a_bytes = (uint32_t)a8;
        }
      }
      p.b_bytes = (uint32_t)b8;
    }
  }
  // twin launch: a second problem of the same shape (other operand pointers, its own workspace) on gridDim.z = 2
  const int nz = twin != nullptr ? 2 : 1;
  IgemmDelta q{};   // byte distances to the twin problem's operands (everything else is shared: same shape, same schedule)
  if (twin != nullptr) {
    const IgemmP& t = *twin;
    if (t.bias != nullptr || p.bias != nullptr || t.res != nullptr || p.res != nullptr || (t.bn_mi == nullptr) != (p.bn_mi == nullptr) ||
        (t.stat_part == nullptr) != (p.stat_part == nullptr) || (t.aux == nullptr) != (p.aux == nullptr) || t.epi != p.epi ||
        t.ldaux != p.ldaux || (t.mi_stride != 0 && t.mi_stride != p.mi_stride) || t.bn_nnbt != p.bn_nnbt ||
        (t.bn_rmean == nullptr) != (p.bn_rmean == nullptr)) {
      mmi_set_error("%s: the two problems of a twin launch must agree in everything but their operand addresses", who);
      return MMI_ERR_ARG;
    }
    q.A = ptr_delta(t.A, p.A); q.B = ptr_delta(t.B, p.B); q.C = ptr_delta(t.C, p.C);
    q.stat_part = ptr_delta(t.stat_part, p.stat_part);
    q.aux = ptr_delta(t.aux, p.aux); q.aux_out = ptr_delta(t.aux_out, p.aux_out);
    q.fold_part = ptr_delta(t.bn_fold.part, p.bn_fold.part); q.fold_l1 = ptr_delta(t.bn_fold.l1, p.bn_fold.l1);
    q.fold_cnt = ptr_delta(t.bn_fold.cnt, p.bn_fold.cnt);
    q.bn_mi = ptr_delta(t.bn_mi, p.bn_mi); q.bn_rmean = ptr_delta(t.bn_rmean, p.bn_rmean); q.bn_rvar = ptr_delta(t.bn_rvar, p.bn_rvar);
    q.bn_nbt = ptr_delta(t.bn_nbt, p.bn_nbt);
    q.sk_slots = ptr_delta(t.sk_slots, p.sk_slots); q.sk_count = ptr_delta(t.sk_count, p.sk_count);
    q.A8 = ptr_delta(t.A8, p.A8); q.B8 = ptr_delta(t.B8, p.B8);
    if ((t.bnr_y == nullptr) != (p.bnr_y == nullptr) || t.bnr_ldy != p.bnr_ldy || t.bnr_act != p.bnr_act) {
      mmi_set_error("%s: the two problems of a twin launch must agree on the BatchNorm reduction riding along", who);
      return MMI_ERR_ARG;
    }
    q.bnr_y = ptr_delta(t.bnr_y, p.bnr_y); q.bnr_g = ptr_delta(t.bnr_g, p.bnr_g); q.bnr_b = ptr_delta(t.bnr_b, p.bnr_b);
  }
  if (f.sk_grid > 0) {
    if (twin != nullptr) {   // (the caller laid out both problems' counters and slots: igemm.hip, "twin launches")
      if (p.sk_count == nullptr || p.sk_slots == nullptr || twin->sk_count == nullptr || twin->sk_slots == nullptr) {
        mmi_set_error("%s: a twin launch of a stream-K shape needs both problems' workspaces", who);
        return MMI_ERR_WORKSPACE;
      }
    } else if (workspace == nullptr || workspace_bytes < slot_offset + sk_slot_bytes(f) || ((uintptr_t)workspace & 15)) {
      mmi_set_error("%s: this shape runs the stream-K schedule and needs a 16-byte aligned workspace of %zu bytes (got %zu)",
                    who, slot_offset + sk_slot_bytes(f), workspace_bytes);
      return MMI_ERR_WORKSPACE;
    }
    if (twin == nullptr) {
      p.sk_count = (int*)workspace;
      p.sk_slots = (float*)((char*)workspace + slot_offset);
    }
    const dim3 grid(f.sk_grid, 1, nz), block(256);
    if (g_pf2 && uni && !EPI && g_gemm_prec != 1 && (t8 == 0 || t8 == 3)) return launch_igemm_pf2<DGRAD>(p, q, f, grid, t8, s);
    if (t8) return launch_igemm_t8<DGRAD>(p, q, f, grid, t8, s);
    if (g_gemm_prec == 1) {
      if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 1, EPI>), grid, block, 0, s, p, q);
      else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 1, EPI>), grid, block, 0, s, p, q);
      MMI_CHECK_LAUNCH(who);
      return MMI_OK;
    }
    if (g_gemm_prec == 2) {
      if (uni) {
        if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 2, EPI, true>), grid, block, 0, s, p, q);
        else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 2, EPI, true>), grid, block, 0, s, p, q);
      } else if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 2, EPI>), grid, block, 0, s, p, q);
      else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 2, EPI>), grid, block, 0, s, p, q);
      MMI_CHECK_LAUNCH(who);
      return MMI_OK;
    }
    if (g_gemm_prec == 3) {
      if (uni) {
        if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 3, EPI, true>), grid, block, 0, s, p, q);
        else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 3, EPI, true>), grid, block, 0, s, p, q);
      } else if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 3, EPI>), grid, block, 0, s, p, q);
      else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 3, EPI>), grid, block, 0, s, p, q);
      MMI_CHECK_LAUNCH(who);
      return MMI_OK;
    }
    if (uni) {
      if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 0, EPI, true>), grid, block, 0, s, p, q);
      else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 0, EPI, true>), grid, block, 0, s, p, q);
    } else if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 0, EPI>), grid, block, 0, s, p, q);
    else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 0, EPI>), grid, block, 0, s, p, q);
    MMI_CHECK_LAUNCH(who);
    return MMI_OK;
  }
  const dim3 grid(f.mtiles * f.ntiles, p.par ? 4 : 1, nz), block(256);
  if (g_pf2 && uni && !EPI && g_gemm_prec != 1 && (t8 == 0 || t8 == 3) && !(DGRAD && f.bm == 128 && f.bn == 64 && p.Ncol <= 32))
    return launch_igemm_pf2<DGRAD>(p, q, f, grid, t8, s);
  if (t8) return launch_igemm_t8<DGRAD>(p, q, f, grid, t8, s);
#define LAUNCH(BM_, BN_, VEC_)                                                                      \
  hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, VEC_, false, 0, (VEC_) && EPI>), grid, block, 0, s, p, q)
  if (vec && (g_gemm_prec == 2 || g_gemm_prec == 3) && uni) {
#define LAUNCH_U2(BM_, BN_)                                                                                                    \
  do {                                                                                                                         \
    if (g_gemm_prec == 2) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 2, EPI, true>), grid, block, 0, s, p, q); \
    else hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 3, EPI, true>), grid, block, 0, s, p, q);                  \
  } while (0)
    if (f.bm == 128 && f.bn == 128) LAUNCH_U2(128, 128);
    else if (f.bm == 128 && f.bn == 64) LAUNCH_U2(128, 64);
    else LAUNCH_U2(64, 64);
#undef LAUNCH_U2
    MMI_CHECK_LAUNCH(who);
    return MMI_OK;
  }
  if (vec && g_gemm_prec >= 1) {
#define LAUNCH_B3(BM_, BN_)                                                                                          \
  do {                                                                                                               \
    if (g_gemm_prec == 1) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 1, EPI>), grid, block, 0, s, p, q); \
    else if (g_gemm_prec == 2) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 2, EPI>), grid, block, 0, s, p, q); \
    else if (g_gemm_prec == 5) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 5, EPI>), grid, block, 0, s, p, q); \
    else hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 3, EPI>), grid, block, 0, s, p, q);                  \
  } while (0)
    if (f.bm == 128 && f.bn == 128) LAUNCH_B3(128, 128);
    else if (f.bm == 128 && f.bn == 64) LAUNCH_B3(128, 64);
    else LAUNCH_B3(64, 64);
#undef LAUNCH_B3
    MMI_CHECK_LAUNCH(who);
    return MMI_OK;
  }
  if (!vec) {
    if (EPI) {
      mmi_set_error("%s: the fused Linear epilogues need channel counts and row strides that are multiples of 4", who);
      return MMI_ERR_ARG;
    }
    if (f.bm == 128 && f.bn == 64) LAUNCH(128, 64, false);
    else LAUNCH(64, 64, false);
  } else if (uni) {
#define LAUNCH_UNI(BM_, BN_) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 0, EPI, true>), grid, block, 0, s, p, q)
    if constexpr (DGRAD && !EPI) {
      static const bool w41_off = getenv("MMIDET_DGRAD_W41") != nullptr && atoi(getenv("MMIDET_DGRAD_W41")) == 0;   // (A/B switch)
      if (f.bm == 128 && f.bn == 64 && p.Ncol <= 32 && !w41_off) {
        hipLaunchKernelGGL((igemm_kernel<128, 64, true, true, false, 0, false, true, true>), grid, block, 0, s, p, q);
        MMI_CHECK_LAUNCH(who);
        return MMI_OK;
      }
    }
    if (f.bm == 128 && f.bn == 128) LAUNCH_UNI(128, 128);
    else if (f.bm == 128 && f.bn == 64) LAUNCH_UNI(128, 64);
    else LAUNCH_UNI(64, 64);
#undef LAUNCH_UNI
  } else if (f.bm == 128 && f.bn == 128) LAUNCH(128, 128, true);
  else if (f.bm == 128 && f.bn == 64) LAUNCH(128, 64, true);
  else LAUNCH(64, 64, true);
#undef LAUNCH
  MMI_CHECK_LAUNCH(who);
  return MMI_OK;
}

// ---- bf16 storage: activations and activation gradients are bf16 in HBM, weights / weight gradients /
// BatchNorm statistics fp32; one bf16 MFMA product per element pair with fp32 accumulation (igemm_kernel<..., PREC = 4>).
// One workgroup per tile (no stream-K: these launches are HBM-bound, not wave-quantisation-bound).
template <bool DGRAD, bool EPI>
int launch_igemm_bf16(IgemmP p, const FwdPlan& f, hipStream_t s, const char* who) {
  p.zero = zero_src();
  if (p.zero == nullptr) {
    mmi_set_error("%s: cannot resolve the zero-source symbol", who);
    return MMI_ERR_LAUNCH;
  }
  p.mtiles = f.mtiles;
  p.ntiles = f.ntiles;
  if (p.mi_stride == 0) p.mi_stride = p.Ncol;
  const IgemmDelta q{};   // (single problem)
  const dim3 grid(f.mtiles * f.ntiles, p.par ? 4 : 1), block(256);
  // uniform-tap loaders (round 3: also for the 2-byte activation operand; same conditions as launch_igemm)
  if (g_uniform_loaders && p.Kc % BK == 0 && p.KH * p.KW <= 32 && !(DGRAD && p.stride == 2 && !p.par)) {
    const int64_t margin = ((int64_t)p.KH * p.Ws + p.KW) * p.lda;
    const int64_t npix = (int64_t)(p.M / ((int64_t)p.P * p.Q)) * p.Hs * p.Ws;
    const int64_t a_bytes = (margin + (npix - 1) * p.lda + p.Kc) * 2;
    const int64_t b_bytes = DGRAD ? (int64_t)p.Kc * p.ldb * 4 : (int64_t)p.Ncol * p.ldb * 4;
    if (a_bytes < (1LL << 31) && b_bytes < (1LL << 31)) {
      p.a_bytes = (uint32_t)a_bytes;
      p.b_bytes = (uint32_t)b_bytes;
      const int64_t c_bytes = ((int64_t)(p.M - 1) * p.ldc + p.Ncol) * 2;
      p.c_bytes = c_bytes < (1LL << 31) ? (uint32_t)c_bytes : 0u;
      if (f.bm == 128 && f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, false, 4, EPI, true>), grid, block, 0, s, p, q);
      else if (f.bm == 128 && f.bn == 64) hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, false, 4, EPI, true>), grid, block, 0, s, p, q);
      else hipLaunchKernelGGL((igemm_kernel<64, 64, DGRAD, true, false, 4, EPI, true>), grid, block, 0, s, p, q);
      MMI_CHECK_LAUNCH(who);
      return MMI_OK;
    }
  }
  if (f.bm == 128 && f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, false, 4, EPI>), grid, block, 0, s, p, q);
  else if (f.bm == 128 && f.bn == 64) hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, false, 4, EPI>), grid, block, 0, s, p, q);
  else hipLaunchKernelGGL((igemm_kernel<64, 64, DGRAD, true, false, 4, EPI>), grid, block, 0, s, p, q);
  MMI_CHECK_LAUNCH(who);
  return MMI_OK;
}

}  // namespace mmi_ig
