// Launchers of the deep-prefetch (PF2) variants of igemm_kernel: fp32 MFMA and the three-term modes (split in the kernel or
// pre-split operands), uniform-tap loaders.
#pragma once
#include "igemm_kernel.h"

namespace mmi_ig {

template <bool DGRAD>
int launch_igemm_pf2(const IgemmP& p, const IgemmDelta& q, const FwdPlan& f, dim3 grid, int t8, hipStream_t s) {
  const char* who = DGRAD ? "mmi_conv_dgrad(pf2)" : "mmi_conv_fwd(pf2)";
  const dim3 block(256);
  const bool sk = f.sk_grid > 0;
#define L_(BM_, BN_, SK_, P_, T_) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, SK_, P_, false, true, false, T_, true>), grid, block, 0, s, p, q)
#define LT_(BM_, BN_, SK_)                                    \
  do {                                                        \
    if (g_gemm_prec == 0) L_(BM_, BN_, SK_, 0, 0);            \
    else if (g_gemm_prec == 3) {                              \
      if (t8 == 3) L_(BM_, BN_, SK_, 3, 3);                   \
      else L_(BM_, BN_, SK_, 3, 0);                           \
    } else {                                                  \
      if (t8 == 3) L_(BM_, BN_, SK_, 2, 3);                   \
      else L_(BM_, BN_, SK_, 2, 0);                           \
    }                                                         \
  } while (0)
  if (sk) {
    if (f.bn == 128) LT_(128, 128, true);
    else LT_(128, 64, true);
  } else if (f.bm == 128 && f.bn == 128) LT_(128, 128, false);
  else if (f.bm == 128 && f.bn == 64) LT_(128, 64, false);
  else LT_(64, 64, false);
#undef LT_
#undef L_
  MMI_CHECK_LAUNCH(who);
  return MMI_OK;
}

}  // namespace mmi_ig
