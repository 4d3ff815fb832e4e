// igemm_kernel instantiations: DGRAD = true, EPI = false (igemm_launch.h)
#include "igemm_launch.h"

namespace mmi_ig {
template int launch_igemm<true, false>(const IgemmP&, const FwdPlan&, bool, void*, size_t, hipStream_t, size_t, const IgemmP*);
template int launch_igemm_bf16<true, false>(IgemmP, const FwdPlan&, hipStream_t, const char*);
template int sk_occupancy<true>(int);
}  // namespace mmi_ig
