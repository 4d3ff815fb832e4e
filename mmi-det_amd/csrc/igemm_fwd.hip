// igemm_kernel instantiations: DGRAD = false, EPI = false (igemm_launch.h)
#include "igemm_launch.h"

namespace mmi_ig {
template int launch_igemm<false, false>(const IgemmP&, const FwdPlan&, bool, void*, size_t, hipStream_t, size_t, const IgemmP*);
template int launch_igemm_bf16<false, false>(IgemmP, const FwdPlan&, hipStream_t, const char*);
template int sk_occupancy<false>(int);
}  // namespace mmi_ig
