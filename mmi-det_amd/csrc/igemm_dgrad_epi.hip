// igemm_kernel instantiations: DGRAD = true, EPI = true (igemm_launch.h)
#include "igemm_launch.h"

namespace mmi_ig {
template int launch_igemm<true, true>(const IgemmP&, const FwdPlan&, bool, void*, size_t, hipStream_t, size_t, const IgemmP*);
template int launch_igemm_bf16<true, true>(IgemmP, const FwdPlan&, hipStream_t, const char*);
}  // namespace mmi_ig
