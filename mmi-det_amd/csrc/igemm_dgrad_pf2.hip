// igemm_kernel instantiations: data gradient, deep prefetch (igemm_launch_pf2.h)
#include "igemm_launch_pf2.h"

namespace mmi_ig {
template int launch_igemm_pf2<true>(const IgemmP&, const IgemmDelta&, const FwdPlan&, dim3, int, hipStream_t);
}  // namespace mmi_ig
