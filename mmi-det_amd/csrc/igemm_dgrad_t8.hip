// igemm_kernel instantiations: data gradient, pre-split operands (igemm_launch_t8.h)
#include "igemm_launch_t8.h"

namespace mmi_ig {
template int launch_igemm_t8<true>(const IgemmP&, const IgemmDelta&, const FwdPlan&, dim3, int, hipStream_t);
}  // namespace mmi_ig
