// rx_fused.hip -- fused fast paths (placeholder: no configuration covered yet).
#include "rx_internal.h"
namespace srx {
FusedPlan plan_fused(const selenite_rx_config &, bool, int, bool) { return FusedPlan{}; }
hipError_t launch_fused(const FusedPlan &, const RxParams &, int, const void *, bool, void *, bool, int, hipStream_t)
{
    return hipErrorNotSupported;
}
}  // namespace srx
