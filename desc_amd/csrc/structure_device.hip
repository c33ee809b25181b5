// Device construction of the sampled 3-cycle structure.
#include "device_utils.h"

namespace desc {
int build_structure_device(const desc_problem*, int32_t, uint64_t, int32_t, desc_structure*) {
    return fail(DESC_ERR_INVALID, "DESC_BUILD_DEVICE is not implemented yet; use DESC_BUILD_HOST");
}
}  // namespace desc
