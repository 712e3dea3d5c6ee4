// Host-callable launcher of the render kernel (render_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "device_types.h"

namespace rt {

// Renders this part's rows (params.part_rows rows of params.width pixels, compact, tile order) into d_rgb8 on
// `stream`.  d_counters: NULL or 4 x u64 {rays, node visits, primitive tests, reserved} that the kernel adds into.
hipError_t launch_render(const DScene& scene, const DParams& params, uint8_t* d_rgb8, unsigned long long* d_counters,
                         hipStream_t stream);

}  // namespace rt
