// Host BVH construction: the reference's top-down median split (Code/acceleration/bvh.cpp:32-92) over the
// reference's per-shape world boxes, flattened for the device (device_types.h: DNode).
#pragma once
#include <vector>

#include "../../include/rt_mi355x.h"
#include "device_types.h"
#include "host_math.h"

namespace rt {

struct HostBvh {
    std::vector<DNode> nodes;      // interior nodes, node 0 = root (empty when the scene has <= 2 shapes)
    std::vector<int32_t> order;    // order[k] = scene-file index of the k-th primitive in depth-first leaf order
    int32_t root_ref = 0;
    Box root_box;
    int32_t max_depth = 0;         // interior-node levels on the longest root->leaf path
};

// World-space box of one shape at build time (transformed_shape.h:42-77, plane.cpp:11-46,
// complex_sphere.cpp:21-27, complex_cube.cpp:25-31, complex_plane.cpp:24-75).
Box shape_bounds(const RtShape& s, double shutter_time);

// FP32 image of the hierarchy for the device.  Every box is moved outward by an absolute pad that covers the rounding
// of a ray origin to FP32 (|o| <= max_abs_coord, so |fl32(o) - o| <= 2^-24 max_abs_coord) and then rounded outward
// to FP32; the relative rounding of the slab arithmetic itself is covered on the device (device_rt.h: box_test32).
struct Bvh32 {
    std::vector<DNode32> nodes;
    float root_lo[3], root_hi[3];
    double pad = 0;
};
Bvh32 to_node32(const HostBvh& bvh, double max_abs_origin);

// What the device traverses: a binned-SAH hierarchy with single-primitive leaves over `boxes` (one per primitive,
// indexed by the reference's depth-first order, so that leaf references are primitive indices).  At most max_depth
// interior levels (the per-lane traversal stack), falling back to balanced splits where the bound is tight.
HostBvh build_traversal_bvh(const std::vector<Box>& boxes, int max_depth);

// n_shapes == 0 yields an empty hierarchy (root_ref is a leaf of zero primitives; callers must check).
HostBvh build_bvh(const RtSceneDesc& desc);

}  // namespace rt
