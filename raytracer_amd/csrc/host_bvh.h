// Host BVH construction: the reference's top-down median split (Code/acceleration/bvh.cpp:32-92) over the
// reference's per-shape world boxes, flattened for the device (device_types.h: DNode).
#pragma once
#include <functional>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "device_types.h"
#include "host_math.h"

namespace rt {

struct HostBvh {
    std::vector<DNode> nodes;      // interior nodes, node 0 = root (empty when the scene has <= 2 shapes)
    std::vector<int32_t> order;    // order[k] = scene-file index of the k-th primitive in depth-first leaf order
    std::vector<Box> shape_boxes;  // build_bvh: the world box of every shape, in scene-file order (callers may drop it)
    int32_t root_ref = 0;
    Box root_box;
    int32_t max_depth = 0;         // interior-node levels on the longest root->leaf path
};

// World-space box of one shape at build time (transformed_shape.h:42-77, plane.cpp:11-46,
// complex_sphere.cpp:21-27, complex_cube.cpp:25-31, complex_plane.cpp:24-75).
Box shape_bounds(const RtShape& s, double shutter_time);
// The box the TRAVERSAL hierarchy uses for a shape: the reference's box, except for (displaced) spheres, whose ellipsoid is
// bounded exactly instead of through the corners of its oriented box.  Contained in shape_bounds() and containing the
// surface: culling with it cannot change a hit.  RT_REFERENCE_BOXES=1 (environment) returns shape_bounds() for A/B runs.
Box traversal_bounds(const RtShape& s, double shutter_time);
// traversal_bounds of shape order[k] for every k (host threads): the boxes of the primitives in primitive order.
std::vector<Box> traversal_boxes(const RtSceneDesc& desc, const std::vector<int32_t>& order);
// The boxes of all shapes (host threads); box k belongs to shape order[k], or to shape k when order is NULL.
std::vector<Box> shape_boxes(const RtSceneDesc& desc, const int32_t* order);
// The union of all boxes (host threads); an empty list yields an inverted box.
Box union_of(const std::vector<Box>& boxes);
// boxes[order[k]] for every k (host threads).
std::vector<Box> permuted_boxes(const std::vector<Box>& boxes, const std::vector<int32_t>& order);
// body(begin, end) over disjoint pieces of [0, n) on up to RT_BUILD_THREADS host threads (one piece when n is small).
void parallel_ranges(size_t n, const std::function<void(size_t, size_t)>& body);

// Device image of the hierarchy: the binary tree collapsed to four children per node with 8-bit quantised child
// boxes (device_types.h: DNode4).  Every box is first moved outward by an absolute pad that covers the FP32 rounding of
// ray origins and of the device's slab arithmetic for coordinates up to max_abs_origin / the root box, then quantised
// outward; relative rounding is covered on the device (device_rt.h: node4_test).
struct Bvh4 {
    std::vector<DNode4> nodes;     // node 0 = root (empty when the root is a leaf)
    float root_lo[3], root_hi[3];
    double pad = 0;
    int32_t root_ref = 0;
    int32_t max_depth = 0;         // 4-wide levels on the longest root->leaf path
};
Bvh4 to_node4(const HostBvh& bvh, double max_abs_origin);
// The absolute outward pad of every box of the device image, and the FP32 roundings used for the root box.
double node4_pad(const Box& root_box, double max_abs_origin);
float round_down(double v);
float round_up(double v);

// What the device traverses: a binned-SAH hierarchy with single-primitive leaves over `boxes` (one per primitive,
// indexed by the reference's depth-first order, so that leaf references are primitive indices).  At most max_depth
// interior levels (the per-lane traversal stack), falling back to balanced splits where the bound is tight.
HostBvh build_traversal_bvh(const std::vector<Box>& boxes, int max_depth);

// All three builders run their subtrees on up to RT_BUILD_THREADS host threads (default: min(16, cores)); their
// results do not depend on the number of threads.
// n_shapes == 0 yields an empty hierarchy (root_ref is a leaf of zero primitives; callers must check).
HostBvh build_bvh(const RtSceneDesc& desc);

}  // namespace rt
