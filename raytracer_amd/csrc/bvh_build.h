// GPU build of the hierarchy the device traverses: a linear BVH over Morton-ordered primitive
// boxes, collapsed and quantised on the device into the same 4-wide 64-byte nodes the host builder produces
// (device_types.h: DNode4; host_bvh.cpp: to_node4).  The reference builds its tree on the host only
// (Code/acceleration/bvh.cpp:32-92); the primitive numbering -- the reference's depth-first visit order, which the
// tie rule of device_rt.h needs -- still comes from the host pass (host_bvh.cpp: build_bvh).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "device_types.h"
#include "host_math.h"

namespace rt {

struct DeviceBvh4 {
    DNode4* nodes = nullptr;   // device memory (hipMalloc); owned by the caller after LbvhStatus::ok
    int32_t n_nodes = 0;
    int32_t root_ref = 0;
    int32_t max_depth = 0;     // 4-wide levels
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
    double pad = 0;
    float build_ms = 0;        // device time from the first kernel to the last (upload excluded)
};

enum class LbvhStatus {
    ok,
    too_deep,       // more 4-wide levels than the traversal stack allows: the caller falls back to the host builder
    device_error    // *err holds the HIP error
};

// boxes: one per primitive, in primitive order (n >= 2); root_box: their union; max_abs_origin: as for to_node4.
// binned_sah: the binary tree comes from the binned surface-area heuristic (bvh_sah.hip) instead of Morton order.
LbvhStatus build_lbvh_device(const std::vector<Box>& boxes, const Box& root_box, double max_abs_origin, hipStream_t stream,
                             DeviceBvh4* out, hipError_t* err, bool binned_sah = false);

// bvh_sah.hip: the binary tree of host_bvh.cpp: SahBuilder built on the device over d_boxes (n x 6 doubles, primitive
// order).  d_vals[i] = primitive at leaf position i; interior node k: children d_left[k] / d_right[k] (>= 0 interior,
// < 0 leaf = ~position), covering positions d_range_lo[k] .. d_range_hi[k].  Synchronises the stream.
hipError_t sah_binary_tree_device(const double* d_boxes, int n, int max_depth, hipStream_t stream, unsigned* d_vals, int* d_left, int* d_right,
                                  int* d_range_lo, int* d_range_hi, bool* incomplete);

// The reference's depth-first visit order of the shapes (bvh.cpp:32-92), computed on the device from the shapes' world
// boxes in scene-file order (bvh_order.hip).  order[k] = scene-file index of the k-th primitive.
enum class OrderStatus {
    ok,
    ties,           // equal box centres meet at a split: the reference's result depends on its nth_element; use the host pass
    device_error
};
OrderStatus reference_order_device(const std::vector<Box>& shape_boxes, hipStream_t stream, std::vector<int32_t>* order, float* device_ms,
                                   hipError_t* err);

}  // namespace rt
