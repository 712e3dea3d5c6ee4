// Layout of the scene in HBM, shared by the host uploader (api.cpp / host_bvh.cpp) and the render kernel.
// All geometry is FP64: the reference computes in double (Code/utilities/vector3.h:18) and the parity bar is
// 8-bit max|diff| <= 1, which FP32 hit/miss predicates do not meet.
#pragma once
#include <cstdint>

#include "../../include/rt_mi355x.h"

#if defined(__HIPCC__)
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

namespace rt {

constexpr int kNodeBytes = 64;        // one DNode4 (what the device traverses)
#ifndef RT_TRAV_LDS_ENTRIES
#define RT_TRAV_LDS_ENTRIES 14
#endif
constexpr int kTravLdsEntries = RT_TRAV_LDS_ENTRIES;   // per-lane traversal stack entries kept in LDS (8 bytes each)
constexpr int kTravSpillEntries = 80; // further entries in per-lane scratch memory (deep, unbalanced hierarchies only)
constexpr int kTravStackDepth = kTravLdsEntries + kTravSpillEntries;
constexpr int kMaxBinaryDepth = 31;   // interior levels of the binary tree; a 4-wide node defers <= 3 children per level
constexpr int kRayStackDepth = 32;    // per-lane pending-ray stack entries (scratch)
constexpr int kShadowMaxSegments = 256;

// Child / root reference: >= 0 -> index of an interior DNode; < 0 -> leaf holding 1 or 2 consecutive
// primitives: ~ref = (first_prim << 1) | (count - 1).
RT_HD inline int32_t make_leaf_ref(int first_prim, int count) { return ~((first_prim << 1) | (count - 1)); }

// Interior node of the flattened BVH.  It carries the boxes of BOTH children (the reference stores each
// BVHNode's own box, Code/acceleration/bvh.h:37-39, and tests it on entry, bvh.cpp:103; testing the two child
// boxes at the parent performs the same set of slab tests with half the dependent memory fetches).
// 128 bytes, 128-byte aligned: one node == one L2 cache line pair, fetched as 8 x 16 B.
struct alignas(128) DNode {
    double lo[2][3];   // [child][axis]
    double hi[2][3];
    int32_t child[2];
    int32_t pad[6];
};
static_assert(sizeof(DNode) == 128, "DNode must be 128 bytes");

// What the device actually traverses: the binary tree collapsed to four children per node (host_bvh.cpp: to_node4),
// child boxes quantised to 8 bits per plane on a per-node grid: plane = org[axis] + q * step[axis], step a power of two.  Boxes
// are moved outward before quantisation (lower planes rounded down, upper planes up), so the FP32 slab test of
// device_rt.h accepts every ray the reference's FP64 test (aabb.cpp:4-38) accepts on the original boxes.  Boxes only
// cull; every hit/miss decision is made by the FP64 primitive tests, so a conservative (superset) traversal yields
// the same closest hit.  64 bytes, 64-byte aligned: four 16-byte loads per lane and visit.
constexpr int32_t kEmptyChild = 0x7fffffff;
struct alignas(64) DNode4 {
    float org[3];        // grid origin (<= every lower plane)
    float step_x;        // grid step per axis, a power of two >= 2^-126, stored as the FP32 number itself (step_yz below): the
                         // node loop multiplies it by 1/d without decoding anything
    uint32_t qlo[3];     // [axis], byte c = lower plane of child c
    uint32_t qhi[3];     // [axis], byte c = upper plane of child c
    int32_t child[4];    // >= 0 interior node, < 0 leaf (make_leaf_ref), kEmptyChild = unused slot, whose box is INVERTED on every
                         // axis (lower plane 255, upper plane 0): no ray enters it, so the node loop never looks at the reference
    float step_yz[2];
};
static_assert(sizeof(DNode4) == kNodeBytes, "DNode4 must be 64 bytes");

// Per-primitive header, in BVH depth-first (leaf) order.
struct alignas(16) DPrim {
    int32_t type;       // RT_SHAPE_*
    int32_t data;       // index into xforms (transformed shapes) or quads (planes)
    int32_t material;
    int32_t file_index; // position in the scene file (== reference list order for --no-bvh)
};

// Transformed shapes (sphere / cube / complex_*): rows 0..2 of the world->object matrix, its [3][3] element
// (the reference divides by w = inv[3][3] when it is not exactly 1, matrix4x4.cpp:85-90), velocity, displacement.
// The inverse-transpose used for normals (transformed_shape.h:21) is the same 12 numbers read column-wise.
struct alignas(16) DXform {
    double inv[12];    // inv[r*4 + c], r = 0..2
    double inv33;
    double vel[3];
    double disp;       // max_displacement (complex shapes), else 0
    double pad;
};
static_assert(sizeof(DXform) == 144, "DXform size");

// Quad = two triangles with precomputed edges (plane.cpp:49-74).
struct alignas(16) DQuad {
    double v0[3], e1[3], e2[3];   // triangle 1: c0, c1-c0, c2-c0
    double w0[3], f1[3], f2[3];   // triangle 2: c1, c3-c1, c2-c1
    double n[3];                  // normalize(e1 x e2)
    double vel[3];
};
static_assert(sizeof(DQuad) == 192, "DQuad size");

struct alignas(16) DMaterial {
    double ambient[3];
    double diffuse[3];
    double specular[3];
    double shininess;
    double reflectivity;
    double transparency;
    double ior;
    int32_t texture;    // index into textures, -1 none
    int32_t bump;
    int32_t is_metal;
    int32_t pad;
};
static_assert(sizeof(DMaterial) == 128, "DMaterial size");

struct DTexture {
    const uint8_t* rgb;
    int32_t width, height;
};

struct DLight {
    double pos[3];
    double intensity[3];
    double radius;
};

struct DCamera {
    double location[3];
    double u[3], v[3], w[3];      // camera.cpp:74-86
    double focal_length, sensor_width, sensor_height;
    double aperture_radius, focal_distance;
    int32_t res_x, res_y;
};

// Everything the kernel needs to find the scene; passed by value as a kernel argument.
struct DScene {
    const DNode4* nodes;
    const DPrim* prims;           // depth-first order
    const int32_t* file_order;    // file_order[i] = prim index of the i-th shape in the scene file
    const DXform* xforms;
    const DQuad* quads;
    const DMaterial* materials;
    const DTexture* textures;
    const DLight* lights;
    const float* prim_boxes;      // scenes of at most kSmallScenePrims primitives: world box of primitive k, widened like the node
                                  // boxes, as lo xyz, hi xyz (FP32); else null
    const float* env;             // float RGB rows in file order, may be null
    const float* xforms32;        // FP32 copy of the transforms for RT_SCHEDULE_WAVEFRONT_FP32 (16 floats each: rows 0..2 of inv, then
                                  // velocity xyz and inv33), made on the first such frame of a spheres-only scene; else null
    int32_t env_present, env_w, env_h;
    int32_t n_prims, n_nodes, n_lights;
    int32_t n_xforms, n_quads, n_materials;
    int32_t root_ref;             // reference to the root (node 0 or a leaf)
    int32_t features;             // RT_FEAT_* bits: which code paths this scene can reach
    float root_lo[3], root_hi[3];   // widened like the node boxes
    int32_t spawning_materials;     // 1 when some material reflects or transmits (secondary radiance rays possible)
    int32_t transparent_materials;  // 1 when some material transmits (shadow rays then need their CLOSEST occluder)
    int32_t area_lights;            // 1 when some light has a radius (else the shadow rays of a hit towards a light coincide: one lane per bundle)
    int32_t reserved0;
    double shutter_time;
    DCamera cam;
    uint64_t serial;                // identifies the scene inside this process (host bookkeeping only)
};

enum : int32_t {
    RT_FEAT_QUAD = 1,
    RT_FEAT_CUBE = 2,
    RT_FEAT_COMPLEX = 4,
    RT_FEAT_TEXTURE = 8,    // any texture or bump map
    RT_FEAT_ENV = 16,
    RT_FEAT_SPHERE = 32,
    RT_FEAT_FP32 = 64      // never in DScene::features: selects the kernel variant with FP32 primitive tests (spheres-only scenes)
};

// Render parameters as the kernel sees them.
struct DParams {
    int32_t width, height;
    int32_t spp, max_depth, max_bounces, tonemap;
    int32_t shadows, fresnel, normals, use_bvh;
    int32_t glossy_samples, shadow_samples, march_steps, rng_mode;
    int32_t tile_rows, part_index, part_count, part_rows;  // part_rows = rows this part owns
    int32_t count_tests;   // also count node / primitive tests
    int32_t fp32_tests;    // RT_SCHEDULE_WAVEFRONT_FP32: the throughput variant (FP32 primitive tests where a kernel variant has them)
    // Several frames of the same scene in one launch (rt_render_frames_device): frame f is rendered with seed + f into the
    // slice that begins f * slice_rows rows into the output.  frames = 1, slice_rows = part_rows for every other entry point.
    int32_t frames, slice_rows;
    int32_t trace_coincident;   // RT_TRACE_COINCIDENT_RAYS=1 (A/B runs, tests): k_shadow traces every one of the coincident rays of a point light
    int32_t pad;
    double exposure, eps_scene, eps_triangle, eps_march, eps_shadow, step_multiplier;
    double background[3];
    double ambient_light[3];
    uint64_t seed;
};

}  // namespace rt
