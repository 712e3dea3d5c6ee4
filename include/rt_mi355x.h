/*
 * rt_mi355x.h — C-ABI of the MI355X-native render core (librt_mi355x.so).
 *
 * Drop-in boundary for the hot path of alexsykes8/raytracer: the body of `render_scene_func`
 * (reference Code/main.cpp:229-359) — camera rays -> BVH closest hit -> Whitted/Blinn-Phong shading ->
 * tone map -> RGB8 framebuffer.  The reference exposes no FFI of its own; every entry point below cites the
 * reference interface it replaces.  Plain pointers and sizes only: no STL, no exceptions, no torch types.
 *
 * Conventions: every int-returning function returns 0 on success and a negative RT_ERR_* code on failure;
 * rt_last_error() then returns a thread-local description (the reference reports errors as C++ exceptions caught
 * in main, Code/main.cpp:562-565).  No ownership is transferred across the ABI: the library copies what it
 * needs during the call.  One host thread may call the rt_render / rt_scene functions on a given scene at a time.
 * The calls that deliver frames into HOST memory (rt_render, rt_render_frames, rt_render_gathered) run on the library's own
 * stream and staging buffers and serialise on a lock inside the library; the *_device calls render on the caller's streams
 * and may overlap (one render at a time per stream).
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK                 0
#define RT_ERR_INVALID       -1   /* bad argument / malformed description */
#define RT_ERR_IO            -2   /* file could not be opened / parsed */
#define RT_ERR_DEVICE        -3   /* HIP runtime error, or no usable GPU */
#define RT_ERR_NOT_INIT      -4
#define RT_ERR_LIMIT         -5   /* a documented capacity limit was exceeded */

/* ------------------------------------------------------------------------------------------------
 * Scene description (POD).  Mirrors what a fully constructed reference `Scene` holds
 * (Code/utilities/scene.h:18-34) before its BVH wrap: camera, lights, shapes with their matrices and
 * materials, textures, optional HDR environment.
 * ---------------------------------------------------------------------------------------------- */

/* Shape kinds == the scene-file block names (Code/utilities/scene.cpp:188-246). */
enum {
    RT_SHAPE_SPHERE = 0,          /* Code/shapes/sphere.cpp        */
    RT_SHAPE_PLANE = 1,           /* Code/shapes/plane.cpp  (quad = two triangles) */
    RT_SHAPE_CUBE = 2,            /* Code/shapes/cube.cpp          */
    RT_SHAPE_COMPLEX_SPHERE = 3,  /* Code/shapes/complex_sphere.cpp (displaced, ray-marched) */
    RT_SHAPE_COMPLEX_CUBE = 4,    /* Code/shapes/complex_cube.cpp  */
    RT_SHAPE_COMPLEX_PLANE = 5    /* Code/shapes/complex_plane.cpp */
};

/* Camera constructor arguments, Code/environment/camera.cpp:27-38 (sensor_height is overwritten by
 * sensor_width/aspect exactly as the reference does, camera.cpp:49-55). */
typedef struct RtCamera {
    double location[3];
    double gaze_direction[3];
    double up_vector[3];
    double focal_length;      /* mm */
    double sensor_width;      /* mm */
    double sensor_height;     /* mm (ignored, see above) */
    int32_t resolution_x;
    int32_t resolution_y;
    double f_stop;            /* scene.cpp:137 default 99999 */
    double focal_distance;    /* scene.cpp:138 default 10    */
} RtCamera;

/* Code/environment/light.h:13-28 */
typedef struct RtLight {
    double position[3];
    double intensity[3];
    double radius;
} RtLight;

/* Code/shapes/material.h:17-51.  texture / bump_map index RtSceneDesc.textures, -1 = none.
 * is_metal = (Material::type == "metal"), tracer.h:216. */
typedef struct RtMaterial {
    double ambient[3];
    double diffuse[3];
    double specular[3];
    double shininess;
    double reflectivity;
    double transparency;
    double refractive_index;
    int32_t texture;
    int32_t bump_map;
    int32_t is_metal;
    int32_t reserved;
} RtMaterial;

/* One scene object.  transform / inv_transform are the row-major 4x4 matrices the reference hands to the shape
 * constructors (scene.cpp:285-290: T*Rz*Ry*Rx*S and its cofactor inverse); planes use `corners` instead
 * (scene.cpp:380-381).  max_displacement is what the Complex* constructors read from
 * advanced.displacement_strength (complex_sphere.cpp:17, complex_cube.cpp:21, complex_plane.cpp:20). */
typedef struct RtShape {
    int32_t type;             /* RT_SHAPE_*            */
    int32_t material;         /* index into materials  */
    double transform[16];
    double inv_transform[16];
    double corners[12];       /* c0,c1,c2,c3 (planes)  */
    double velocity[3];
    double max_displacement;
} RtShape;

/* RGB8 image, row 0 = top (Code/utilities/Image.h:46-47). */
typedef struct RtTexture {
    int32_t width;
    int32_t height;
    const uint8_t* rgb;       /* width*height*3 bytes  */
} RtTexture;

/* HDR environment (Code/environment/HDRImage.h:41): float RGB, row 0 = first row of the PFM payload.
 * present=1 with width==0 reproduces "HDR_BACKGROUND given but the file failed to load => black environment"
 * (scene.cpp:172, HDRImage.cpp:114). */
typedef struct RtEnvMap {
    int32_t present;
    int32_t width;
    int32_t height;
    int32_t reserved;
    const float* rgb;
} RtEnvMap;

typedef struct RtSceneDesc {
    RtCamera camera;
    int32_t n_lights;
    int32_t n_shapes;
    int32_t n_materials;
    int32_t n_textures;
    const RtLight* lights;
    const RtShape* shapes;
    const RtMaterial* materials;
    const RtTexture* textures;
    RtEnvMap env;
    double shutter_time;      /* Scene ctor argument (scene.h:16): sizes the motion-blur AABBs and the ray times */
} RtSceneDesc;

/* ------------------------------------------------------------------------------------------------
 * Render parameters == the remaining Scene ctor arguments (scene.h:16), the Config keys the hot loop reads
 * (Code/config.json) and main's per-render locals (main.cpp:246-249, :68).
 * ---------------------------------------------------------------------------------------------- */
enum { RT_TONEMAP_NONE = 0, RT_TONEMAP_REINHARD = 1, RT_TONEMAP_ACES = 2, RT_TONEMAP_FILMIC = 3 }; /* main.cpp:68 */
enum {
    RT_SCHEDULE_WAVEFRONT = 0,  /* ray queues in HBM: persistent traversal kernel + shading kernel per level (default) */
    RT_SCHEDULE_MEGAKERNEL = 1, /* one launch, one lane per pixel with a bounded pending-ray stack                   */
    /* The THROUGHPUT variant of the wavefront schedule ("FP32 is the throughput variant, reported with its
     * mismatch count"): the hit-deciding primitive tests of the hierarchy traversal run in FP32 where a kernel variant has
     * them -- today the sphere test of spheres-only scenes (BASELINE config 2); every other scene renders exactly as with
     * RT_SCHEDULE_WAVEFRONT.  NOT the parity mode: silhouette pixels may differ from the reference by more than one level
     * (tests/test_gpu_fp32.py and bench.py's `fp32_variant` count them).  rt_render, rt_render_device and
     * rt_render_frames_device accept it. */
    RT_SCHEDULE_WAVEFRONT_FP32 = 2
};
enum {
    RT_RNG_D0 = 0,      /* deterministic: every uniform 0.5, ball/disk samples 0 (oracle mode "d0") */
    RT_RNG_SEEDED = 1   /* counter-based generator keyed by (seed, pixel, sample, draw index)     */
};

typedef struct RtRenderParams {
    int32_t samples_per_pixel;   /* --aa / settings.samples_per_pixel, main.cpp:54,113           */
    int32_t max_depth;           /* MAX_DEPTH, settings.max_bounces default 10, main.cpp:249      */
    int32_t max_bounces;         /* Scene::m_max_bounces, default 5, scene.cpp:96                 */
    int32_t tonemap;             /* RT_TONEMAP_*                                                  */
    int32_t shadows;             /* --shadows                                                     */
    int32_t fresnel;             /* --fresnel                                                     */
    int32_t normals;             /* --normals                                                     */
    int32_t use_bvh;             /* 0 == --no-bvh: linear list in scene-file order                */
    int32_t glossy_samples;      /* 0 unless --glossy (then render.glossy_samples), main.cpp:152  */
    int32_t shadow_samples;      /* render.shadow_samples default 4, scene.cpp:94                 */
    int32_t ray_march_steps;     /* advanced.ray_march_steps default 64, complex_*.cpp            */
    int32_t rng_mode;            /* RT_RNG_*                                                      */
    double exposure;             /* image.exposure / --exposure                                   */
    double eps_scene;            /* advanced.epsilon default 1e-4  (scene.cpp:95, tracer.h:152)   */
    double eps_triangle;         /* advanced.epsilon default 0.001 (plane.cpp:83)                 */
    double eps_march;            /* advanced.epsilon default 0.001 (complex_*.cpp)                */
    double eps_shadow;           /* advanced.epsilon default 1e-4  (shading.h:83)                 */
    double step_multiplier;      /* advanced.step_multiplier default 0.8 (complex_plane.cpp:167)  */
    double background[3];        /* background.{r,g,b} defaults .5 .7 1 (tracer.h:290-292)        */
    double ambient_light[3];     /* lighting.{r,g,b} defaults .15 .15 .25 (shading.h:142-144)     */
    uint64_t seed;               /* RT_RNG_SEEDED only                                            */
    /* Image partition for multi-GPU rendering (no counterpart in the reference, which shares one Image between
     * OpenMP threads, main.cpp:242,279): the frame is cut into row tiles of `tile_rows` rows; tile k belongs to
     * part (k % part_count).  part_count<=1 renders the whole frame. */
    int32_t tile_rows;
    int32_t part_index;
    int32_t part_count;
    int32_t schedule;            /* RT_SCHEDULE_*: how the device evaluates the ray tree (same pixels either way) */
} RtRenderParams;

#define RT_N_COUNTERS 8

typedef struct RtStats {
    uint64_t rays;            /* closest-hit queries == reference world.intersect calls (tracer.h:156, shading.h:45) */
    uint64_t nodes_visited;   /* BVH node boxes tested   (filled only when collect_counters != 0) */
    uint64_t prims_tested;    /* primitive intersection tests (same)                              */
    double kernel_ms;         /* device time of the whole render (all kernels, hipEvent)            */
    double traverse_ms;       /* of which: the BVH traversal kernel launches (wavefront schedule)   */
    double total_ms;          /* wall time of the call incl. copies                                */
    uint64_t nodes_fetched;   /* 64-byte hierarchy nodes fetched (each holds up to four child boxes; same)   */
    uint64_t rays_coincident; /* of `rays`: queries the reference repeats with the very same ray -- the 2 x shadow_samples rays of a hit
                                 towards a light of radius 0 (shading.h:31-36,154,192) -- which the wavefront schedule traces once
                                 and counts as the reference does: rays - rays_coincident closest-hit searches were executed */
} RtStats;

typedef struct rt_scene_s* rt_scene_t;
typedef struct rt_config_s* rt_config_t;
typedef struct rt_scene_file_s* rt_scene_file_t;

/* ---- device lifecycle ------------------------------------------------------------------------- */

/* Select the GPU this process renders on and create its stream.  One process per GPU (torchrun / RCCL ranks);
 * device_id < 0 keeps the current HIP device.  Fails with RT_ERR_DEVICE when no gfx950-compatible device exists:
 * there is no CPU fallback.  (The reference has no device: OpenMP threads, main.cpp:257-264.) */
int rt_init(int device_id);
void rt_shutdown(void);
const char* rt_last_error(void);

/* ---- scene ------------------------------------------------------------------------------------- */

/* Build the device scene: camera basis (camera.cpp:74-86), per-shape inverse transforms, quads as two
 * triangles (plane.cpp:49-74), world AABBs (transformed_shape.h:42-77, plane.cpp:11-46, complex_*::getBoundingBox),
 * the reference's visit order of the shapes (bvh.cpp:32-92; on the GPU for large scenes), the hierarchy the device
 * traverses as 64-byte 4-wide nodes (rt_set_bvh_builder), materials, textures, environment; upload all of it to HBM.  Replaces `Scene::Scene` step scene.cpp:102-121 + the object graph the hot loop walks. */
int rt_scene_create(const RtSceneDesc* desc, rt_scene_t* out);
void rt_scene_destroy(rt_scene_t scene);

/* Host-only BVH facts (no GPU needed): for tests and for the roofline's bytes-per-node figure. */
typedef struct RtBvhInfo {
    int32_t n_nodes;
    int32_t n_prims;
    int32_t max_depth;
    int32_t node_bytes;
    double root_min[3];
    double root_max[3];
} RtBvhInfo;
int rt_scene_bvh_info(rt_scene_t scene, RtBvhInfo* out);
/* Same build without any device work: returns the facts and, when order_out != NULL (n_shapes ints), the
 * depth-first leaf order of the shapes — the order the reference's BVHNode::intersect visits them. */
int rt_bvh_build_host(const RtSceneDesc* desc, RtBvhInfo* info_out, int32_t* order_out);
/* The hierarchy as the device traverses it (also host-only): a binned-SAH tree over the reference's per-shape boxes
 * (transformed_shape.h:42-77, plane.cpp:11-46), collapsed to four children per 64-byte node with child boxes quantised
 * outward to 8 bits per plane.  Node layout (little endian, 16 dwords): float org[3]; u32 exps (biased FP32 exponent
 * of the grid step per axis, x | y<<8 | z<<16); u32 qlo[3], qhi[3] (per axis, byte c = plane of child c: plane =
 * org + q * 2^(exp-127)); i32 child[4] (>= 0 node index, < 0 leaf: ~child = first_primitive << 1 | (count - 1),
 * 0x7fffffff unused); 2 dwords padding.  Primitives are numbered in the reference's depth-first order
 * (rt_bvh_build_host's order_out).  nodes_out: room for max_nodes nodes or NULL; info_out: counts, depth in 4-wide
 * levels and the widened FP32 root box; root_ref_out: node 0 or a leaf reference; pad_out: the outward pad applied to
 * every box before quantisation; prim_boxes_out: n_shapes x {min xyz, max xyz} in primitive order, or NULL.
 * For tests of the host builder; the reference has its pointer tree only (bvh.h:37-39). */
int rt_bvh_device_image(const RtSceneDesc* desc, void* nodes_out, int32_t max_nodes, RtBvhInfo* info_out, int32_t* root_ref_out,
                        double* pad_out, double* prim_boxes_out);

/* Which builder rt_scene_create uses for the hierarchy the device traverses (process-wide, for scenes created
 * afterwards).  The primitive numbering -- the reference's depth-first visit order (bvh.cpp:32-92), which decides
 * equal-distance hits -- always comes from the host pass; the closest hit does not depend on the hierarchy, so all
 * builders yield the same frames.
 *   RT_BVH_SAH     binned surface-area heuristic on host threads (fewest node visits per ray)
 *   RT_BVH_LBVH    built on the GPU: Morton order, radix sort, binary radix tree, segment-tree boxes, 4-wide collapse and
 *                  8-bit quantisation in HIP kernels.  Milliseconds for 10^6 shapes; falls back
 *                  to RT_BVH_SAH when the tree would be deeper than the traversal stack (32 4-wide levels).
 *   RT_BVH_MEDIAN  the reference's own topology (A/B runs)
 *   RT_BVH_SAH_GPU the same binned-SAH tree as RT_BVH_SAH, built on the GPU (node for node the host's tree unless a span of
 *                  more than two shapes cannot be split by bins; falls back like RT_BVH_LBVH)
 *   RT_BVH_DEFAULT environment variable RT_BVH_BUILDER = sah | lbvh | sah_gpu | median, else the SAH tree: RT_BVH_SAH_GPU
 *                  from 65536 shapes, RT_BVH_SAH below
 * The reference has one builder, on the host (bvh.cpp:32-92). */
#define RT_BVH_DEFAULT -1
#define RT_BVH_SAH      0
#define RT_BVH_LBVH     1
#define RT_BVH_MEDIAN   2
#define RT_BVH_SAH_GPU  3   /* the binned-SAH tree of RT_BVH_SAH built on the GPU, level by level over all spans (bvh_sah.hip) */
int rt_set_bvh_builder(int builder);
/* How a device scene was built, and its node array copied back to the host (layout: rt_bvh_device_image), for tests
 * of the device builders.  nodes_out: room for max_nodes nodes or NULL; info_out may be NULL. */
typedef struct RtBuildInfo {
    int32_t root_ref;         /* node 0 or a leaf reference */
    int32_t builder;          /* RT_BVH_*: the builder that actually built the traversal hierarchy */
    int32_t order_on_device;  /* 1: the reference's visit order was computed on the GPU, 0: on host threads */
    int32_t reserved;
    double pad;               /* outward pad of the node boxes */
    double device_build_ms;   /* RT_BVH_LBVH: device time of the hierarchy build */
    double device_order_ms;   /* order_on_device: device time of the order pass */
} RtBuildInfo;
int rt_scene_bvh_nodes(rt_scene_t scene, void* nodes_out, int32_t max_nodes, RtBuildInfo* info_out);
/* The reference's depth-first visit order of the shapes (what rt_bvh_build_host returns) computed on the GPU: level by
 * level, every span of the reference's top-down median split (bvh.cpp:32-92) is sorted by the box centre on the span's
 * axis and halved.  Returns RT_OK and fills order_out (n_shapes ints), or RT_ERR_LIMIT when equal centres meet at a
 * split -- the reference's result then depends on the internals of its std::nth_element, and the caller uses the host
 * pass (rt_scene_create does so by itself).  It is used for scenes of at least 65536 shapes unless RT_ORDER_ON_HOST is
 * set.  device_ms_out may be NULL. */
int rt_bvh_order_device(const RtSceneDesc* desc, int32_t* order_out, double* device_ms_out);

/* ---- render ------------------------------------------------------------------------------------ */

/* Whole-frame render into a caller-owned host buffer rgb8_out[H*W*3], row-major, top row first — the bytes of
 * `Image::m_pixel_data` after main.cpp:281-322 (Image.h:46-47).  With part_count>1 only the rows of tiles
 * owned by part_index are written; the rest of the buffer is left untouched.  stats may be NULL. */
int rt_render(rt_scene_t scene, const RtRenderParams* params, uint8_t* rgb8_out, RtStats* stats);

/* `frames` whole frames of the same scene into host memory, frame f with seed params->seed + f: rgb8_out receives
 * frames x H x W x 3 bytes, frame after frame, each as rt_render writes it.  What the reference does with `--time N`
 * (main.cpp:486-507: render_scene_func N times) when the frames are wanted, and the throughput form of the drop-in: the frames
 * are rendered in batches that share one set of launches (rt_render_frames_device) and the device -> host copy of a batch
 * overlaps the render of the next.  Wavefront schedule, whole frames (part_count <= 1).  stats (may be NULL): rays and device
 * time of all frames together.
 * The copies go through pinned staging buffers the library owns; the caller's buffer is ordinary memory and is only written by
 * the calling thread(s) before the call returns.  Opt-in, environment RT_HOST_REGISTER=second|always: rt_render and
 * rt_render_frames register the caller's buffer with the driver instead (hipHostRegister, the second / first time they see it;
 * at most 8 ranges, least recently used first out, all released by rt_shutdown), which makes the copy one DMA -- and obliges
 * the caller to keep such a buffer allocated until rt_shutdown (a registration pins the pages behind the addresses). */
int rt_render_frames(rt_scene_t scene, const RtRenderParams* params, int frames, uint8_t* rgb8_out, RtStats* stats);

/* Same pixels, but into device memory and asynchronously on `hip_stream` (a hipStream_t with HIP's own meaning: NULL is
 * the null stream, which is also PyTorch's default stream, so whatever the caller enqueues there afterwards is ordered
 * behind the render): d_rgb8 receives this part's rows compacted in tile order (rt_part_rows() rows of W*3 bytes) — the slice
 * a rank hands to the RCCL gather.  d_counters may be NULL or a device pointer to RT_N_COUNTERS (8) uint64 (rays, box tests,
 * primitive tests, nanoseconds spent in the traversal kernel, hierarchy nodes fetched, most nodes fetched by one ray, most scheduler iterations of one wave, rays counted but not traced: RtStats.rays_coincident) that the
 * kernel adds into (the two maxima are max-combined); collect_counters!=0 additionally counts node/primitive tests. */
int rt_render_device(rt_scene_t scene, const RtRenderParams* params, void* d_rgb8, void* d_counters,
                     int collect_counters, void* hip_stream);

/* `frames` frames of the same scene and the same part in one set of launches (the pixel loop of main.cpp:281-322 run
 * `frames` times: successive frames of a throughput run, or passes of a progressive render; frame f uses seed
 * params->seed + f, so in RT_RNG_D0 mode all frames are equal).  d_rgb8 receives the frames' slices one after another,
 * slice_rows rows of W*3 bytes apart (slice_rows >= rt_part_rows(); 0 = rt_part_rows(): the padded equal slices a gather
 * over ranks wants are written in place); each slice holds what rt_render_device writes.  Why: the persistent traversal kernel ends with a tail of
 * its longest rays that does not shrink with the launch, and a rank's share of a frame on an 8-GPU node is small; the
 * rays of all frames share one queue (8-row bands of the frames alternate, so every XCD still works on one region of
 * the image) and one tail.  Wavefront schedule only.  d_counters as in rt_render_device (sums over the frames). */
int rt_render_frames_device(rt_scene_t scene, const RtRenderParams* params, int frames, int slice_rows, void* d_rgb8,
                            void* d_counters, void* hip_stream);

/* rt_render_device for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc,
 * code under HIP graph capture -- for the frames named below).  It has NO counterpart in the reference (which has no
 * device): same pixels as rt_render_device with the wavefront schedule, but every buffer the kernels need beyond the scene
 * and d_rgb8 -- ray queues, hit records, FP64 accumulators, queue counters -- lies in a 64 MiB __device__ array inside the
 * library (one copy per GPU, held until the process ends), and the call allocates and frees nothing on the device (the
 * pinned host words of its read-backs are allocated by rt_init).
 * What the call does on the stream: a frame that cannot queue a ray (no material reflects or transmits and shadows are off,
 * or --normals: BASELINE config 2) is enqueued and the call returns at once, like rt_render_device -- nothing is read back,
 * so such a call is legal under stream capture.  A frame that CAN queue rays is cut into chunks, and after every chunk the
 * host reads 24 bytes back (queue overflow flag, rays still queued, ray total) and WAITS for the stream
 * (hipMemcpyAsync + hipStreamSynchronize, illegal during capture): the call blocks until all but the last kernels of the frame
 * have run.  Do not call it under capture for such frames.
 * Limits (RT_ERR_LIMIT with a message): the rendered part (rt_part_rows x width) holds at most 2^18 pixels and is at most
 * 1024 pixels wide -- larger frames are rendered in parts -- an 8-row band of the part must fit one of the queues that fit the
 * array, and so must the frame's ray tree (one queue per ray-tree level; chains of shadow segments through many transmitting
 * surfaces may not).
 * All calls on a GPU share the array: calls that overlap on one GPU (different streams or threads) race; run them one
 * after the other. */
int rt_render_device_noalloc(rt_scene_t scene, const RtRenderParams* params, void* d_rgb8, void* hip_stream);

/* Ray queues of the wavefront schedule are kept per HIP stream for the life of the process, not per scene: a service that
 * opens and closes scenes keeps them.  rt_reserve allocates (or grows to) n_queues queues of rays_per_queue rays (144 bytes
 * per ray) for `hip_stream` and touches every page, so that the first frame rendered on that stream pays neither the
 * allocation nor the first touch (seconds, for tens of GB); rt_render_device otherwise sizes the queues by the first
 * frame and grows them when a larger one comes.  rt_release_workspaces waits for the device and frees all of them.
 * No counterpart in the reference (host memory, Code/main.cpp:242). */
int rt_reserve(void* hip_stream, uint64_t rays_per_queue, int n_queues);
int rt_release_workspaces(void);

/* Linear radiance: the per-pixel average over the samples BEFORE tone mapping and 8-bit truncation (the `avg_colour`
 * of main.cpp:306-308), 3 doubles per pixel, row-major, top row first.  For float (PFM) output and for progressive /
 * resumable accumulation: frames rendered with different seeds average to a frame of the summed sample count; apply
 * tracer.h:23-54,297-308 afterwards (raytracer_amd.progressive does).  Wavefront schedule only.  rt_render_linear writes
 * a whole frame to host memory; rt_render_linear_device writes this part's rows (compact, tile order) to device memory
 * asynchronously on `hip_stream`.  The reference only ever stores the tone-mapped bytes. */
int rt_render_linear(rt_scene_t scene, const RtRenderParams* params, double* rgb_out);
int rt_render_linear_device(rt_scene_t scene, const RtRenderParams* params, void* d_rgb_f64, void* hip_stream);

/* Progressive, resumable accumulation with the running sum in HBM (the reference's sample loop,
 * main.cpp:288-308, is all or nothing).  rt_accumulate_device renders `passes` passes of params->samples_per_pixel
 * samples each -- pass k of the series with seed params->seed + first_pass + k (RT_RNG_SEEDED) -- and ADDS each pass's
 * linear average to d_sum_f64 (3 doubles per pixel of this part, compact tile order, zeroed by the caller before the
 * first pass), asynchronously on `hip_stream`: nothing crosses PCIe until the caller wants a picture or a checkpoint
 * (sum + pass count).  The passes run one after the other on `hip_stream`; within a pass of many samples the chunks of
 * primary rays are rendered two at a time like those of any large frame (a second set of ray queues on a stream and a host
 * thread of the library's own, both adding into the pass's accumulators with FP64 atomics: the order of the additions, and
 * with it the last bits of the sum, may differ from run to run; RT_ONE_CHUNK_AT_A_TIME=1: one by one).
 * rt_resolve_sum_device turns the sum of `passes` passes into the bytes Image::setPixel would hold:
 * divide, tone map, truncate (main.cpp:308-320, tracer.h:23-54,297-308). */
int rt_accumulate_device(rt_scene_t scene, const RtRenderParams* params, void* d_sum_f64, uint64_t first_pass, int passes, void* hip_stream);
int rt_resolve_sum_device(const void* d_sum_f64, uint64_t passes, int width, int rows, int tonemap, void* d_rgb8, void* hip_stream);

/* Device time (hipEvents on the launch stream) of the BVH-traversal kernel launches that the render entry points enqueued on
 * `hip_stream` since the last call of this function, and how many launches that was -- of EVERY scene rendered on that
 * stream: the ray queues and their event lists belong to the stream, not to the scene (rt_reserve), so `scene` only has to be
 * a valid handle; render one scene per stream between two calls to time that scene.  The stream must have been
 * synchronised by the caller (RT_ERR_DEVICE otherwise).  hip_stream == NULL means the NULL stream here and in every
 * *_device entry point (HIP's own meaning; round 1 used a stream of the library's own for NULL).  No counterpart in the reference: it feeds the
 * roofline figure of bench.py (the reference only has wall-clock timing, main.cpp:271,350-358). */
int rt_traverse_time(rt_scene_t scene, void* hip_stream, double* ms_out, uint64_t* launches_out);

/* Number of image rows owned by (part_index, part_count) for a frame of `height` rows cut into `tile_rows` tiles. */
int rt_part_rows(int height, int tile_rows, int part_index, int part_count);
/* Scatter a compact part slice (as written by rt_render_device) into a full row-major frame (host memory). */
int rt_part_scatter(const uint8_t* slice, uint8_t* frame, int width, int height, int tile_rows, int part_index,
                    int part_count);

/* ---- several GPUs: one process per GPU, slices gathered over RCCL / xGMI ------------------------
 * The reference shares one `Image` between its OpenMP threads (main.cpp:242,279-320: every thread's setPixel lands in the
 * same buffer).  With one process per GPU that buffer is assembled on rank 0: every rank renders its interleaved row
 * tiles into a compact slice (rt_render_device with part_index = rank, part_count = world), the slices travel to rank 0
 * as one grouped ncclSend / ncclRecv per rank (RCCL, bound at run time: single-GPU users never load it), and a HIP
 * kernel puts the rows where Image::m_pixel_data has them (Image.h:46-47).  No other exchange exists on the path.
 *
 * Bootstrap: rank 0 calls rt_comm_unique_id and hands the RT_COMM_ID_BYTES to the other ranks by whatever the launcher
 * offers (torch.distributed broadcast, a file, MPI); every rank then calls rt_comm_init after rt_init.  One communicator
 * per process.  rt_comm_rank / rt_comm_world return 0 / 1 without one. */
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(void* id_out);
int rt_comm_init(int rank, int world, const void* id);
int rt_comm_rank(void);
int rt_comm_world(void);
int rt_comm_destroy(void);
/* Collective (every rank, same arguments except the pointers), asynchronous on `hip_stream`.  d_slices: this rank's
 * slices of `frames` consecutive frames, device memory, frames x max_rows x width x 3 bytes, where max_rows is the
 * largest rt_part_rows over the ranks (frame b at offset b * max_rows * width * 3: a rank with fewer rows leaves the tail
 * of each slot unused) -- several frames travel in one message, because a rank's share of one frame is small (0.78 MB of
 * a 1080p frame at 8 ranks) and the exchange is bound by its launch cost.  d_frames (rank 0 only, else ignored):
 * frames x height x width x 3 bytes, row-major, top row first.  Without a communicator (one GPU) it only reorders.
 * Rank 0 keeps a staging buffer of world x frames x max_rows x width x 3 bytes PER STREAM it has gathered on, grown on demand
 * (growing waits for the whole device once) and kept until rt_comm_destroy: gathers on different streams may be in flight
 * together; RCCL orders the messages of the one communicator.  The call holds the library's communicator lock while it
 * enqueues, so it may be called from several threads.  An RCCL error inside the grouped send / receive closes the group
 * before it is reported.  A partial batch passes its own `frames`: the message is frames x max_rows x width x 3 bytes. */
int rt_gather_frames(const void* d_slices, int frames, int width, int height, int tile_rows, void* d_frames, void* hip_stream);
/* Device time of the gathers since the last call of this function, summed: `exchange` = the grouped ncclSend / ncclRecv of a
 * gather, `deinterleave` = its k_deinterleave launch (rank 0; zero elsewhere), from HIP events the library records on the
 * gather's stream, and how many gathers that was (at most the last 512).  The streams must have been synchronised by the
 * caller (RT_ERR_DEVICE otherwise).  It tells a slow exchange from a slow render on rank 0, which is receiver, de-interleaver
 * and a full renderer at once.  No counterpart in the reference (shared memory, main.cpp:242). */
int rt_gather_time(double* exchange_ms_out, double* deinterleave_ms_out, uint64_t* gathers_out);
/* rt_render for an N-GPU job -- the drop-in for the body of render_scene_func (main.cpp:229-359) when the frame is shared
 * by several GPUs.  Collective: every rank calls it with the same scene and params (part_index / part_count are set from
 * the communicator); each rank renders its tiles, the slices are gathered, and rank 0 gets the whole frame in rgb8_out
 * (H*W*3 bytes; other ranks may pass NULL).  stats (may be NULL) describe this rank's share.  Without a communicator it
 * is rt_render. */
int rt_render_gathered(rt_scene_t scene, const RtRenderParams* params, uint8_t* rgb8_out, RtStats* stats);
/* The reordering kernel alone: d_gathered = [world][frames][max_rows][width*3] as rt_gather_frames receives it; d_own
 * (may be NULL) replaces rank 0's block with [frames][max_rows][width*3] elsewhere.  The device-side rt_part_scatter. */
int rt_deinterleave_device(const void* d_own, const void* d_gathered, void* d_frames, int frames, int width, int height, int tile_rows,
                           int world, void* hip_stream);

/* ---- host side kept from the reference: config.json, scene files, images ---------------------- */

/* Code/config.cpp:7-103 (`Config::load/getInt/getDouble/getBool`).  A missing file is not an error for the
 * reference (warning + defaults); rt_config_load returns RT_ERR_IO but still yields an empty, usable config. */
int rt_config_load(const char* path, rt_config_t* out);
int rt_config_parse(const char* text, rt_config_t* out);
int rt_config_get_int(rt_config_t cfg, const char* key, int default_value);
double rt_config_get_double(rt_config_t cfg, const char* key, double default_value);
int rt_config_get_bool(rt_config_t cfg, const char* key, int default_value);
void rt_config_destroy(rt_config_t cfg);

/* Fill params with what main.cpp + Scene would use for this config and no command-line flags
 * (main.cpp:52-68,249; scene.cpp:94-96; call-site defaults for missing keys). */
int rt_params_from_config(rt_config_t cfg, RtRenderParams* out);

/* `Scene::parseSceneFile` (scene.cpp:124-546) for the Blender-ASCII format.  texture_root replaces the
 * reference's cwd-relative "../" prefix (scene.cpp:171,270): a texture named X is loaded from texture_root/X.
 * cfg supplies advanced.displacement_strength; shutter_time is the Scene ctor argument. */
int rt_scene_file_load(const char* scene_path, const char* texture_root, rt_config_t cfg, double shutter_time,
                       rt_scene_file_t* out);
int rt_scene_file_parse(const char* scene_text, const char* texture_root, rt_config_t cfg, double shutter_time,
                        rt_scene_file_t* out);
const RtSceneDesc* rt_scene_file_desc(rt_scene_file_t f);
void rt_scene_file_destroy(rt_scene_file_t f);

/* Compose shape matrices exactly as scene.cpp:277-287 does (T*Rz*Ry*Rx*S, cofactor inverse, matrix4x4.cpp):
 * trs = n x 9 doubles (translation xyz, rotation_euler_radians xyz, scale xyz); fills transform/inv_transform of
 * shapes[i].  For building large synthetic scenes without going through text. */
int rt_shapes_from_trs(int n, const double* trs, RtShape* shapes);

/* `Image::write` (Image.cpp:170-188): "P6\nW H\n255\n" + bytes.  `Image::read` (Image.cpp:84-167): P3/P6. */
int rt_write_ppm(const char* path, int width, int height, const uint8_t* rgb);
/* "PF" portable float map, little endian, rows bottom to top (readable by rt_read_pfm / HDRImage::load); rgb is the
 * top-first double frame of rt_render_linear. */
int rt_write_pfm(const char* path, int width, int height, const double* rgb);
int rt_read_ppm(const char* path, int* width, int* height, uint8_t** rgb_out /* free with rt_free */);
/* `HDRImage::load` (HDRImage.cpp:22-109): "PF" portable float map. */
int rt_read_pfm(const char* path, int* width, int* height, float** rgb_out /* free with rt_free */);
void rt_free(void* p);

const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
