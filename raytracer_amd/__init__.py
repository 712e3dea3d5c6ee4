"""raytracer_amd — Python host-side mirror of the reference's render interface over librt_mi355x.so.

The reference is a C++ program whose render seam is `render_scene_func(scene_path, use_bvh, output_path)`
(Code/main.cpp:229-359) built from `Config` (Code/config.cpp), `Scene` (Code/utilities/scene.cpp) and `Image`
(Code/utilities/Image.cpp).  This module binds the C-ABI of include/rt_mi355x.h with ctypes and exposes the same
three objects — `Config`, `SceneFile`/`Scene`, `render` — for tests, bench.py and scripting.  All rendering happens
in the HIP kernels inside the shared library; there is no Python or CPU rendering path here, and importing the
module fails loudly when the library has not been built.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_MI355X_LIB") or os.path.join(_HERE, "librt_mi355x.so")  # override: A/B builds of the same ABI

RT_SHAPE_SPHERE, RT_SHAPE_PLANE, RT_SHAPE_CUBE = 0, 1, 2
RT_SHAPE_COMPLEX_SPHERE, RT_SHAPE_COMPLEX_CUBE, RT_SHAPE_COMPLEX_PLANE = 3, 4, 5
RT_TONEMAP_NONE, RT_TONEMAP_REINHARD, RT_TONEMAP_ACES, RT_TONEMAP_FILMIC = 0, 1, 2, 3
RT_RNG_D0, RT_RNG_SEEDED = 0, 1
RT_SCHEDULE_WAVEFRONT, RT_SCHEDULE_MEGAKERNEL, RT_SCHEDULE_WAVEFRONT_FP32 = 0, 1, 2
RT_ERR_INVALID, RT_ERR_IO, RT_ERR_DEVICE, RT_ERR_NOT_INIT, RT_ERR_LIMIT = -1, -2, -3, -4, -5
TONEMAP_BY_NAME = {"none": 0, "reinhard": 1, "aces": 2, "filmic": 3}  # main.cpp:198-216


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rt error %d: %s" % (code, message))
        self.code = code


# ---------------------------------------------------------------------------------------- POD structs
class RtCamera(C.Structure):
    _fields_ = [("location", C.c_double * 3), ("gaze_direction", C.c_double * 3), ("up_vector", C.c_double * 3),
                ("focal_length", C.c_double), ("sensor_width", C.c_double), ("sensor_height", C.c_double),
                ("resolution_x", C.c_int32), ("resolution_y", C.c_int32), ("f_stop", C.c_double),
                ("focal_distance", C.c_double)]


class RtLight(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("intensity", C.c_double * 3), ("radius", C.c_double)]


class RtMaterial(C.Structure):
    _fields_ = [("ambient", C.c_double * 3), ("diffuse", C.c_double * 3), ("specular", C.c_double * 3),
                ("shininess", C.c_double), ("reflectivity", C.c_double), ("transparency", C.c_double),
                ("refractive_index", C.c_double), ("texture", C.c_int32), ("bump_map", C.c_int32),
                ("is_metal", C.c_int32), ("reserved", C.c_int32)]


class RtShape(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("transform", C.c_double * 16),
                ("inv_transform", C.c_double * 16), ("corners", C.c_double * 12), ("velocity", C.c_double * 3),
                ("max_displacement", C.c_double)]


class RtTexture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_uint8))]


class RtEnvMap(C.Structure):
    _fields_ = [("present", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("reserved", C.c_int32),
                ("rgb", C.POINTER(C.c_float))]


class RtSceneDesc(C.Structure):
    _fields_ = [("camera", RtCamera), ("n_lights", C.c_int32), ("n_shapes", C.c_int32), ("n_materials", C.c_int32),
                ("n_textures", C.c_int32), ("lights", C.POINTER(RtLight)), ("shapes", C.POINTER(RtShape)),
                ("materials", C.POINTER(RtMaterial)), ("textures", C.POINTER(RtTexture)), ("env", RtEnvMap),
                ("shutter_time", C.c_double)]


class RtRenderParams(C.Structure):
    _fields_ = [("samples_per_pixel", C.c_int32), ("max_depth", C.c_int32), ("max_bounces", C.c_int32),
                ("tonemap", C.c_int32), ("shadows", C.c_int32), ("fresnel", C.c_int32), ("normals", C.c_int32),
                ("use_bvh", C.c_int32), ("glossy_samples", C.c_int32), ("shadow_samples", C.c_int32),
                ("ray_march_steps", C.c_int32), ("rng_mode", C.c_int32), ("exposure", C.c_double),
                ("eps_scene", C.c_double), ("eps_triangle", C.c_double), ("eps_march", C.c_double),
                ("eps_shadow", C.c_double), ("step_multiplier", C.c_double), ("background", C.c_double * 3),
                ("ambient_light", C.c_double * 3), ("seed", C.c_uint64), ("tile_rows", C.c_int32),
                ("part_index", C.c_int32), ("part_count", C.c_int32), ("schedule", C.c_int32)]

    def copy(self):
        o = RtRenderParams()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(RtRenderParams))
        return o


class RtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("kernel_ms", C.c_double), ("traverse_ms", C.c_double), ("total_ms", C.c_double),
                ("nodes_fetched", C.c_uint64), ("rays_coincident", C.c_uint64)]


class RtBuildInfo(C.Structure):
    _fields_ = [("root_ref", C.c_int32), ("builder", C.c_int32), ("order_on_device", C.c_int32), ("reserved", C.c_int32),
                ("pad", C.c_double), ("device_build_ms", C.c_double), ("device_order_ms", C.c_double)]


class RtBvhInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_prims", C.c_int32), ("max_depth", C.c_int32), ("node_bytes", C.c_int32),
                ("root_min", C.c_double * 3), ("root_max", C.c_double * 3)]


# numpy view of RtShape for building big synthetic scenes without Python loops
SHAPE_DTYPE = np.dtype([("type", "<i4"), ("material", "<i4"), ("transform", "<f8", (16,)), ("inv_transform", "<f8", (16,)),
                        ("corners", "<f8", (12,)), ("velocity", "<f8", (3,)), ("max_displacement", "<f8")], align=True)
assert SHAPE_DTYPE.itemsize == C.sizeof(RtShape)

# Every symbol include/rt_mi355x.h declares (tests check that the library exports all of them).
ABI_SYMBOLS = [
    "rt_init", "rt_shutdown", "rt_last_error", "rt_scene_create", "rt_scene_destroy", "rt_scene_bvh_info",
    "rt_bvh_build_host", "rt_render", "rt_render_device", "rt_part_rows", "rt_part_scatter", "rt_config_load",
    "rt_config_parse", "rt_config_get_int", "rt_config_get_double", "rt_config_get_bool", "rt_config_destroy",
    "rt_params_from_config", "rt_scene_file_load", "rt_scene_file_parse", "rt_scene_file_desc",
    "rt_scene_file_destroy", "rt_shapes_from_trs", "rt_write_ppm", "rt_read_ppm", "rt_read_pfm", "rt_free",
    "rt_version", "rt_traverse_time", "rt_bvh_device_image", "rt_render_linear", "rt_render_linear_device", "rt_write_pfm",
    "rt_set_bvh_builder", "rt_scene_bvh_nodes", "rt_bvh_order_device", "rt_render_device_noalloc", "rt_render_frames_device", "rt_reserve",
    "rt_release_workspaces", "rt_comm_unique_id", "rt_comm_init", "rt_comm_rank", "rt_comm_world", "rt_comm_destroy",
    "rt_gather_frames", "rt_deinterleave_device", "rt_render_gathered", "rt_accumulate_device", "rt_resolve_sum_device",
    "rt_gather_time", "rt_render_frames",
]


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels ship their own libamdhip64.so (SONAME libamdhip64.so.7, the same as /opt/rocm's).  If this
    library is loaded first it binds /opt/rocm's copy, a later `import torch` maps its own, and a process with two HIP
    runtimes cannot share streams or buffers (torch then reports "No HIP GPUs are available").  Mapping torch's copy
    first makes both use one runtime whichever is imported first.  Without torch installed nothing happens."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("librt_mi355x.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C raytracer_amd/csrc`); raytracer_amd has no fallback path")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, i32, dbl, cp = C.c_void_p, C.c_int, C.c_double, C.c_char_p
    sig = {
        "rt_init": (i32, [i32]), "rt_shutdown": (None, []), "rt_last_error": (cp, []), "rt_version": (cp, []),
        "rt_scene_create": (i32, [C.POINTER(RtSceneDesc), C.POINTER(vp)]), "rt_scene_destroy": (None, [vp]),
        "rt_scene_bvh_info": (i32, [vp, C.POINTER(RtBvhInfo)]),
        "rt_bvh_build_host": (i32, [C.POINTER(RtSceneDesc), C.POINTER(RtBvhInfo), C.POINTER(C.c_int32)]),
        "rt_traverse_time": (i32, [vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "rt_render_linear": (i32, [vp, C.POINTER(RtRenderParams), vp]),
        "rt_render_linear_device": (i32, [vp, C.POINTER(RtRenderParams), vp, vp]),
        "rt_write_pfm": (i32, [cp, i32, i32, vp]),
        "rt_bvh_device_image": (i32, [C.POINTER(RtSceneDesc), vp, C.c_int32, C.POINTER(RtBvhInfo), C.POINTER(C.c_int32),
                                C.POINTER(C.c_double), vp]),
        "rt_set_bvh_builder": (i32, [i32]),
        "rt_scene_bvh_nodes": (i32, [vp, vp, C.c_int32, C.POINTER(RtBuildInfo)]),
        "rt_bvh_order_device": (i32, [C.POINTER(RtSceneDesc), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
        "rt_render": (i32, [vp, C.POINTER(RtRenderParams), vp, C.POINTER(RtStats)]),
        "rt_render_device": (i32, [vp, C.POINTER(RtRenderParams), vp, vp, i32, vp]),
        "rt_render_frames": (i32, [vp, C.POINTER(RtRenderParams), i32, vp, C.POINTER(RtStats)]),
        "rt_render_device_noalloc": (i32, [vp, C.POINTER(RtRenderParams), vp, vp]),
        "rt_render_frames_device": (i32, [vp, C.POINTER(RtRenderParams), i32, i32, vp, vp, vp]),
        "rt_reserve": (i32, [vp, C.c_uint64, i32]), "rt_release_workspaces": (i32, []),
        "rt_comm_unique_id": (i32, [vp]), "rt_comm_init": (i32, [i32, i32, vp]), "rt_comm_rank": (i32, []),
        "rt_comm_world": (i32, []), "rt_comm_destroy": (i32, []),
        "rt_gather_frames": (i32, [vp, i32, i32, i32, i32, vp, vp]),
        "rt_gather_time": (i32, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "rt_render_gathered": (i32, [vp, C.POINTER(RtRenderParams), vp, C.POINTER(RtStats)]),
        "rt_accumulate_device": (i32, [vp, C.POINTER(RtRenderParams), vp, C.c_uint64, i32, vp]),
        "rt_resolve_sum_device": (i32, [vp, C.c_uint64, i32, i32, i32, vp, vp]),
        "rt_deinterleave_device": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "rt_part_rows": (i32, [i32, i32, i32, i32]),
        "rt_part_scatter": (i32, [vp, vp, i32, i32, i32, i32, i32]),
        "rt_config_load": (i32, [cp, C.POINTER(vp)]), "rt_config_parse": (i32, [cp, C.POINTER(vp)]),
        "rt_config_get_int": (i32, [vp, cp, i32]), "rt_config_get_double": (dbl, [vp, cp, dbl]),
        "rt_config_get_bool": (i32, [vp, cp, i32]), "rt_config_destroy": (None, [vp]),
        "rt_params_from_config": (i32, [vp, C.POINTER(RtRenderParams)]),
        "rt_scene_file_load": (i32, [cp, cp, vp, dbl, C.POINTER(vp)]),
        "rt_scene_file_parse": (i32, [cp, cp, vp, dbl, C.POINTER(vp)]),
        "rt_scene_file_desc": (C.POINTER(RtSceneDesc), [vp]), "rt_scene_file_destroy": (None, [vp]),
        "rt_shapes_from_trs": (i32, [i32, vp, vp]),
        "rt_write_ppm": (i32, [cp, i32, i32, vp]),
        "rt_read_ppm": (i32, [cp, C.POINTER(i32), C.POINTER(i32), C.POINTER(vp)]),
        "rt_read_pfm": (i32, [cp, C.POINTER(i32), C.POINTER(i32), C.POINTER(vp)]),
        "rt_free": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def _check(rc):
    if rc != 0:
        raise RtError(rc, lib.rt_last_error().decode("utf-8", "replace"))


def version():
    return lib.rt_version().decode()


def init(device_id=-1):
    """Select the GPU of this process (one process per GPU).  Raises RtError(RT_ERR_DEVICE) without a GPU."""
    _check(lib.rt_init(device_id))


def shutdown():
    lib.rt_shutdown()


def reserve(stream, rays_per_queue, n_queues=2):
    """Allocate and first-touch the ray queues of `stream` ahead of the first frame (see rt_reserve)."""
    _check(lib.rt_reserve(stream, int(rays_per_queue), int(n_queues)))


def release_workspaces():
    _check(lib.rt_release_workspaces())


# ---------------------------------------------------------------------------------------- several GPUs (RCCL inside the library)
RT_COMM_ID_BYTES = 128


def comm_unique_id():
    """Rank 0: the bytes every rank passes to comm_init (hand them over with the launcher's own means)."""
    buf = C.create_string_buffer(RT_COMM_ID_BYTES)
    _check(lib.rt_comm_unique_id(buf))
    return buf.raw


def comm_init(rank, world, unique_id):
    _check(lib.rt_comm_init(int(rank), int(world), C.create_string_buffer(bytes(unique_id), RT_COMM_ID_BYTES)))


def comm_rank():
    return lib.rt_comm_rank()


def comm_world():
    return lib.rt_comm_world()


def comm_destroy():
    _check(lib.rt_comm_destroy())


def gather_frames(d_slices_ptr, frames, width, height, tile_rows, d_frames_ptr, stream=None):
    """Collective: slices of `frames` frames to rank 0 over RCCL + de-interleave kernel (see rt_gather_frames)."""
    _check(lib.rt_gather_frames(d_slices_ptr, int(frames), int(width), int(height), int(tile_rows), d_frames_ptr, stream))


def gather_time():
    """(exchange ms, de-interleave ms, gathers) of the gathers since the last call (their streams synchronised): rt_gather_time."""
    ex, de, n = C.c_double(0.0), C.c_double(0.0), C.c_uint64(0)
    _check(lib.rt_gather_time(C.byref(ex), C.byref(de), C.byref(n)))
    return ex.value, de.value, n.value


def deinterleave_device(d_own_ptr, d_gathered_ptr, d_frames_ptr, frames, width, height, tile_rows, world, stream=None):
    _check(lib.rt_deinterleave_device(d_own_ptr, d_gathered_ptr, d_frames_ptr, int(frames), int(width), int(height), int(tile_rows),
                                      int(world), stream))


BVH_BUILDERS = {"default": -1, "sah": 0, "lbvh": 1, "median": 2, "sah_gpu": 3}   # RT_BVH_* of rt_mi355x.h


def set_bvh_builder(name):
    """Builder of the hierarchy the device traverses, for scenes created afterwards: "sah" (host threads), "sah_gpu" (the same
    tree built on the GPU), "lbvh" (Morton-order tree built on the GPU), "median" (the reference's topology), "default"
    (environment RT_BVH_BUILDER, else the SAH tree: on the GPU from 65536 shapes, on host threads below)."""
    _check(lib.rt_set_bvh_builder(BVH_BUILDERS[name]))


# ---------------------------------------------------------------------------------------- Config (config.cpp)
class Config:
    """`Config::Instance()` of the reference as an object: same file grammar, same typed getters."""

    def __init__(self, path=None, text=None):
        self._h = C.c_void_p()
        if path is not None:
            rc = lib.rt_config_load(os.fsencode(path), C.byref(self._h))
            self.loaded = rc == 0  # a missing file only warns in the reference (config.cpp:9-12)
        else:
            _check(lib.rt_config_parse((text or "").encode(), C.byref(self._h)))
            self.loaded = True

    def get_int(self, key, default):
        return lib.rt_config_get_int(self._h, key.encode(), default)

    def get_double(self, key, default):
        return lib.rt_config_get_double(self._h, key.encode(), default)

    def get_bool(self, key, default):
        return bool(lib.rt_config_get_bool(self._h, key.encode(), int(default)))

    def params(self, **flags):
        """RtRenderParams for this config plus command-line style overrides (see apply_flags)."""
        p = RtRenderParams()
        _check(lib.rt_params_from_config(self._h, C.byref(p)))
        return apply_flags(p, self, **flags)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:   # (lib is None while the interpreter shuts down)
            lib.rt_config_destroy(self._h)
            self._h = None


def apply_flags(p, cfg=None, aa=None, exposure=None, shadows=None, glossy=None, fresnel=None, normals=None, no_bvh=None,
                tonemap=None, rng=None, seed=None):
    """The reference's CLI flags (main.cpp:82-216) applied to render params.  `--motion-blur` is a scene argument
    (shutter time) and is passed to SceneFile instead."""
    if aa is not None:
        p.samples_per_pixel = int(aa)
    if exposure is not None:
        p.exposure = float(exposure)
    if shadows is not None:
        p.shadows = int(bool(shadows))
    if glossy:
        p.glossy_samples = cfg.get_int("render.glossy_samples", 8) if cfg is not None else 8  # main.cpp:152
    if fresnel is not None:
        p.fresnel = int(bool(fresnel))
    if normals is not None:
        p.normals = int(bool(normals))
    if no_bvh is not None:
        p.use_bvh = 0 if no_bvh else 1
    if tonemap is not None:
        p.tonemap = TONEMAP_BY_NAME.get(str(tonemap).lower(), 0) if not isinstance(tonemap, int) else tonemap
    if rng is not None:
        p.rng_mode = {"d0": RT_RNG_D0, "seeded": RT_RNG_SEEDED}[rng] if isinstance(rng, str) else int(rng)
    if seed is not None:
        p.seed = int(seed)
    return p


# ---------------------------------------------------------------------------------------- scene description
class SceneFile:
    """A parsed Blender-ASCII scene (scene.cpp:124-546) held on the host as an RtSceneDesc."""

    def __init__(self, path=None, text=None, texture_root="", config=None, shutter_time=None):
        self._h = C.c_void_p()
        self._cfg = config
        if shutter_time is None:  # main.cpp:62: image.shutter_time unless --motion-blur overrides it
            shutter_time = config.get_double("image.shutter_time", 0.0) if config is not None else 0.0
        ch = config._h if config is not None else None
        if path is not None:
            _check(lib.rt_scene_file_load(os.fsencode(path), os.fsencode(texture_root), ch, float(shutter_time), C.byref(self._h)))
        else:
            _check(lib.rt_scene_file_parse(text.encode(), os.fsencode(texture_root), ch, float(shutter_time), C.byref(self._h)))

    @property
    def desc(self):
        return lib.rt_scene_file_desc(self._h).contents

    def desc_ptr(self):
        return lib.rt_scene_file_desc(self._h)

    @property
    def width(self):
        return self.desc.camera.resolution_x

    @property
    def height(self):
        return self.desc.camera.resolution_y

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.rt_scene_file_destroy(self._h)
            self._h = None


class SceneBuilder:
    """Programmatic RtSceneDesc (keeps the numpy / ctypes buffers alive).  Used for synthetic scenes such as the
    10^6-sphere BVH stress of BASELINE config 2."""

    def __init__(self, camera, lights, shapes, materials, textures=(), env=None, shutter_time=0.0):
        self.shapes = np.ascontiguousarray(shapes, dtype=SHAPE_DTYPE)
        self._lights = (RtLight * max(1, len(lights)))(*lights)
        self._materials = (RtMaterial * max(1, len(materials)))(*materials)
        self._tex_arrays = [np.ascontiguousarray(t, dtype=np.uint8) for t in textures]
        self._textures = (RtTexture * max(1, len(textures)))()
        for i, t in enumerate(self._tex_arrays):
            self._textures[i].width = t.shape[1]
            self._textures[i].height = t.shape[0]
            self._textures[i].rgb = t.ctypes.data_as(C.POINTER(C.c_uint8))
        d = RtSceneDesc()
        d.camera = camera
        d.n_lights, d.n_shapes, d.n_materials, d.n_textures = len(lights), len(self.shapes), len(materials), len(textures)
        d.lights = C.cast(self._lights, C.POINTER(RtLight))
        d.shapes = C.cast(self.shapes.ctypes.data, C.POINTER(RtShape))
        d.materials = C.cast(self._materials, C.POINTER(RtMaterial))
        d.textures = C.cast(self._textures, C.POINTER(RtTexture))
        if env is not None:
            self._env = np.ascontiguousarray(env, dtype=np.float32)
            d.env.present = 1
            d.env.height, d.env.width = self._env.shape[0], self._env.shape[1]
            d.env.rgb = self._env.ctypes.data_as(C.POINTER(C.c_float))
        d.shutter_time = shutter_time
        self.desc = d

    def desc_ptr(self):
        return C.pointer(self.desc)

    @property
    def width(self):
        return self.desc.camera.resolution_x

    @property
    def height(self):
        return self.desc.camera.resolution_y


def shapes_from_trs(trs, shape_type=RT_SHAPE_SPHERE, material=0):
    """n x 9 array (translation, rotation_euler_radians, scale) -> SHAPE_DTYPE array with the reference's matrices."""
    trs = np.ascontiguousarray(trs, dtype=np.float64).reshape(-1, 9)
    out = np.zeros(len(trs), dtype=SHAPE_DTYPE)
    out["type"] = shape_type
    out["material"] = material
    _check(lib.rt_shapes_from_trs(len(trs), trs.ctypes.data, out.ctypes.data))
    return out


def default_material():
    """material.h:51"""
    m = RtMaterial()
    m.ambient[:] = [0.1] * 3
    m.diffuse[:] = [0.7] * 3
    m.specular[:] = [1.0] * 3
    m.shininess, m.reflectivity, m.transparency, m.refractive_index = 32.0, 0.0, 0.0, 1.0
    m.texture, m.bump_map, m.is_metal = -1, -1, 0
    return m


def bvh_build_host(scene_desc):
    """Host-only BVH build: (RtBvhInfo, depth-first shape order).  Needs no GPU."""
    d = scene_desc.desc_ptr()
    info = RtBvhInfo()
    order = np.zeros(max(1, d.contents.n_shapes), dtype=np.int32)
    _check(lib.rt_bvh_build_host(d, C.byref(info), order.ctypes.data_as(C.POINTER(C.c_int32))))
    return info, order[: d.contents.n_shapes]


def bvh_order_device(scene_desc):
    """The reference's depth-first visit order computed on the GPU: (order, device ms).  Raises RtError(RT_ERR_LIMIT) when
    equal box centres meet at a split (the host pass then decides, see rt_bvh_order_device)."""
    d = scene_desc.desc_ptr()
    order = np.zeros(max(1, d.contents.n_shapes), dtype=np.int32)
    ms = C.c_double(0.0)
    _check(lib.rt_bvh_order_device(d, order.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ms)))
    return order[: d.contents.n_shapes], ms.value


NODE4_DTYPE = np.dtype([("org", "<f4", 3), ("step_x", "<f4"), ("qlo", "<u4", 3), ("qhi", "<u4", 3), ("child", "<i4", 4),
                        ("step_yz", "<f4", 2)])   # device_types.h: DNode4 (64 bytes)


def node4_steps(node):
    """Grid steps (x, y, z) of one NODE4_DTYPE record: powers of two, stored as FP32 numbers."""
    return (float(node["step_x"]), float(node["step_yz"][0]), float(node["step_yz"][1]))
EMPTY_CHILD = 0x7FFFFFFF


def bvh_device_image(scene_desc):
    """The hierarchy as the device traverses it, built on the host (no GPU): dict with `nodes` (structured array of
    NODE4_DTYPE), `info` (RtBvhInfo of the 4-wide tree), `root_ref`, `pad` and `prim_boxes` [n, 6] in primitive order."""
    d = scene_desc.desc_ptr()
    n = d.contents.n_shapes
    nodes = np.zeros(max(1, n), dtype=NODE4_DTYPE)
    boxes = np.zeros((max(1, n), 6), dtype=np.float64)
    info, root, pad = RtBvhInfo(), C.c_int32(0), C.c_double(0.0)
    _check(lib.rt_bvh_device_image(d, nodes.ctypes.data, len(nodes), C.byref(info), C.byref(root), C.byref(pad), boxes.ctypes.data))
    return {"nodes": nodes[: info.n_nodes], "info": info, "root_ref": root.value, "pad": pad.value, "prim_boxes": boxes[:n]}


def part_rows(height, tile_rows, part_index, part_count):
    return lib.rt_part_rows(height, tile_rows, part_index, part_count)


def part_scatter(slice_u8, frame_u8, tile_rows, part_index, part_count):
    h, w = frame_u8.shape[0], frame_u8.shape[1]
    s = np.ascontiguousarray(slice_u8)
    _check(lib.rt_part_scatter(s.ctypes.data, frame_u8.ctypes.data, w, h, tile_rows, part_index, part_count))


# ---------------------------------------------------------------------------------------- device scene + render
class Scene:
    """The device-resident scene (`Scene::Scene` + BVH wrap, scene.cpp:93-122) on the GPU chosen by init()."""

    def __init__(self, scene_desc):
        self._h = C.c_void_p()
        self.width, self.height = scene_desc.width, scene_desc.height
        _check(lib.rt_scene_create(scene_desc.desc_ptr(), C.byref(self._h)))

    def bvh_info(self):
        info = RtBvhInfo()
        _check(lib.rt_scene_bvh_info(self._h, C.byref(info)))
        return info

    def bvh_nodes(self):
        """The hierarchy as it lies in device memory: dict with `nodes` (NODE4_DTYPE), `root_ref`, `pad`, `builder` (name of
        the builder that actually built it), `device_build_ms` (lbvh only), `order_on_device` and `device_order_ms`."""
        info = self.bvh_info()
        nodes = np.zeros(max(1, info.n_nodes), dtype=NODE4_DTYPE)
        b = RtBuildInfo()
        _check(lib.rt_scene_bvh_nodes(self._h, nodes.ctypes.data, len(nodes), C.byref(b)))
        names = {v: k for k, v in BVH_BUILDERS.items()}
        return {"nodes": nodes[: info.n_nodes], "info": info, "root_ref": b.root_ref, "pad": b.pad, "builder": names[b.builder],
                "device_build_ms": b.device_build_ms, "order_on_device": bool(b.order_on_device), "device_order_ms": b.device_order_ms}

    def render(self, params, stats=False, out=None):
        """One frame -> uint8 [H, W, 3] (the bytes Image::write would store).  stats=True also returns RtStats.  `out`: a
        buffer to reuse, as a render loop would (the library keeps it registered with the driver between calls)."""
        img = np.zeros((self.height, self.width, 3), dtype=np.uint8) if out is None else out
        st = RtStats()
        _check(lib.rt_render(self._h, C.byref(params), img.ctypes.data, C.byref(st) if stats else None))
        return (img, st) if stats else img

    def render_frames(self, params, frames, out=None, stats=False):
        """`frames` whole frames (seeds params.seed + f) -> uint8 [frames, H, W, 3] in host memory (rt_render_frames); `out`: a
        buffer to reuse (the library keeps it registered with the driver between calls)."""
        if out is None:
            out = np.zeros((frames, self.height, self.width, 3), dtype=np.uint8)
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.shape == (frames, self.height, self.width, 3)
        st = RtStats()
        _check(lib.rt_render_frames(self._h, C.byref(params), int(frames), out.ctypes.data, C.byref(st) if stats else None))
        return (out, st) if stats else out

    def render_gathered(self, params, stats=False):
        """Collective over the library's communicator: every rank renders its tiles, rank 0 returns the whole frame (the
        other ranks None).  See rt_render_gathered."""
        root = comm_rank() == 0
        img = np.zeros((self.height, self.width, 3), dtype=np.uint8) if root else None
        st = RtStats()
        _check(lib.rt_render_gathered(self._h, C.byref(params), img.ctypes.data if root else None, C.byref(st) if stats else None))
        return (img, st) if stats else img

    def render_linear(self, params):
        """One frame of linear radiance (before tone mapping): float64 [H, W, 3] (see rt_render_linear)."""
        img = np.zeros((self.height, self.width, 3), dtype=np.float64)
        _check(lib.rt_render_linear(self._h, C.byref(params), img.ctypes.data))
        return img

    def render_linear_device(self, params, d_rgb_f64_ptr, stream=None):
        _check(lib.rt_render_linear_device(self._h, C.byref(params), d_rgb_f64_ptr, stream))

    def render_device(self, params, d_rgb8_ptr, d_counters_ptr=None, collect_counters=False, stream=None):
        """Asynchronous render of this part's rows into device memory (see rt_render_device).  stream: a hipStream_t value
        (torch: `stream.cuda_stream`); None is the null stream, i.e. torch's default stream."""
        _check(lib.rt_render_device(self._h, C.byref(params), d_rgb8_ptr, d_counters_ptr, int(collect_counters), stream))

    def accumulate_device(self, params, d_sum_ptr, first_pass, passes, stream=None):
        """Add `passes` passes (seeds params.seed + first_pass ...) to the FP64 sum in device memory (rt_accumulate_device)."""
        _check(lib.rt_accumulate_device(self._h, C.byref(params), d_sum_ptr, int(first_pass), int(passes), stream))

    def render_frames_device(self, params, frames, d_rgb8_ptr, slice_rows=0, d_counters_ptr=None, stream=None):
        """`frames` frames (seeds params.seed + f) of this part in one set of launches; slices `slice_rows` rows apart
        (0 = rt_part_rows) in device memory (see rt_render_frames_device)."""
        _check(lib.rt_render_frames_device(self._h, C.byref(params), int(frames), int(slice_rows), d_rgb8_ptr, d_counters_ptr, stream))

    def render_device_noalloc(self, params, d_rgb8_ptr, stream=None):
        """rt_render_device through the library's static workspace: no device allocation (see rt_render_device_noalloc)."""
        _check(lib.rt_render_device_noalloc(self._h, C.byref(params), d_rgb8_ptr, stream))

    def traverse_time(self, stream=None):
        """(ms, launches) of the traversal-kernel launches enqueued on `stream` since the last call (stream synchronised)."""
        ms, n = C.c_double(0.0), C.c_uint64(0)
        _check(lib.rt_traverse_time(self._h, stream, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def resolve_sum_device(d_sum_ptr, passes, width, rows, tonemap, d_rgb8_ptr, stream=None):
    _check(lib.rt_resolve_sum_device(d_sum_ptr, int(passes), int(width), int(rows), int(tonemap), d_rgb8_ptr, stream))


def write_pfm(path, img):
    img = np.ascontiguousarray(img, dtype=np.float64)
    _check(lib.rt_write_pfm(os.fsencode(path), img.shape[1], img.shape[0], img.ctypes.data))


def write_ppm(path, img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    _check(lib.rt_write_ppm(os.fsencode(path), img.shape[1], img.shape[0], img.ctypes.data))


def read_ppm(path):
    w, h, p = C.c_int(), C.c_int(), C.c_void_p()
    _check(lib.rt_read_ppm(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p)))
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 3)).copy()
    finally:
        lib.rt_free(p)


def read_pfm(path):
    w, h, p = C.c_int(), C.c_int(), C.c_void_p()
    _check(lib.rt_read_pfm(os.fsencode(path), C.byref(w), C.byref(h), C.byref(p)))
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(h.value, w.value, 3)).copy()
    finally:
        lib.rt_free(p)
