"""ctypes mirror of include/raytracer_amd.h and the loader of the HIP library.

The product path is `librt_amd.so` (raytracer_amd/csrc, hand-written HIP for gfx950). There is no CPU
fallback: if the library is missing, `load()` raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB: another build of the library (A/B experiments: tools/exp_ab_builds.sh); unset = the in-tree build
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(HERE, "csrc", "librt_amd.so")

RT_SCENE_BVH_DEVICE = 1
RT_SCENE_NO_PACKETS = 2
RT_OBJ_SPHERE, RT_OBJ_PLANE, RT_OBJ_MODEL = 0, 1, 2
RT_TRAVERSE_QUAD, RT_TRAVERSE_PACKET, RT_TRAVERSE_SERVE = 0, 1, 2


class rt_material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("ka", C.c_float), ("kd", C.c_float), ("ks", C.c_float),
                ("kr", C.c_float), ("kt", C.c_float), ("specRefExp", C.c_float),
                ("refractiveIdx", C.c_float), ("reflective", C.c_int32), ("refractive", C.c_int32)]


class rt_light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("color", C.c_float * 3)]


class rt_sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float)]


class rt_plane(C.Structure):
    _fields_ = [("a", C.c_float), ("b", C.c_float), ("c", C.c_float), ("d", C.c_float)]


class rt_model(C.Structure):
    _fields_ = [("first_face", C.c_uint32), ("num_faces", C.c_uint32), ("bb_center", C.c_float * 3),
                ("bb_radius", C.c_float)]


class rt_object(C.Structure):
    _fields_ = [("type", C.c_int32), ("index", C.c_uint32), ("material", C.c_uint32),
                ("reserved", C.c_uint32)]


FP = C.POINTER(C.c_float)


class rt_flat_scene(C.Structure):
    _fields_ = [("objects", C.POINTER(rt_object)), ("n_objects", C.c_uint32),
                ("spheres", C.POINTER(rt_sphere)), ("n_spheres", C.c_uint32),
                ("planes", C.POINTER(rt_plane)), ("n_planes", C.c_uint32),
                ("models", C.POINTER(rt_model)), ("n_models", C.c_uint32),
                ("materials", C.POINTER(rt_material)), ("n_materials", C.c_uint32),
                ("lights", C.POINTER(rt_light)), ("n_lights", C.c_uint32),
                ("v0", FP), ("v1", FP), ("v2", FP), ("n0", FP), ("n1", FP), ("n2", FP),
                ("n_faces", C.c_uint32)]


class rt_camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("at", C.c_float * 3), ("aspectRatio", C.c_float), ("fov", C.c_float),
                ("imWidth", C.c_uint32), ("imHeight", C.c_uint32)]


class rt_params(C.Structure):
    _fields_ = [("background", C.c_float * 3), ("max_depth", C.c_int32), ("n_aa", C.c_uint32),
                ("aa_xy", FP)]


class rt_stats(C.Structure):
    _fields_ = [("trace_calls", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("level_rays", C.c_uint64 * 16), ("passes", C.c_uint32), ("retries", C.c_uint32),
                ("kernel_ms", C.c_float), ("kernel_ms_by", C.c_float * 5),
                ("kernel_launches_by", C.c_uint32 * 5), ("reserved", C.c_uint32)]


class rt_bvh_info(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_faces", C.c_uint32), ("max_depth", C.c_uint32),
                ("node_bytes", C.c_uint32), ("tri_bytes", C.c_uint32), ("built_on_device", C.c_uint32),
                ("device_bytes", C.c_uint64), ("build_seconds", C.c_double)]


class rt_bvh_check_result(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("n_faces", C.c_uint32),
                ("max_depth", C.c_uint32), ("max_stack", C.c_uint32), ("reserved", C.c_uint32),
                ("errors", C.c_uint64), ("host_bytes", C.c_uint64), ("build_seconds", C.c_double)]


class rt_work(C.Structure):
    _fields_ = [("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("spheres_tested", C.c_uint64), ("planes_tested", C.c_uint64),
                ("wave_node_steps", C.c_uint64), ("wave_tri_steps", C.c_uint64),
                ("trav_ticks_sum", C.c_uint64), ("trav_ticks_max", C.c_uint64), ("batches", C.c_uint64),
                ("trav_waves", C.c_uint64), ("wave_ticks_sum", C.c_uint64), ("wave_ticks_max", C.c_uint64),
                ("wave_cycles_sum", C.c_uint64)]


def fptr(a):
    return a.ctypes.data_as(FP)


def u8ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


# Every symbol include/raytracer_amd.h declares; tests check the library exports all of them.
EXPORTS = [
    "rt_scene_create", "rt_scene_create_ex", "rt_scene_destroy", "rt_render", "rt_render_device", "rt_render_tiles_device",
    "rt_render_batch_device", "rt_render_tiles_batch_device", "rt_render_device_noalloc", "rt_render_batch_host_async", "rt_pinned_alloc", "rt_pinned_free",
    "rt_tiles_owned", "rt_untile_device", "rt_untile_batch_device", "rt_cast_rays", "rt_intersect", "rt_occluded", "rt_trace",
    "rt_intersect_form", "rt_occluded_form", "rt_trace_form", "rt_scene_set_pass_samples",
    "rt_scene_bvh_info", "rt_bvh_check", "rt_scene_bvh_check", "rt_scene_status", "rt_scene_stream", "rt_set_count_work", "rt_get_work", "rt_last_error", "rt_version",
    "rt_render_progress", "rt_multi_create", "rt_multi_destroy", "rt_multi_parts", "rt_multi_scene", "rt_multi_render", "rt_multi_render_device",
    "rt_multi_status", "rt_scene_pass_samples", "rt_scene_set_option",
]

PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64)

HOST_EXPORTS = [
    "rt_host_camera_make", "rt_host_plane_make", "rt_host_mat3_rotation", "rt_host_bbvol", "rt_host_model_load_ply",
    "rt_host_model_from_arrays", "rt_host_model_transform", "rt_host_model_num_faces", "rt_host_model_get", "rt_host_model_free",
    "rt_host_known_answers", "rt_host_scene_load_yaml", "rt_host_scene_flat", "rt_host_scene_free", "rt_host_write_image",
    "rt_host_primitive_intersect",
]

_lib = None


def load(path=None):
    """Load the HIP library. Raises if it is not built: the product path has no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). raytracer_amd has no CPU fallback.")
    lib = C.CDLL(p)
    vp = C.c_void_p
    u32 = C.c_uint32
    lib.rt_scene_create.argtypes = [C.POINTER(rt_flat_scene), C.c_int, C.POINTER(vp)]
    lib.rt_scene_create_ex.argtypes = [C.POINTER(rt_flat_scene), C.c_int, C.c_uint32, C.POINTER(vp)]
    lib.rt_scene_destroy.argtypes = [vp]
    lib.rt_scene_destroy.restype = None
    lib.rt_render.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), u32, u32, u32, u32,
                              C.POINTER(C.c_uint8), C.POINTER(rt_stats)]
    lib.rt_render_device.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), u32, u32, u32, u32,
                                     vp, vp, C.POINTER(rt_stats)]
    lib.rt_render_device_noalloc.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), u32, u32, u32, u32, vp, vp]
    lib.rt_render_batch_host_async.argtypes = [vp, C.POINTER(rt_camera), u32, C.POINTER(rt_params), u32, u32, u32, u32,
                                               vp, C.c_uint64, vp]
    lib.rt_pinned_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.rt_pinned_free.argtypes = [vp]
    lib.rt_pinned_free.restype = None
    lib.rt_render_tiles_device.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), u32, u32, u32,
                                           u32, vp, vp, C.POINTER(rt_stats)]
    lib.rt_render_batch_device.argtypes = [vp, C.POINTER(rt_camera), u32, C.POINTER(rt_params), u32, u32, u32, u32,
                                           vp, C.c_uint64, vp, C.POINTER(rt_stats)]
    lib.rt_render_tiles_batch_device.argtypes = [vp, C.POINTER(rt_camera), u32, C.POINTER(rt_params), u32, u32, u32,
                                                 u32, vp, C.c_uint64, vp, C.POINTER(rt_stats)]
    lib.rt_tiles_owned.argtypes = [u32] * 6
    lib.rt_tiles_owned.restype = u32
    lib.rt_untile_device.argtypes = [C.c_int, u32, u32, u32, u32, u32, u32, vp, vp, vp]
    lib.rt_untile_batch_device.argtypes = [C.c_int, u32, u32, u32, u32, u32, u32, u32, C.c_uint64, C.c_uint64, vp, vp, vp]
    lib.rt_cast_rays.argtypes = [C.c_int, C.POINTER(rt_camera), C.POINTER(rt_params),
                                 C.POINTER(C.c_uint32), u32, FP]
    lib.rt_intersect.argtypes = [vp, C.c_int, FP, u32, C.POINTER(C.c_int32), FP, FP]
    lib.rt_occluded.argtypes = [vp, FP, FP, u32, C.POINTER(C.c_uint8)]
    lib.rt_trace.argtypes = [vp, C.POINTER(rt_params), FP, u32, FP, C.POINTER(rt_stats)]
    lib.rt_intersect_form.argtypes = [vp, C.c_int, C.c_int, FP, u32, C.POINTER(C.c_int32), FP, FP]
    lib.rt_occluded_form.argtypes = [vp, C.c_int, FP, FP, u32, C.POINTER(C.c_uint8)]
    lib.rt_trace_form.argtypes = [vp, C.c_int, C.POINTER(rt_params), FP, u32, FP, C.POINTER(rt_stats)]
    lib.rt_scene_set_pass_samples.argtypes = [vp, u32]
    lib.rt_scene_pass_samples.argtypes = [vp]
    lib.rt_scene_pass_samples.restype = u32
    if hasattr(lib, "rt_scene_set_option"):      # (absent from round-2 builds of the library: tools/exp_ab_builds.sh A/B runs)
        lib.rt_scene_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.rt_scene_bvh_info.argtypes = [vp, C.POINTER(rt_bvh_info)]
    lib.rt_bvh_check.argtypes = [C.POINTER(rt_flat_scene), C.POINTER(rt_bvh_check_result)]
    lib.rt_scene_bvh_check.argtypes = [vp, C.POINTER(rt_flat_scene), C.POINTER(rt_bvh_check_result)]
    lib.rt_scene_status.argtypes = [vp]
    lib.rt_scene_stream.argtypes = [vp]
    lib.rt_scene_stream.restype = C.c_void_p
    lib.rt_set_count_work.argtypes = [vp, C.c_int]
    lib.rt_get_work.argtypes = [vp, C.c_int, C.POINTER(rt_work)]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_version.restype = C.c_char_p
    lib.rt_render_progress.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), u32, u32, u32, u32,
                                       C.POINTER(C.c_uint8), C.POINTER(rt_stats), PROGRESS_FN, vp]
    lib.rt_multi_create.argtypes = [C.POINTER(rt_flat_scene), C.POINTER(C.c_int), u32, u32, C.POINTER(vp)]
    lib.rt_multi_destroy.argtypes = [vp]
    lib.rt_multi_destroy.restype = None
    lib.rt_multi_parts.argtypes = [vp]
    lib.rt_multi_parts.restype = u32
    lib.rt_multi_scene.argtypes = [vp, u32]
    lib.rt_multi_scene.restype = C.c_void_p
    lib.rt_multi_render.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), C.POINTER(C.c_uint8), C.POINTER(rt_stats)]
    lib.rt_multi_render_device.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), vp, C.POINTER(rt_stats)]
    lib.rt_multi_status.argtypes = [vp]
    if path is None:
        _lib = lib
    return lib


class RtError(RuntimeError):
    pass


def check(lib, rc):
    if rc != 0:
        raise RtError(f"rt error {rc}: {lib.rt_last_error().decode()}")


def as_f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a
