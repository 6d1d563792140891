"""Python view of the host surface: build a flat scene, a camera and render parameters, and drive the C-ABI.

All host arithmetic that must match the reference bit for bit (plane normalisation, camera construction,
mesh loading / transform / bounding sphere) is done by the C++ host surface (raytracer_amd/host/raytracer.hpp)
through the `rt_host_*` entry points; this module only moves arrays around.
"""
import ctypes as C

import numpy as np

from . import cabi
from .cabi import (RT_OBJ_MODEL, RT_OBJ_PLANE, RT_OBJ_SPHERE, fptr, rt_camera, rt_flat_scene, rt_light,
                   rt_material, rt_model, rt_object, rt_params, rt_plane, rt_sphere, rt_stats)

_host_bound = False


def host_lib():
    """The library with the rt_host_* prototypes attached (host-only calls: no GPU needed)."""
    global _host_bound
    lib = cabi.load()
    if not _host_bound:
        FP = cabi.FP
        lib.rt_host_camera_make.argtypes = [FP, FP, FP, C.c_float, C.c_float, C.c_uint32, C.c_uint32, FP, FP,
                                            C.POINTER(rt_camera)]
        lib.rt_host_plane_make.argtypes = [FP, C.c_float, C.POINTER(rt_plane)]
        lib.rt_host_mat3_rotation.argtypes = [C.c_float, C.c_float, C.c_float, FP]
        lib.rt_host_bbvol.argtypes = [FP, FP, FP, C.c_uint32, FP]
        lib.rt_host_model_load_ply.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.rt_host_model_from_arrays.argtypes = [FP, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32,
                                                  C.POINTER(C.c_void_p)]
        lib.rt_host_model_transform.argtypes = [C.c_void_p, FP, C.c_double, FP]
        lib.rt_host_model_num_faces.argtypes = [C.c_void_p]
        lib.rt_host_model_num_faces.restype = C.c_uint32
        lib.rt_host_model_get.argtypes = [C.c_void_p, FP, FP, FP, FP, FP, FP, FP]
        lib.rt_host_model_free.argtypes = [C.c_void_p]
        lib.rt_host_model_free.restype = None
        lib.rt_host_known_answers.argtypes = [FP]
        lib.rt_host_scene_load_yaml.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.rt_host_scene_flat.argtypes = [C.c_void_p]
        lib.rt_host_scene_flat.restype = C.POINTER(rt_flat_scene)
        lib.rt_host_scene_free.argtypes = [C.c_void_p]
        lib.rt_host_scene_free.restype = None
        lib.rt_host_write_image.argtypes = [C.c_char_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]
        _host_bound = True
    return lib


def write_image(path, frame):
    """cv::imwrite of the reference's main() (src/raytracer.cpp:720) for an (H,W,3) BGR8 frame: .png or .ppm."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    lib = host_lib()
    cabi.check(lib, lib.rt_host_write_image(str(path).encode(), cabi.u8ptr(frame), frame.shape[1], frame.shape[0]))


def _f3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(3))


def material(color=(0, 0, 0), ka=0.2, kd=0.7, ks=0.2, kr=0.0, kt=0.0, specRefExp=8.0, refractiveIdx=0.0,
             reflective=False, refractive=False):
    """Material with the reference's defaults (src/material.h:41-47). color may be a scalar grey."""
    col = np.broadcast_to(np.asarray(color, dtype=np.float32), (3,))
    m = rt_material()
    m.color[:] = [float(c) for c in col]
    m.ka, m.kd, m.ks, m.kr, m.kt = ka, kd, ks, kr, kt
    m.specRefExp, m.refractiveIdx = specRefExp, refractiveIdx
    m.reflective, m.refractive = int(bool(reflective)), int(bool(refractive))
    return m


def make_camera(width, height, fov=60.0, eye=(0, 0, 0), up=(0, 1, 0), at=(0, 0, -1), aspect=None, rotate=None,
                move=None):
    """Camera::Camera (+ rotate, + move) -> rt_camera (reference src/camera.h:30-37,81-94)."""
    lib = host_lib()
    cam = rt_camera()
    if aspect is None:
        aspect = float(np.float32(width) / np.float32(height))   # (Float) imWidth / imHeight, src/raytracer.cpp:674
    rot = _f3(rotate) if rotate is not None else None
    mv = _f3(move) if move is not None else None
    cabi.check(lib, lib.rt_host_camera_make(fptr(_f3(eye)), fptr(_f3(up)), fptr(_f3(at)), aspect, fov, width, height,
                                            fptr(rot) if rot is not None else None,
                                            fptr(mv) if mv is not None else None, C.byref(cam)))
    return cam


def camera_to_array(cam):
    return np.array([*cam.eye, *cam.up, *cam.right, *cam.at, cam.aspectRatio, cam.fov, cam.imWidth, cam.imHeight],
                    dtype=np.float64)


def camera_from_array(a):
    cam = rt_camera()
    cam.eye[:] = [float(np.float32(x)) for x in a[0:3]]
    cam.up[:] = [float(np.float32(x)) for x in a[3:6]]
    cam.right[:] = [float(np.float32(x)) for x in a[6:9]]
    cam.at[:] = [float(np.float32(x)) for x in a[9:12]]
    cam.aspectRatio, cam.fov = float(np.float32(a[12])), float(np.float32(a[13]))
    cam.imWidth, cam.imHeight = int(a[14]), int(a[15])
    return cam


def reference_camera(width, height):
    """The camera main() builds (reference src/raytracer.cpp:674-689) at another resolution."""
    return make_camera(width, height, fov=60.0, move=(-0.5, 0, -0.25))


class Params:
    """rt_params plus the array it points to."""

    def __init__(self, background=(0, 0.5, 0.5), max_depth=10, aa=()):
        self.aa = np.ascontiguousarray(np.asarray(aa, dtype=np.float32).reshape(-1, 2))
        self.c = rt_params()
        self.c.background[:] = [float(np.float32(b)) for b in background]
        self.c.max_depth = int(max_depth)
        self.c.n_aa = self.aa.shape[0]
        self.c.aa_xy = fptr(self.aa) if self.aa.size else None

    @property
    def spp(self):
        return 1 + self.aa.shape[0]


DEFAULT_AA = [(0.5, 0.5), (0.25, 0.25), (0.75, 0.25), (0.75, 0.75), (0.25, 0.75), (1.0, 0), (1.0, 1.0), (0, 1.0)]


class Mesh:
    """Triangle arrays of one Model: v0,v1,v2,n0,n1,n2 as (F,3) float32 and bbvol (cx,cy,cz,r)."""

    def __init__(self, v0, v1, v2, n0, n1, n2, bb):
        self.v = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3) for a in (v0, v1, v2, n0, n1, n2)]
        self.bb = np.ascontiguousarray(bb, dtype=np.float32).reshape(4)

    @property
    def num_faces(self):
        return self.v[0].shape[0]

    @staticmethod
    def load_ply(path, rotation=(0, 0, 0), scale=1.0, translation=(0, 0, 0), transform=True):
        """Model::load_ply(calcNormal=true) + the loader's transform chain (reference src/primitives.cpp:238-362,
        src/scene.cpp:156-163), executed by the C++ host surface."""
        lib = host_lib()
        h = C.c_void_p()
        cabi.check(lib, lib.rt_host_model_load_ply(str(path).encode(), C.byref(h)))
        return Mesh._finish(lib, h, rotation, scale, translation, transform)

    @staticmethod
    def from_arrays(V, F, rotation=(0, 0, 0), scale=1.0, translation=(0, 0, 0), transform=True):
        """The same as writing V, F to an ASCII .ply and load_ply() (rt_host_model_from_arrays), without the text."""
        lib = host_lib()
        V = np.ascontiguousarray(V, dtype=np.float32).reshape(-1, 3)
        F = np.ascontiguousarray(F, dtype=np.uint32).reshape(-1, 3)
        h = C.c_void_p()
        cabi.check(lib, lib.rt_host_model_from_arrays(fptr(V), V.shape[0], F.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                      F.shape[0], C.byref(h)))
        return Mesh._finish(lib, h, rotation, scale, translation, transform)

    @staticmethod
    def _finish(lib, h, rotation, scale, translation, transform):
        try:
            if transform:
                cabi.check(lib, lib.rt_host_model_transform(h, fptr(_f3(rotation)), float(scale),
                                                            fptr(_f3(translation))))
            n = lib.rt_host_model_num_faces(h)
            arrs = [np.empty((n, 3), np.float32) for _ in range(6)]
            bb = np.empty(4, np.float32)
            cabi.check(lib, lib.rt_host_model_get(h, *[fptr(a) for a in arrs], fptr(bb)))
        finally:
            lib.rt_host_model_free(h)
        return Mesh(*arrs, bb)


class FlatScene:
    """Owns the numpy/ctypes arrays an rt_flat_scene points into."""

    def __init__(self):
        self.objects, self.spheres, self.planes, self.models = [], [], [], []
        self.materials, self.lights, self.meshes = [], [], []
        self._c = None

    def _mat(self, m):
        self.materials.append(m)
        return len(self.materials) - 1

    def add_sphere(self, center, radius, mat):
        s = rt_sphere()
        s.center[:] = [float(np.float32(c)) for c in center]
        s.radius = float(np.float32(radius))
        self.spheres.append(s)
        self.objects.append((RT_OBJ_SPHERE, len(self.spheres) - 1, self._mat(mat)))

    def add_plane(self, normal, distance, mat):
        lib = host_lib()
        p = rt_plane()
        cabi.check(lib, lib.rt_host_plane_make(fptr(_f3(normal)), float(np.float32(distance)), C.byref(p)))
        self.planes.append(p)
        self.objects.append((RT_OBJ_PLANE, len(self.planes) - 1, self._mat(mat)))

    def add_mesh(self, mesh, mat):
        first = sum(m.num_faces for m in self.meshes)
        md = rt_model()
        md.first_face, md.num_faces = first, mesh.num_faces
        md.bb_center[:] = [float(x) for x in mesh.bb[:3]]
        md.bb_radius = float(mesh.bb[3])
        self.meshes.append(mesh)
        self.models.append(md)
        self.objects.append((RT_OBJ_MODEL, len(self.models) - 1, self._mat(mat)))

    def add_light(self, pos, color, factor=1.0):
        """Light{pos, color*factor} (reference src/scene.cpp:117-122)."""
        l = rt_light()
        l.pos[:] = [float(np.float32(c)) for c in pos]
        col = np.broadcast_to(np.asarray(color, dtype=np.float32), (3,)) * np.float32(factor)
        l.color[:] = [float(c) for c in col]
        self.lights.append(l)

    @property
    def num_faces(self):
        return sum(m.num_faces for m in self.meshes)

    def to_arrays(self):
        """Plain numpy arrays (for .npz fixtures): every field of the flat scene, nothing else."""
        mats = np.array([[*m.color, m.ka, m.kd, m.ks, m.kr, m.kt, m.specRefExp, m.refractiveIdx, m.reflective,
                          m.refractive] for m in self.materials], dtype=np.float32).reshape(-1, 12)
        d = {
            "objects": np.array(self.objects, dtype=np.int64).reshape(-1, 3),
            "spheres": np.array([[*s.center, s.radius] for s in self.spheres], dtype=np.float32).reshape(-1, 4),
            "planes": np.array([[p.a, p.b, p.c, p.d] for p in self.planes], dtype=np.float32).reshape(-1, 4),
            "materials": mats,
            "lights": np.array([[*l.pos, *l.color] for l in self.lights], dtype=np.float32).reshape(-1, 6),
            "mesh_faces": np.array([m.num_faces for m in self.meshes], dtype=np.int64),
            "mesh_bb": np.array([m.bb for m in self.meshes], dtype=np.float32).reshape(-1, 4),
        }
        for k, name in enumerate(("v0", "v1", "v2", "n0", "n1", "n2")):
            d[name] = (np.concatenate([m.v[k] for m in self.meshes]) if self.meshes
                       else np.zeros((0, 3), np.float32))
        return d

    @staticmethod
    def from_arrays(d):
        fs = FlatScene()
        mats = []
        for r in d["materials"]:
            m = rt_material()
            m.color[:] = [float(x) for x in r[:3]]
            m.ka, m.kd, m.ks, m.kr, m.kt, m.specRefExp, m.refractiveIdx = [float(x) for x in r[3:10]]
            m.reflective, m.refractive = int(r[10]), int(r[11])
            mats.append(m)
        for r in d["spheres"]:
            s = rt_sphere()
            s.center[:] = [float(x) for x in r[:3]]
            s.radius = float(r[3])
            fs.spheres.append(s)
        for r in d["planes"]:
            p = rt_plane()
            p.a, p.b, p.c, p.d = [float(x) for x in r]
            fs.planes.append(p)
        off = 0
        for nf, bb in zip(d["mesh_faces"], d["mesh_bb"]):
            nf = int(nf)
            mesh = Mesh(*[d[k][off:off + nf] for k in ("v0", "v1", "v2", "n0", "n1", "n2")], bb)
            md = rt_model()
            md.first_face, md.num_faces = off, nf
            md.bb_center[:] = [float(x) for x in bb[:3]]
            md.bb_radius = float(bb[3])
            fs.meshes.append(mesh)
            fs.models.append(md)
            off += nf
        fs.materials = mats
        fs.objects = [(int(t), int(i), int(m)) for t, i, m in d["objects"]]
        for r in d["lights"]:
            l = rt_light()
            l.pos[:] = [float(x) for x in r[:3]]
            l.color[:] = [float(x) for x in r[3:]]
            fs.lights.append(l)
        return fs

    @staticmethod
    def load_yaml(path):
        """Scene::Scene(filename) of the C++ host surface (reference src/scene.cpp:32-189, OpenCV-free): a scene.yaml
        with its meshes loaded and transformed, as a FlatScene in file order."""
        lib = host_lib()
        h = C.c_void_p()
        cabi.check(lib, lib.rt_host_scene_load_yaml(str(path).encode(), C.byref(h)))
        try:
            c = lib.rt_host_scene_flat(h).contents
            nf = c.n_faces

            def tri(ptr):
                return np.ctypeslib.as_array(ptr, shape=(nf * 3,)).reshape(nf, 3).copy() if nf else np.zeros((0, 3), np.float32)
            d = {
                "objects": np.array([[c.objects[i].type, c.objects[i].index, c.objects[i].material]
                                     for i in range(c.n_objects)], dtype=np.int64).reshape(-1, 3),
                "spheres": np.array([[*c.spheres[i].center, c.spheres[i].radius] for i in range(c.n_spheres)],
                                    dtype=np.float32).reshape(-1, 4),
                "planes": np.array([[c.planes[i].a, c.planes[i].b, c.planes[i].c, c.planes[i].d]
                                    for i in range(c.n_planes)], dtype=np.float32).reshape(-1, 4),
                "materials": np.array([[*c.materials[i].color, c.materials[i].ka, c.materials[i].kd, c.materials[i].ks,
                                        c.materials[i].kr, c.materials[i].kt, c.materials[i].specRefExp,
                                        c.materials[i].refractiveIdx, c.materials[i].reflective, c.materials[i].refractive]
                                       for i in range(c.n_materials)], dtype=np.float32).reshape(-1, 12),
                "lights": np.array([[*c.lights[i].pos, *c.lights[i].color] for i in range(c.n_lights)],
                                   dtype=np.float32).reshape(-1, 6),
                "mesh_faces": np.array([c.models[i].num_faces for i in range(c.n_models)], dtype=np.int64),
                "mesh_bb": np.array([[*c.models[i].bb_center, c.models[i].bb_radius] for i in range(c.n_models)],
                                    dtype=np.float32).reshape(-1, 4),
                "v0": tri(c.v0), "v1": tri(c.v1), "v2": tri(c.v2), "n0": tri(c.n0), "n1": tri(c.n1), "n2": tri(c.n2),
            }
        finally:
            lib.rt_host_scene_free(h)
        return FlatScene.from_arrays(d)

    def c_struct(self):
        if self._c is not None:
            return self._c
        fs = rt_flat_scene()
        self._keep = []

        def arr(ctype, items):
            a = (ctype * max(1, len(items)))(*items)
            self._keep.append(a)
            return a

        objs = []
        for t, i, m in self.objects:
            o = rt_object()
            o.type, o.index, o.material, o.reserved = t, i, m, 0
            objs.append(o)
        fs.objects, fs.n_objects = arr(rt_object, objs), len(objs)
        fs.spheres, fs.n_spheres = arr(rt_sphere, self.spheres), len(self.spheres)
        fs.planes, fs.n_planes = arr(rt_plane, self.planes), len(self.planes)
        fs.models, fs.n_models = arr(rt_model, self.models), len(self.models)
        fs.materials, fs.n_materials = arr(rt_material, self.materials), len(self.materials)
        fs.lights, fs.n_lights = arr(rt_light, self.lights), len(self.lights)
        self.tri = []
        for k in range(6):
            if self.meshes:
                a = np.ascontiguousarray(np.concatenate([m.v[k] for m in self.meshes], axis=0), dtype=np.float32)
            else:
                a = np.zeros((1, 3), np.float32)
            self.tri.append(a)
        fs.v0, fs.v1, fs.v2, fs.n0, fs.n1, fs.n2 = [fptr(a) for a in self.tri]
        fs.n_faces = self.num_faces
        self._c = fs
        return fs


def bvh_check(flat):
    """rt_bvh_check: build the scene's BVHs on the host (no GPU) and verify them; returns the result struct."""
    lib = cabi.load()
    out = cabi.rt_bvh_check_result()
    cabi.check(lib, lib.rt_bvh_check(C.byref(flat.c_struct()), C.byref(out)))
    return out


class MultiScene:
    """rt_multi: the scene replicated on several GPUs of this process (a device may be listed more than once); whole
    frames are rendered as round-robin tiles, gathered on devices[0] by peer copies. Fails loudly without the library or a GPU."""

    def __init__(self, flat, devices, bvh="host"):
        self.lib = cabi.load()
        self.flat = flat
        self.h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        flags = {"host": 0, "device": cabi.RT_SCENE_BVH_DEVICE}[bvh]
        cabi.check(self.lib, self.lib.rt_multi_create(C.byref(flat.c_struct()), devs, len(devices), flags, C.byref(self.h)))

    def parts(self):
        return int(self.lib.rt_multi_parts(self.h))

    def render(self, cam, params, out=None, want_stats=True):
        """rt_multi_render -> (H,W,3) uint8 BGR frame in host memory, summed rt_stats."""
        if out is None:
            out = np.zeros((cam.imHeight, cam.imWidth, 3), np.uint8)
        st = rt_stats()
        cabi.check(self.lib, self.lib.rt_multi_render(self.h, C.byref(cam), C.byref(params.c), cabi.u8ptr(out),
                                                      C.byref(st) if want_stats else None))
        return out, st

    def render_device(self, cam, params, d_ptr):
        """rt_multi_render_device (asynchronous): the frame into a device buffer on devices[0]; status() waits."""
        cabi.check(self.lib, self.lib.rt_multi_render_device(self.h, C.byref(cam), C.byref(params.c), C.c_void_p(d_ptr), None))

    def status(self):
        cabi.check(self.lib, self.lib.rt_multi_status(self.h))

    def close(self):
        if self.h:
            self.lib.rt_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceScene:
    """rt_scene on one GPU. Fails loudly when the HIP library or a GPU is missing."""

    def __init__(self, flat, device=0, bvh="host", packets=True):
        """bvh = "host": the SAH builder (rt_scene_create); "device": rt_scene_create_ex with RT_SCENE_BVH_DEVICE.
        packets=False: RT_SCENE_NO_PACKETS (every mesh ray traversed by itself; per-ray work counters)."""
        self.lib = cabi.load()
        self.flat = flat
        self.device = device
        self.h = C.c_void_p()
        flags = {"host": 0, "device": cabi.RT_SCENE_BVH_DEVICE}[bvh] | (0 if packets else cabi.RT_SCENE_NO_PACKETS)
        cabi.check(self.lib, self.lib.rt_scene_create_ex(C.byref(flat.c_struct()), device, flags, C.byref(self.h)))

    def bvh_check(self):
        """rt_scene_bvh_check: the tree this scene holds in HBM, copied back and walked on the host."""
        out = cabi.rt_bvh_check_result()
        cabi.check(self.lib, self.lib.rt_scene_bvh_check(self.h, C.byref(self.flat.c_struct()), C.byref(out)))
        return out

    def close(self):
        if self.h:
            self.lib.rt_scene_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam, params, rect=None, out=None, want_stats=True):
        """rt_render -> (H,W,3) uint8 BGR frame (only the rectangle is written), rt_stats."""
        W, H = cam.imWidth, cam.imHeight
        if out is None:
            out = np.zeros((H, W, 3), np.uint8)
        x0, y0, w, h = rect if rect is not None else (0, 0, W, H)
        st = rt_stats()
        cabi.check(self.lib, self.lib.rt_render(self.h, C.byref(cam), C.byref(params.c), x0, y0, w, h,
                                                cabi.u8ptr(out), C.byref(st) if want_stats else None))
        return out, st

    def render_progress(self, cam, params, callback, rect=None, out=None):
        """rt_render_progress: rt_render that calls callback(pixels_done, pixels_total) after every pass and at the end."""
        W, H = cam.imWidth, cam.imHeight
        if out is None:
            out = np.zeros((H, W, 3), np.uint8)
        x0, y0, w, h = rect if rect is not None else (0, 0, W, H)
        st = rt_stats()
        fn = cabi.PROGRESS_FN(lambda user, done, total: callback(int(done), int(total)))
        cabi.check(self.lib, self.lib.rt_render_progress(self.h, C.byref(cam), C.byref(params.c), x0, y0, w, h,
                                                         cabi.u8ptr(out), C.byref(st), fn, None))
        return out, st

    def render_device(self, cam, params, d_ptr, rect=None, stream=None, stats=None):
        W, H = cam.imWidth, cam.imHeight
        x0, y0, w, h = rect if rect is not None else (0, 0, W, H)
        cabi.check(self.lib, self.lib.rt_render_device(self.h, C.byref(cam), C.byref(params.c), x0, y0, w, h,
                                                       C.c_void_p(d_ptr), C.c_void_p(stream or 0),
                                                       C.byref(stats) if stats is not None else None))

    def render_device_noalloc(self, cam, params, d_ptr, rect=None, stream=None):
        """rt_render_device_noalloc: rt_render_device without any allocation (static 20 MiB workspace, <= 2^21 samples)."""
        W, H = cam.imWidth, cam.imHeight
        x0, y0, w, h = rect if rect is not None else (0, 0, W, H)
        cabi.check(self.lib, self.lib.rt_render_device_noalloc(self.h, C.byref(cam), C.byref(params.c), x0, y0, w, h,
                                                               C.c_void_p(d_ptr), C.c_void_p(stream or 0)))

    def render_tiles_device(self, cam, params, tile, part, n_parts, d_ptr, stream=None, stats=None):
        cabi.check(self.lib, self.lib.rt_render_tiles_device(self.h, C.byref(cam), C.byref(params.c), tile[0],
                                                             tile[1], part, n_parts, C.c_void_p(d_ptr),
                                                             C.c_void_p(stream or 0),
                                                             C.byref(stats) if stats is not None else None))

    def render_batch_device(self, cams, params, d_ptr, frame_stride, rect=None, stream=None, stats=None):
        """rt_render_batch_device: len(cams) frames (same size) in one call; frame b at d_ptr + b * frame_stride."""
        arr = (rt_camera * len(cams))(*cams)
        W, H = cams[0].imWidth, cams[0].imHeight
        x0, y0, w, h = rect if rect is not None else (0, 0, W, H)
        cabi.check(self.lib, self.lib.rt_render_batch_device(self.h, arr, len(cams), C.byref(params.c), x0, y0, w, h,
                                                             C.c_void_p(d_ptr), int(frame_stride), C.c_void_p(stream or 0),
                                                             C.byref(stats) if stats is not None else None))

    def render_batch_host_async(self, cams, params, host_ptr, frame_stride, rect=None, stream=None):
        """rt_render_batch_host_async: len(cams) frames into host memory (page-locked for a truly asynchronous copy);
        valid after status()."""
        arr = (rt_camera * len(cams))(*cams)
        W, H = cams[0].imWidth, cams[0].imHeight
        x0, y0, w, h = rect if rect is not None else (0, 0, W, H)
        cabi.check(self.lib, self.lib.rt_render_batch_host_async(self.h, arr, len(cams), C.byref(params.c), x0, y0, w, h,
                                                                 C.c_void_p(host_ptr), int(frame_stride), C.c_void_p(stream or 0)))

    def render_tiles_batch_device(self, cams, params, tile, part, n_parts, d_ptr, frame_stride, stream=None, stats=None):
        arr = (rt_camera * len(cams))(*cams)
        cabi.check(self.lib, self.lib.rt_render_tiles_batch_device(self.h, arr, len(cams), C.byref(params.c), tile[0], tile[1],
                                                                   part, n_parts, C.c_void_p(d_ptr), int(frame_stride),
                                                                   C.c_void_p(stream or 0),
                                                                   C.byref(stats) if stats is not None else None))

    def intersect(self, rays, obj=-1, form=None):
        """rt_intersect, or rt_intersect_form when `form` (cabi.RT_TRAVERSE_*) is given."""
        rays = cabi.as_f32(rays, (-1, 6))
        n = rays.shape[0]
        hit = np.full(n, -2, np.int32)
        dist = np.zeros(n, np.float32)
        nrm = np.zeros((n, 3), np.float32)
        hp = hit.ctypes.data_as(C.POINTER(C.c_int32))
        if form is None:
            cabi.check(self.lib, self.lib.rt_intersect(self.h, obj, fptr(rays), n, hp, fptr(dist), fptr(nrm)))
        else:
            cabi.check(self.lib, self.lib.rt_intersect_form(self.h, form, obj, fptr(rays), n, hp, fptr(dist), fptr(nrm)))
        return hit, dist, nrm

    def occluded(self, rays, max_dist, form=None):
        rays = cabi.as_f32(rays, (-1, 6))
        md = cabi.as_f32(max_dist, (-1,))
        occ = np.zeros(rays.shape[0], np.uint8)
        if form is None:
            cabi.check(self.lib, self.lib.rt_occluded(self.h, fptr(rays), fptr(md), rays.shape[0], cabi.u8ptr(occ)))
        else:
            cabi.check(self.lib, self.lib.rt_occluded_form(self.h, form, fptr(rays), fptr(md), rays.shape[0], cabi.u8ptr(occ)))
        return occ

    def trace(self, params, rays, form=None):
        rays = cabi.as_f32(rays, (-1, 6))
        rgb = np.zeros((rays.shape[0], 3), np.float32)
        st = rt_stats()
        if form is None:
            cabi.check(self.lib, self.lib.rt_trace(self.h, C.byref(params.c), fptr(rays), rays.shape[0], fptr(rgb),
                                                   C.byref(st)))
        else:
            cabi.check(self.lib, self.lib.rt_trace_form(self.h, form, C.byref(params.c), fptr(rays), rays.shape[0],
                                                        fptr(rgb), C.byref(st)))
        return rgb, st

    def pass_samples(self):
        """rt_scene_pass_samples: level-0 samples per pass (frames with more are rendered in several passes)."""
        return int(self.lib.rt_scene_pass_samples(self.h))

    def set_pass_samples(self, samples):
        """rt_scene_set_pass_samples: level-0 samples per pass (power of two in [2^12, 2^26]; 0 = default)."""
        cabi.check(self.lib, self.lib.rt_scene_set_pass_samples(self.h, int(samples)))

    def set_option(self, name, value):
        """rt_scene_set_option: fix one choice of the launch schedule (tests, experiments); results never depend on it."""
        cabi.check(self.lib, self.lib.rt_scene_set_option(self.h, name.encode(), int(value)))

    def bvh_info(self):
        info = cabi.rt_bvh_info()
        cabi.check(self.lib, self.lib.rt_scene_bvh_info(self.h, C.byref(info)))
        return info

    def status(self):
        """rt_scene_status: wait for the asynchronous frames; raises RtError (RT_ERR_OVERFLOW) if one ran out of pool."""
        cabi.check(self.lib, self.lib.rt_scene_status(self.h))

    def stream(self):
        """The scene's own hipStream_t (integer handle), used by asynchronous calls when no stream is passed."""
        return self.lib.rt_scene_stream(self.h)

    def set_count_work(self, on):
        cabi.check(self.lib, self.lib.rt_set_count_work(self.h, int(on)))

    def get_work(self, kernel=-1):
        """Traversal work counters of the last counted call: kernel 0..4 = k_extend_wide, k_shade_wide, k_extend_quad,
        k_shade_quad<0|1>, k_shade_quad<2>; -1 = all."""
        w = cabi.rt_work()
        cabi.check(self.lib, self.lib.rt_get_work(self.h, int(kernel), C.byref(w)))
        return w


def cast_rays(cam, params, xy, device=0):
    """rt_cast_rays -> (n, spp, 6) float32."""
    lib = cabi.load()
    xy = np.ascontiguousarray(np.asarray(xy, dtype=np.uint32).reshape(-1, 2))
    out = np.zeros((xy.shape[0], params.spp, 6), np.float32)
    cabi.check(lib, lib.rt_cast_rays(device, C.byref(cam), C.byref(params.c),
                                     xy.ctypes.data_as(C.POINTER(C.c_uint32)), xy.shape[0], fptr(out)))
    return out
