"""The scenes BASELINE.json's configs name, built from the reference's own scene files' values
(/root/reference/scene.yaml and scene_exp.yaml, cited per item) with the procedural mesh of meshgen.py where
the reference points at a .ply that does not ship with it.
"""
import os
import tempfile

import numpy as np

from . import meshgen
from .scene import DEFAULT_AA, FlatScene, Mesh, Params, material, reference_camera

# scene.yaml:2-8 colours
GLASS_RGB = (0.01, 0.1, 1.0)
WHITE, LIGHT_GRAY = 1.0, 0.8


def yaml_materials(shadows_only=False):
    """scene.yaml:10-73. shadows_only clears every reflective/refractive flag (BASELINE configs 2 and 3)."""
    m = {
        "glass": dict(color=GLASS_RGB, ka=0.1, kd=0.1, ks=0.2, kr=0.8, kt=0.8, specRefExp=8, refractiveIdx=1.5,
                      reflective=True, refractive=False),
        "glass2": dict(color=GLASS_RGB, ka=0, kd=0, ks=0.8, kr=0.2, kt=0.8, specRefExp=8, refractiveIdx=1.5,
                       reflective=True, refractive=True),
        "porcelan": dict(color=WHITE, ka=0.2, kd=0.5, ks=0.7, kr=0.3, kt=0, specRefExp=8, refractiveIdx=0,
                         reflective=True),
        "mirror": dict(color=WHITE, ka=0, kd=0, ks=0, kr=1, kt=1, specRefExp=8, refractiveIdx=0, reflective=True),
        "m1": dict(color=LIGHT_GRAY, ka=0.1, kd=0.7, ks=0, kr=0.3, kt=0, specRefExp=8, refractiveIdx=0,
                   reflective=False),
        "m2": dict(color=WHITE, ka=0.1, kd=0.7, ks=0.4),
        "blue": dict(color=(0, 0, 1)),
        "red": dict(color=(1, 0, 0)),
        "green": dict(color=(0, 1, 0)),
    }
    out = {}
    for k, v in m.items():
        if shadows_only:
            v = dict(v, reflective=False, refractive=False)
        out[k] = material(**v)
    return out


def add_yaml_lights(fs):
    """scene.yaml:75-81."""
    fs.add_light((0.0, 3.0, -7.5), WHITE, 3.0)
    fs.add_light((0.0, 0.0, -1.0), WHITE, 10.0)


def add_yaml_spheres(fs, mats, third="glass"):
    """scene.yaml:85-96."""
    fs.add_sphere((0.0, 0.0, -8.0), 1.0, mats["mirror"])
    fs.add_sphere((2.0, 0.25, -8.0), 0.75, mats["m1"])
    fs.add_sphere((-2.5, 2.0, -5.0), 1.0, mats[third])


def add_yaml_planes(fs, mats):
    """scene.yaml:106-125."""
    fs.add_plane((0.0, 0.0, 1.0), 12.0, mats["green"])
    fs.add_plane((1.0, 0.0, 0.0), 4.0, mats["red"])
    fs.add_plane((-1.0, 0.0, 0.0), 4.0, mats["blue"])
    fs.add_plane((0.0, 1.0, 0.0), 4.0, mats["blue"])
    fs.add_plane((0.0, -1.0, 0.0), 4.0, mats["red"])


_mesh_cache = {}
PLY_OVERRIDE = None      # bench.py --ply: a supplied mesh file (the reference's scene.yaml:103 names bun_zipper_res2.ply) instead of the blob


def yaml_mesh(n_subdiv, seed=2, cache_dir=None):
    """The mesh object of scene.yaml:98-104 (scale 20, rotation y 0.5236, translation (-1.5,-1,-6)) with a
    procedural 20*n*n-face blob in place of bun_zipper*.ply, taken through the PLY loader + transform chain of
    the C++ host surface exactly as the reference's loader would."""
    key = (n_subdiv, seed)
    if PLY_OVERRIDE:      # the same loader and transform chain, on the caller's file
        key = ("ply", PLY_OVERRIDE)
        if key not in _mesh_cache:
            _mesh_cache[key] = Mesh.load_ply(PLY_OVERRIDE, rotation=(0.0, 0.5235987755982988, 0.0), scale=20.0,
                                             translation=(-1.5, -1.0, -6.0))
        return _mesh_cache[key]
    if key in _mesh_cache:
        return _mesh_cache[key]
    d = cache_dir or tempfile.gettempdir()
    path = os.path.join(d, f"rtamd_blob_n{n_subdiv}_s{seed}_{os.getpid()}.ply")
    V, F = meshgen.blob(n_subdiv, seed=seed)
    meshgen.write_ply(path, V, F)
    try:
        mesh = Mesh.load_ply(path, rotation=(0.0, 0.5235987755982988, 0.0), scale=20.0,
                             translation=(-1.5, -1.0, -6.0))
    finally:
        os.unlink(path)
    _mesh_cache[key] = mesh
    return mesh


def scene_analytic(third="glass", shadows_only=False):
    """Analytic part of scene.yaml: 3 spheres + 5 planes + 2 lights, no mesh (SURVEY.md §8c known answers)."""
    fs = FlatScene()
    mats = yaml_materials(shadows_only)
    add_yaml_spheres(fs, mats, third)
    add_yaml_planes(fs, mats)
    add_yaml_lights(fs)
    return fs


def scene_bunny(n_subdiv=59, shadows_only=False, spheres=True, third="glass"):
    """scene.yaml as shipped, objects in file order: 3 spheres, the mesh (porcelan), 5 planes.
    n_subdiv=59 -> 69,620 faces (bun_zipper.ply has 69,451); n_subdiv=29 -> 16,820 (bun_zipper_res2: 16,301)."""
    fs = FlatScene()
    mats = yaml_materials(shadows_only)
    if spheres:
        add_yaml_spheres(fs, mats, third)
    fs.add_mesh(yaml_mesh(n_subdiv), mats["porcelan"])
    add_yaml_planes(fs, mats)
    add_yaml_lights(fs)
    return fs


FIELD_TILT = 0.5      # radians about x: the far end of the field rises towards the camera's upper half


def scene_field(n_blobs=2000, n_subdiv=16, seed=3):
    """BASELINE config 5: ONE mesh of n_blobs displaced geodesic blobs on a square grid
    (2000 x 20*16*16 = 10.24 M triangles) resting on a floor plane, a wall behind, under scene.yaml's two lights.
    Scale 2 keeps every edge above 0.01 (the absolute |det| < 1e-6 test of Triangle::intersect, primitives.cpp:63).
    The camera is the reference's (origin, looking down -z); field and floor are tilted by FIELD_TILT about x and
    centred at (0,-3,-20), so the field covers the lower two thirds of the frame out to the wall at z = -40."""
    import ctypes as C
    from .scene import fptr, host_lib
    fs = FlatScene()
    mats = yaml_materials()
    V, F = meshgen.blob_field(n_blobs, n_subdiv, seed=seed)
    centre = np.array([0.0, -3.0, -20.0])
    fs.add_mesh(Mesh.from_arrays(V, F, rotation=(FIELD_TILT, 0.0, 0.0), scale=2.0, translation=tuple(centre)),
                mats["porcelan"])
    R = np.zeros(9, np.float32)
    assert host_lib().rt_host_mat3_rotation(C.c_float(FIELD_TILT), C.c_float(0.0), C.c_float(0.0), fptr(R)) == 0
    n = R.reshape(3, 3).astype(np.float64) @ np.array([0.0, 1.0, 0.0])      # the field's up after Model *= rotation
    fs.add_plane(tuple(n), float(-(n @ centre) + 0.3), mats["m1"])          # 0.3 below the blobs' centres
    fs.add_plane((0.0, 0.0, 1.0), 40.0, mats["green"])
    add_yaml_lights(fs)
    return fs


def aa_grid(n):
    """n x n stratified grid minus (0,0) (the corner ray is always cast): n*n samples per pixel."""
    pts = [((i % n) / n, (i // n) / n) for i in range(1, n * n)]
    return pts


WORKLOADS = {
    # name: (scene factory, width, height, aa pattern, max_depth)
    # BASELINE config 1 at the reference's own resolution, with the low-res mesh class
    "c1_scene_yaml_800x600": (lambda: scene_bunny(29), 800, 600, (), 10),
    # BASELINE config 2
    "c2_analytic_1080p_4spp": (lambda: scene_analytic(shadows_only=True), 1920, 1080,
                               ((0.5, 0.5), (0.25, 0.75), (0.75, 0.25)), 10),
    # the analytic part of scene.yaml alone at the bench's size (diagnostics: what the frame costs without its mesh)
    "analytic_1080p": (lambda: scene_analytic(), 1920, 1080, (), 10),
    # BASELINE config 3 (strict reading: Phong + hard shadows only)
    "c3_bunny_1080p_shadows": (lambda: scene_bunny(59, shadows_only=True, spheres=False), 1920, 1080, (), 10),
    # The metric's "bunny scene 1920x1080" with the scene file's own materials (reflections + shadows): bench default
    "bunny_1080p": (lambda: scene_bunny(59), 1920, 1080, (), 10),
    # BASELINE config 4
    "c4_bunny_glass_4k_16spp": (lambda: scene_bunny(59, third="glass2"), 3840, 2160, aa_grid(4), 8),
    # BASELINE config 5 (one GPU renders its 1/n_parts of the tiles; the whole frame on one GPU is the N=1 case)
    "c5_field_10m_8k_64spp": (lambda: scene_field(), 7680, 4320, aa_grid(8), 10),
    # the 10 M-triangle scene at a frame size counter passes can afford (2 passes of 4 M samples)
    "c5_field_10m_1080p_4spp": (lambda: scene_field(), 1920, 1080, aa_grid(2), 10),
    # the same scene at a size the tests and a quick look can afford
    "c5_field_small_720p": (lambda: scene_field(64, 8), 1280, 720, (), 10),
}


def orbit_cameras(width, height, n, radius=0.05):
    """n cameras for the frames of a batch: camera 0 is the reference's (src/raytracer.cpp:674-689), camera k's eye is
    moved along a small circle through it (radius 0.05 in the x-y plane, 32 positions per turn), so that the frames
    of a batch are different pictures that share no rays."""
    import math
    from .scene import make_camera
    cams = []
    for k in range(n):
        th = 2.0 * math.pi * (k % 32) / 32.0
        cams.append(make_camera(width, height, fov=60.0,
                                move=(-0.5 + radius * (math.cos(th) - 1.0), radius * math.sin(th), -0.25)))
    return cams


def make_workload(name, ply=None, scene_file=None):
    """ply: a .ply to use wherever the workload's scene holds scene.yaml's mesh object (its transform chain applied);
    scene_file: a scene file in the reference's schema (src/scene.cpp:32-178) in place of the workload's scene - frame
    size, samples per pixel and depth stay the workload's, the camera is the reference's."""
    global PLY_OVERRIDE
    factory, W, H, aa, depth = WORKLOADS[name]
    if scene_file:
        fs = FlatScene.load_yaml(scene_file)
    else:
        PLY_OVERRIDE = os.path.abspath(ply) if ply else None
        try:
            fs = factory()
        finally:
            PLY_OVERRIDE = None
    return fs, reference_camera(W, H), Params(max_depth=depth, aa=aa)
