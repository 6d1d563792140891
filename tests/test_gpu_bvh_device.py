"""The GPU BVH builder (raytracer_amd/csrc/bvh_device.hip): the tree it leaves in HBM is
sound (same walk as the host builder's, rt_scene_bvh_check) and closest hits, shadow tests and frames through it equal
the oracle's brute-force loop (reference src/primitives.cpp:376-382) exactly as with the host's SAH tree."""
import numpy as np
import pytest

from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene, Params, reference_camera
from test_bvh_build import raw_mesh, scene_of
from test_gpu_parity import frame_stats, random_rays

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["bunny10", "bunny29", "field_small", "two_models", "tiny"])
def test_device_tree_is_sound(name):
    if name == "bunny10":
        fs = wl.scene_bunny(10)
    elif name == "bunny29":
        fs = wl.scene_bunny(29)
    elif name == "field_small":
        fs = wl.scene_field(64, 8)
    elif name == "two_models":
        fs = scene_of(wl.yaml_mesh(6), wl.yaml_mesh(9))
    else:       # 1, 4, 5 and 9 faces: a lone leaf, a full leaf, two leaves, three leaves
        t = np.array([[0, 0, -5], [1, 0, -5], [0, 1, -5]], np.float32)
        ms = []
        for n in (1, 4, 5, 9):
            off = np.arange(n, dtype=np.float32)[:, None] * np.array([[0.1, 0.2, -0.3]], np.float32)
            ms.append(raw_mesh(t[0] + off, t[1] + off, t[2] + off))
        fs = scene_of(*ms)
    dev = DeviceScene(fs, bvh="device")
    info = dev.bvh_info()
    r = dev.bvh_check()
    print(name, "faces", r.n_faces, "nodes", r.n_nodes, "leaves", r.n_leaves, "depth", r.max_depth, "stack", r.max_stack,
          "build %.1f ms" % (info.build_seconds * 1e3))
    assert info.built_on_device == 1
    assert r.errors == 0
    assert r.n_faces == fs.num_faces and r.max_stack <= 64
    # the host-built tree of the same scene passes the same walk
    assert DeviceScene(fs).bvh_check().errors == 0


def test_identical_triangles_fall_back_or_build():
    """All centroids equal: the keys differ only by index, the radix tree is as deep as log2(n) - still within the
    stack bound; whatever the builder decides, the tree must be sound."""
    n = 1000
    t = np.array([[0, 0, -5], [1, 0, -5], [0, 1, -5]], np.float32)
    fs = scene_of(raw_mesh(np.repeat(t[0:1], n, 0), np.repeat(t[1:2], n, 0), np.repeat(t[2:3], n, 0)))
    dev = DeviceScene(fs, bvh="device")
    assert dev.bvh_check().errors == 0


def test_hits_equal_the_oracle(oracle_mod):
    O = oracle_mod
    fs = wl.scene_bunny(10)
    dev = DeviceScene(fs, bvh="device")
    orc = O.Oracle(fs)
    rays = random_rays(20000, 5)
    mesh_obj = [k for k, o in enumerate(fs.objects) if o[0] == 2][0]
    for obj in (-1, mesh_obj):
        h1, d1, n1 = dev.intersect(rays, obj)
        h2, d2, n2 = orc.intersect(rays, obj)
        assert np.array_equal(h1, h2)
        hit = h2 >= 0
        assert hit.any()
        assert np.array_equal(d1[hit].view(np.uint32), d2[hit].view(np.uint32))
        assert np.array_equal(n1[hit].view(np.uint32), n2[hit].view(np.uint32))
    md = np.random.default_rng(3).uniform(0.1, 12, rays.shape[0]).astype(np.float32)
    assert np.array_equal(dev.occluded(rays, md), orc.occluded(rays, md))


def test_frame_equals_host_tree_frame():
    fs = wl.scene_bunny(29)
    cam = reference_camera(640, 480)
    p = Params(aa=wl.aa_grid(2))
    a, sa = DeviceScene(fs, bvh="device").render(cam, p)
    b, sb = DeviceScene(fs).render(cam, p)
    assert np.array_equal(a, b)
    assert sa.trace_calls == sb.trace_calls and sa.shadow_rays == sb.shadow_rays
