"""The host BVH builder (raytracer_amd/csrc/bvh_build.cpp) through rt_bvh_check: built on the CPU, no GPU call, and
walked for the properties the quad traversal relies on (every face in exactly one leaf, leaf floats equal to what
Triangle::intersect forms (reference src/primitives.cpp:52-53), child boxes contain what is below them, stack bound).
That the hierarchy returns the brute-force loop's hits (reference src/primitives.cpp:376-382) is the GPU parity
tests' job (tests/test_gpu_parity.py::test_intersect_bit_exact)."""
import numpy as np
import pytest

from raytracer_amd import meshgen, workloads as wl
from raytracer_amd.scene import FlatScene, Mesh, bvh_check, material


def scene_of(*meshes):
    fs = FlatScene()
    for m in meshes:
        fs.add_mesh(m, material(color=(1, 1, 1)))
    fs.add_light((0, 3, 0), 1.0, 1.0)
    return fs


def raw_mesh(v0, v1, v2):
    v0, v1, v2 = (np.asarray(a, np.float32).reshape(-1, 3) for a in (v0, v1, v2))
    n = np.zeros_like(v0)
    c = np.concatenate([v0, v1, v2]).mean(0)
    r = np.linalg.norm(np.concatenate([v0, v1, v2]) - c, axis=1).max()
    return Mesh(v0, v1, v2, n, n, n, [c[0], c[1], c[2], r])


@pytest.mark.parametrize("name", ["bunny10", "bunny29", "field_small", "two_models"])
def test_builder_sound_on_workload_meshes(name):
    if name == "bunny10":
        fs = wl.scene_bunny(10)
    elif name == "bunny29":
        fs = wl.scene_bunny(29)
    elif name == "field_small":
        fs = wl.scene_field(64, 8)
    else:
        fs = scene_of(wl.yaml_mesh(6), wl.yaml_mesh(9))
    r = bvh_check(fs)
    assert r.errors == 0
    assert r.n_faces == fs.num_faces
    assert r.n_leaves * 4 >= r.n_faces and r.max_stack <= 64
    # leaves are kept full: on a closed mesh at most a fifth of the slots stay empty
    assert r.n_faces / (4 * r.n_leaves) > 0.8


@pytest.mark.parametrize("n", [1, 2, 4, 5, 17, 100])
def test_builder_sound_on_identical_triangles(n):
    """All centroids equal: the SAH finds no split and the median split has to carry the build."""
    t = np.array([[0, 0, -5], [1, 0, -5], [0, 1, -5]], np.float32)
    fs = scene_of(raw_mesh(np.repeat(t[0:1], n, 0), np.repeat(t[1:2], n, 0), np.repeat(t[2:3], n, 0)))
    r = bvh_check(fs)
    assert r.errors == 0 and r.n_faces == n and r.n_leaves == (n + 3) // 4


def test_builder_sound_on_sliver_fan():
    """A pathological input for a SAH tree (long thin triangles sharing a vertex, sizes over six decades)."""
    k = np.arange(400)
    s = (10.0 ** (-6 + 6 * k / 399)).astype(np.float32)
    v0 = np.zeros((400, 3), np.float32)
    v1 = np.stack([s, np.zeros(400, np.float32), -np.ones(400, np.float32)], 1)
    v2 = np.stack([s, 1e-3 * s, -np.ones(400, np.float32)], 1)
    r = bvh_check(scene_of(raw_mesh(v0, v1, v2)))
    assert r.errors == 0 and r.max_stack <= 64


def test_from_arrays_equals_ply_loader(tmp_path):
    """rt_host_model_from_arrays is Model::load_ply without the text (reference src/primitives.cpp:287-362)."""
    V, F = meshgen.blob(5, seed=7)
    p = tmp_path / "m.ply"
    meshgen.write_ply(str(p), V, F)
    kw = dict(rotation=(0.1, 0.5, -0.2), scale=3.0, translation=(1, 2, -3))
    a, b = Mesh.load_ply(str(p), **kw), Mesh.from_arrays(V, F, **kw)
    for x, y in zip(a.v, b.v):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert np.array_equal(a.bb.view(np.uint32), b.bb.view(np.uint32))
    with pytest.raises(Exception):
        Mesh.from_arrays(V, F + V.shape[0])
