"""Where the reference build exists (oracle/_ref/libref.so, i.e. in the container that has /root/reference, and on
the GPU box, to which the prebuilt library travels) the C restatement is checked against it directly on fresh
random inputs, beyond the committed fixtures."""
import numpy as np
import pytest

from helpers import same_bits
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DEFAULT_AA, Params, reference_camera


@pytest.fixture(scope="module")
def O(oracle_mod):
    if not oracle_mod.have_ref():
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    return oracle_mod


@pytest.mark.parametrize("third,depth,aa", [("glass", 10, ()), ("glass2", 10, ()), ("glass2", 3, DEFAULT_AA)])
def test_mesh_scene_frames(O, third, depth, aa):
    fs = wl.scene_bunny(4, third=third)
    cam = reference_camera(128, 96)
    p = Params(max_depth=depth, aa=aa)
    a, _ = O.Reference(fs).render(cam, p)
    b, _ = O.Oracle(fs).render(cam, p)
    assert np.array_equal(a, b)


def test_random_rays(O):
    fs = wl.scene_bunny(6, third="glass2")
    rng = np.random.default_rng(99)
    o = rng.uniform(-3.9, 3.9, (20000, 3)).astype(np.float32)
    o[:, 2] = rng.uniform(-11.5, 0.5, 20000).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    rays = np.concatenate([o, d], axis=1)
    R, Q = O.Reference(fs), O.Oracle(fs)
    for obj in (-1, 0, 2, 3, 5):
        h1, d1, n1 = R.intersect(rays, obj)
        h2, d2, n2 = Q.intersect(rays, obj)
        assert np.array_equal(h1, h2)
        hit = h1 >= 0
        assert same_bits(d1[hit], d2[hit]) and same_bits(n1[hit], n2[hit])
    c1, _ = R.trace(Params(), rays[:4000])
    c2, _ = Q.trace(Params(), rays[:4000])
    assert same_bits(c1, c2)


def scanned_like_mesh(tmp_path):
    """What a scanned .ply brings that the procedural meshes do not: faces of zero area (a vertex named twice; three
    vertices on a line; three identical points), a vertex no face uses, two faces with the same three vertices in
    opposite orders. calcNormal divides by a zero length there, so their face normals - and the normals of the
    vertices they touch - are NaN (reference src/primitives.cpp:238-362)."""
    from raytracer_amd import meshgen
    V, F = meshgen.blob(3, seed=5, radius=0.9)
    V = np.concatenate([V, [[0.1, 0.2, 0.3], [0.2, 0.4, 0.6], [0.4, 0.8, 1.2], [5, 5, 5]]]).astype(np.float32)
    n = V.shape[0] - 4
    extra = [[0, 0, 1], [n, n + 1, n + 2], [7, 7, 7], [F[3][0], F[3][2], F[3][1]], [2, n, 9]]
    F = np.concatenate([F, np.array(extra, F.dtype)])
    path = tmp_path / "scanned.ply"
    meshgen.write_ply(path, V, F)
    return path


def test_ply_loader_on_degenerate_faces(O, tmp_path):
    """The host surface's loader against the reference's own Model::load_ply on such a file, bit for bit (NaNs by their
    bits too), raw and through the scene loader's transform chain."""
    from raytracer_amd.scene import Mesh
    path = scanned_like_mesh(tmp_path)
    for kw in (dict(transform=False), dict(rotation=(0.1, 0.5236, -0.2), scale=2.5, translation=(-1.5, -1, -6))):
        mine = Mesh.load_ply(path, **kw)
        arrs, bb = O.ref_load_ply(path, **kw)
        assert np.isnan(np.stack(arrs[3:])).any()                       # the file does produce NaN normals
        for a, b in zip(mine.v, arrs):
            assert same_bits(a, b)
        assert same_bits(mine.bb, bb)
