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
