"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the same inputs.

Bars (SURVEY.md §8c / BASELINE.json north_star): rays, hit/miss, distances, normals and occlusion are float
or integer results of the same IEEE operations in the same order -> compared BIT-EXACT. Frames: 8-bit BGR within
1 LSB on >= 99.9 % of pixels (the tolerance north_star states); the tests also print the exactly-equal share.
"""
import numpy as np
import pytest

from raytracer_amd import workloads as wl
from raytracer_amd.scene import DEFAULT_AA, DeviceScene, Params, cast_rays, reference_camera

pytestmark = pytest.mark.gpu


def frame_stats(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16)).max(axis=2)
    return float((d <= 1).mean()), float((d == 0).mean()), int(d.max())


def camera_rays(O, cam, params, step=1):
    xy = [(x, y) for y in range(0, cam.imHeight, step) for x in range(0, cam.imWidth, step)]
    return O.cast_rays("orc", cam, params, xy).reshape(-1, 6)


def random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-3.5, 3.5, (n, 3)).astype(np.float32)
    o[:, 2] = rng.uniform(-11, 0, n).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    unit = rng.random(n) < 0.7          # the reference feeds non-unit directions to its tests too
    d[unit] /= np.linalg.norm(d[unit], axis=1, keepdims=True).astype(np.float32)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def test_cast_rays_bit_exact(oracle_mod):
    O = oracle_mod
    for (W, H) in [(800, 600), (1920, 1080)]:
        cam = reference_camera(W, H)
        for aa in [(), DEFAULT_AA]:
            p = Params(aa=aa)
            rng = np.random.default_rng(W)
            xy = np.stack([rng.integers(0, W, 4000), rng.integers(0, H, 4000)], axis=1)
            xy = np.concatenate([xy, [[0, 0], [W - 1, H - 1], [0, H - 1], [W - 1, 0]]]).astype(np.uint32)
            got = cast_rays(cam, p, xy)
            want = O.cast_rays("orc", cam, p, xy)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("scene_name", ["analytic", "mesh"])
def test_intersect_bit_exact(oracle_mod, scene_name):
    O = oracle_mod
    fs = wl.scene_analytic() if scene_name == "analytic" else wl.scene_bunny(12)
    dev = DeviceScene(fs)
    orc = O.Oracle(fs)
    cam = reference_camera(320, 240)
    rays = np.concatenate([camera_rays(O, cam, Params(), 1), random_rays(40000, 7)])
    n_obj = len(fs.objects)
    for obj in [-1] + list(range(n_obj)):
        h1, d1, n1 = dev.intersect(rays, obj)
        h2, d2, n2 = orc.intersect(rays, obj)
        assert np.array_equal(h1, h2), f"object {obj}: hit/miss differs on {(h1 != h2).sum()} rays"
        hit = h2 >= 0
        assert hit.any() or obj >= 0
        assert np.array_equal(d1[hit].view(np.uint32), d2[hit].view(np.uint32)), f"object {obj}: dist"
        assert np.array_equal(n1[hit].view(np.uint32), n2[hit].view(np.uint32)), f"object {obj}: normal"


def test_occluded_exact(oracle_mod):
    O = oracle_mod
    fs = wl.scene_bunny(12)
    dev = DeviceScene(fs)
    orc = O.Oracle(fs)
    rays = random_rays(60000, 11)
    md = np.random.default_rng(3).uniform(0.1, 12, rays.shape[0]).astype(np.float32)
    assert np.array_equal(dev.occluded(rays, md), orc.occluded(rays, md))


def edge_rays(fs, n_faces, seed):
    """Rays aimed exactly at shared edges and vertices of the scene's first mesh (edge midpoints, vertices, points a
    few ulps off an edge), from several origins, unit and non-unit directions: where two or more faces report a hit
    at the same or nearly the same t and the reference's ascending strict-< scan (src/primitives.cpp:376-382) decides."""
    rng = np.random.default_rng(seed)
    m = fs.meshes[0]
    v0, v1, v2 = m.v[0], m.v[1], m.v[2]
    f = rng.integers(0, m.num_faces, n_faces)
    targets = np.concatenate([
        (v0[f] + v1[f]) * np.float32(0.5), (v1[f] + v2[f]) * np.float32(0.5), (v0[f] + v2[f]) * np.float32(0.5),
        v0[f], v1[f], v2[f],
        np.nextafter((v0[f] + v1[f]) * np.float32(0.5), np.float32(np.inf)),
        v0[f] + (v1[f] - v0[f]) * np.float32(0.25),
    ]).astype(np.float32)
    rays = []
    for o in [(-0.5, 0.0, -0.25), (0.0, 3.0, -7.5), (0.0, 0.0, -1.0), (2.0, -3.0, -2.0)]:
        o = np.asarray(o, np.float32)
        d = targets - o
        rays.append(np.concatenate([np.broadcast_to(o, d.shape), d], axis=1))                       # non-unit
        dn = (d * (np.float32(1.0) / np.sqrt((d * d).sum(axis=1, keepdims=True)))).astype(np.float32)
        rays.append(np.concatenate([np.broadcast_to(o, d.shape), dn], axis=1))
    return np.concatenate(rays).astype(np.float32)


def far_rays(fs, n, seed):
    """Rays at the centroids of faces of the first mesh from origins 10 .. 10,000 away, a third of them along one axis
    exactly (two direction components 0) and a third in an axis plane (one component 0): the box tests run on a 1-ulp
    reciprocal, +-1e-30 in place of a zero component and an 8-ulp far bound (rt_kernels.hip: slab, safe_rcp), and must
    never cull a box whose triangle the exact test accepts."""
    rng = np.random.default_rng(seed)
    m = fs.meshes[0]
    f = rng.integers(0, m.num_faces, n)
    c = ((m.v[0][f] + m.v[1][f] + m.v[2][f]) * np.float32(1.0 / 3.0)).astype(np.float32)
    dirs = rng.normal(size=(n, 3))
    kind = np.arange(n) % 3
    axis = rng.integers(0, 3, n)
    for k in range(n):
        if kind[k] == 1:
            keep = dirs[k, axis[k]]; dirs[k] = 0.0; dirs[k, axis[k]] = keep if keep != 0 else 1.0
        elif kind[k] == 2:
            dirs[k, axis[k]] = 0.0
    dirs /= np.sqrt((dirs * dirs).sum(axis=1, keepdims=True))
    dist = 10.0 ** rng.uniform(1.0, 4.0, (n, 1))
    o = (c.astype(np.float64) - dirs * dist).astype(np.float32)
    d = c - o                                    # non-unit, exactly 0 where the origin was not moved
    d[dirs == 0.0] = 0.0
    dn = (d * (np.float32(1.0) / np.sqrt((d * d).sum(axis=1, keepdims=True)))).astype(np.float32)
    return np.concatenate([np.concatenate([o, d], axis=1), np.concatenate([o, dn], axis=1)]).astype(np.float32)


def doubled_faces_scene():
    """A mesh in which every face exists TWICE (and a third time displaced by a few ulps), the copies carrying
    different vertex normals: the normal that comes back says which face index won the tie."""
    from raytracer_amd.scene import FlatScene, Mesh, material
    base = wl.yaml_mesh(6)
    F = base.num_faces
    v = [np.concatenate([a, a, a]) for a in base.v[:3]]
    v[0][2 * F:] = np.nextafter(v[0][2 * F:], np.float32(np.inf))
    n = [np.concatenate([a, a * np.float32(0.5), a * np.float32(0.25)]) for a in base.v[3:]]
    perm = np.random.default_rng(5).permutation(3 * F)          # duplicates far apart in face order
    fs = FlatScene()
    fs.add_mesh(Mesh(*[a[perm] for a in v], *[a[perm] for a in n], base.bb), material((0.8, 0.8, 0.8)))
    fs.add_plane((0, 1, 0), 4.0, material((0.5, 0.5, 0.9)))
    wl.add_yaml_lights(fs)
    return fs


FORMS = [1, 2]      # cabi.RT_TRAVERSE_PACKET, RT_TRAVERSE_SERVE (0, the quads, is what the tests above use)


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("scene_name", ["mesh", "doubled", "field"])
def test_intersect_forms_bit_exact(oracle_mod, scene_name, form):
    """The traversals that carry most of a frame's rays - 64-ray packets (level 0 and its shadow rays) and quad service
    (bounce levels of large passes) - against the oracle's brute-force loop, bit for bit: hit object, distance and
    normal, on coherent camera rays (packets that share their path), unrelated random rays (packets that share
    nothing), and rays through shared edges, vertices and duplicated faces (ties broken by the lowest face index)."""
    O = oracle_mod
    fs = {"mesh": lambda: wl.scene_bunny(12), "doubled": doubled_faces_scene, "field": lambda: wl.scene_field(16, 4)}[scene_name]()
    dev = DeviceScene(fs)
    orc = O.Oracle(fs)
    cam = reference_camera(256, 192)
    rays = np.concatenate([camera_rays(O, cam, Params(), 1), random_rays(20000, 7), edge_rays(fs, 400, 3), far_rays(fs, 3000, 11)])
    mesh_obj = [k for k, o in enumerate(fs.objects) if o[0] == 2][0]
    for obj in (-1, mesh_obj):
        h2, d2, n2 = orc.intersect(rays, obj)
        hit = h2 >= 0
        assert hit.sum() > 1000
        for f in (0, form):
            h1, d1, n1 = dev.intersect(rays, obj, form=f)
            assert np.array_equal(h1, h2), f"form {f} object {obj}: hit/miss differs on {(h1 != h2).sum()} rays"
            assert np.array_equal(d1[hit].view(np.uint32), d2[hit].view(np.uint32)), f"form {f} object {obj}: dist"
            assert np.array_equal(n1[hit].view(np.uint32), n2[hit].view(np.uint32)), f"form {f} object {obj}: normal"
    md = np.random.default_rng(3).uniform(0.1, 30, rays.shape[0]).astype(np.float32)
    want = orc.occluded(rays, md)
    assert 0.05 < want.mean() < 0.95
    assert np.array_equal(dev.occluded(rays, md, form=form), want)


@pytest.mark.parametrize("form", FORMS)
def test_intersections_golden_forms(form):
    """The reference's own answers (tests/golden/intersect.npz, made by the compiled reference) through packets and
    quad service."""
    from helpers import gold, same_bits, sub
    from raytracer_amd.scene import FlatScene
    g = gold("intersect.npz")
    fs = FlatScene.from_arrays(sub(g, "scene_"))
    dev = DeviceScene(fs)
    rays = g["rays"]
    for obj in [-1] + list(range(len(fs.objects))):
        h, d, n = dev.intersect(rays, obj, form=form)
        assert np.array_equal(h, g[f"hit_{obj}"]), f"object {obj}: {(h != g[f'hit_{obj}']).sum()} rays differ"
        hit = h >= 0
        assert same_bits(d[hit], g[f"dist_{obj}"][hit]) and same_bits(n[hit], g[f"normal_{obj}"][hit])
    assert np.array_equal(dev.occluded(rays, g["max_dist"], form=form), g["occluded"])
    lone = DeviceScene(FlatScene.from_arrays(sub(g, "lone_")))
    h, d, n = lone.intersect(rays, -1, form=form)
    assert np.array_equal(h, g["lone_hit"])
    hit = h >= 0
    assert same_bits(d[hit], g["lone_dist"][hit]) and same_bits(n[hit], g["lone_normal"][hit])


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("third,depth", [("glass2", 10), ("glass", 3)])
def test_trace_colours_forms(oracle_mod, third, depth, form):
    """trace() with every level's rays and shadow rays through packets / quad service: float colours and per-level ray
    counts against the oracle's recursion (same bar as test_trace_colours)."""
    O = oracle_mod
    fs = wl.scene_bunny(8, third=third)
    dev = DeviceScene(fs)
    cam = reference_camera(200, 150)
    p = Params(max_depth=depth)
    rays = camera_rays(O, cam, p, 1)
    got, st1 = dev.trace(p, rays, form=form)
    want, st2 = O.Oracle(fs).trace(p, rays)
    assert st1.trace_calls == st2.trace_calls and st1.shadow_rays == st2.shadow_rays
    assert list(st1.level_rays) == list(st2.level_rays)
    same = (got.view(np.uint32) == want.view(np.uint32)).all(axis=1).mean()
    print(f"trace form {form} {third} depth {depth}: bit-equal colours {same:.6f}, max abs diff {np.abs(got - want).max():.3g}")
    assert np.abs(got - want).max() <= 2e-6
    assert same >= 0.999
    quad, _ = dev.trace(p, rays)
    assert np.array_equal(got.view(np.uint32), quad.view(np.uint32))      # the device forms agree to the bit (same pow)


@pytest.mark.parametrize("third,depth", [("glass", 10), ("glass2", 10), ("glass2", 1), ("glass2", 0), ("glass2", 8)])
def test_trace_colours(oracle_mod, third, depth):
    O = oracle_mod
    fs = wl.scene_bunny(8, third=third)
    dev = DeviceScene(fs)
    orc = O.Oracle(fs)
    cam = reference_camera(200, 150)
    p = Params(max_depth=depth)
    rays = camera_rays(O, cam, p, 1)
    got, st1 = dev.trace(p, rays)
    want, st2 = orc.trace(p, rays)
    assert st1.trace_calls == st2.trace_calls and st1.shadow_rays == st2.shadow_rays
    assert list(st1.level_rays) == list(st2.level_rays)
    same = (got.view(np.uint32) == want.view(np.uint32)).all(axis=1).mean()
    print(f"trace {third} depth {depth}: bit-equal colours {same:.6f}, max abs diff {np.abs(got - want).max():.3g}")
    # pow() is the one libm call on the path whose device implementation is not correctly rounded
    assert np.abs(got - want).max() <= 2e-6
    assert same >= 0.999


def test_generic_pow_exponent(oracle_mod):
    """A material with a non-integer specRefExp takes the shade kernels built with the device pow (DScene::generic_pow);
    scene.yaml's materials (exponent 8) take the ones without it."""
    O = oracle_mod
    fs = wl.scene_bunny(8)
    for m in fs.materials:
        m.specRefExp = 7.5
    dev = DeviceScene(fs)
    cam = reference_camera(200, 150)
    p = Params()
    rays = camera_rays(O, cam, p, 1)
    got, st1 = dev.trace(p, rays)
    want, st2 = O.Oracle(fs).trace(p, rays)
    assert st1.trace_calls == st2.trace_calls and st1.shadow_rays == st2.shadow_rays
    assert np.abs(got - want).max() <= 2e-6


@pytest.mark.parametrize("case", ["analytic_1spp", "analytic_9spp", "glass2", "mesh", "mesh_glass2_depth8", "mesh_rect", "field"])
def test_frame_parity(oracle_mod, case):
    O = oracle_mod
    W, H = (800, 600) if case.startswith("analytic") or case == "glass2" else (320, 240)
    cam = reference_camera(W, H)
    p = Params()
    rect = None
    if case == "analytic_1spp":
        fs = wl.scene_analytic()
    elif case == "analytic_9spp":
        fs, p = wl.scene_analytic(), Params(aa=DEFAULT_AA)
    elif case == "glass2":
        fs = wl.scene_analytic(third="glass2")
    elif case == "mesh":
        fs = wl.scene_bunny(10)
    elif case == "mesh_glass2_depth8":
        fs, p = wl.scene_bunny(10, third="glass2"), Params(max_depth=8, aa=wl.aa_grid(2))
    elif case == "field":      # BASELINE config 5's scene with 36 blobs of 320 faces
        fs, p = wl.scene_field(36, 4), Params(aa=wl.aa_grid(2))
    else:
        fs, rect = wl.scene_bunny(10), (37, 21, 201, 107)
    dev = DeviceScene(fs)
    got, st1 = dev.render(cam, p, rect=rect)
    want, st2 = O.Oracle(fs).render(cam, p, rect=rect)
    within1, equal, worst = frame_stats(got, want)
    print(f"{case}: within 1 LSB {within1:.6f}, equal {equal:.6f}, worst {worst}, rays {st1.trace_calls}+{st1.shadow_rays}")
    assert st1.trace_calls == st2.trace_calls and st1.shadow_rays == st2.shadow_rays
    assert within1 >= 0.999           # the north_star bar
    assert equal >= 0.9999            # what is measured: every pixel equal
    if case == "analytic_1spp":   # SURVEY.md §8c known answer of the reference itself
        assert st1.trace_calls == 534339 and st1.shadow_rays == 1068678
        assert abs(int(got.astype(np.int64).sum()) - 42340267) <= 0.001 * got.size
    if rect is not None:          # bytes outside the rectangle are untouched
        mask = np.ones((H, W), bool)
        mask[rect[1]:rect[1] + rect[3], rect[0]:rect[0] + rect[2]] = False
        assert not got[mask].any()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_spheres_frames(oracle_mod, seed):
    """Spheres of very different sizes and distances (radius 0.01 .. 30, up to 200 away, some behind the camera, some
    around it), shadows and reflections on: pins the wave-uniform exit of the sphere test (rt_kernels.hip: sphere_isect,
    `coherent`: a conservative bound on val3 decides that every ray of a wave misses before the exact test runs) - a hit
    it dropped would change a pixel from sphere to background and the ray counts with it. Frames and ray counts against
    the oracle; every silhouette in a 640 x 480 frame is a few thousand grazing rays."""
    O = oracle_mod
    rng = np.random.default_rng(seed)
    fs = wl.FlatScene()
    mats = wl.yaml_materials()
    names = ["mirror", "m1", "glass", "red", "green", "blue"]
    for k in range(10):
        r = float(10.0 ** rng.uniform(-2.0, 1.5))
        d = float(10.0 ** rng.uniform(0.0, 2.3))
        ang = rng.uniform(-0.6, 0.6, 2)
        c = (d * np.sin(ang[0]), d * np.sin(ang[1]), -d * np.cos(ang[0]) * np.cos(ang[1]) * (1.0 if k else -1.0))
        fs.add_sphere(tuple(float(x) for x in c), r, mats[names[k % len(names)]])
    fs.add_sphere((0.0, 0.0, 0.0), 250.0, mats["green"])      # the camera sits inside this one
    wl.add_yaml_lights(fs)
    cam, p = reference_camera(640, 480), Params(max_depth=3)
    got, st1 = DeviceScene(fs).render(cam, p)
    want, st2 = O.Oracle(fs).render(cam, p)
    within1, equal, worst = frame_stats(got, want)
    print(f"seed {seed}: within 1 LSB {within1:.6f}, equal {equal:.6f}, worst {worst}, rays {st1.trace_calls}+{st1.shadow_rays}")
    assert st1.trace_calls == st2.trace_calls and st1.shadow_rays == st2.shadow_rays
    assert list(st1.level_rays) == list(st2.level_rays)
    assert within1 >= 0.9999 and equal >= 0.9999


def test_schedule_forms_agree(oracle_mod):
    """The same frame through the different launch schedules: first call of a scene (level sizes unknown: per-lane
    kernels + resumed queues at every level), second call (small levels run as one quad kernel), the asynchronous
    multi-stream path into a device buffer, and the oracle."""
    import torch
    fs = wl.scene_bunny(10, third="glass2")
    cam, p = reference_camera(320, 240), Params(max_depth=6)
    dev = DeviceScene(fs)
    a, st_a = dev.render(cam, p)
    b, st_b = dev.render(cam, p)
    assert np.array_equal(a, b)
    assert st_a.trace_calls == st_b.trace_calls and st_a.shadow_rays == st_b.shadow_rays
    assert list(st_a.level_rays) == list(st_b.level_rays)
    frame = torch.zeros((240, 320, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):     # async: no stats, shade kernels on their own streams
        dev.render_device(cam, p, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), a)
    want, st_o = oracle_mod.Oracle(fs).render(cam, p)
    within1, equal, worst = frame_stats(a, want)
    assert within1 >= 0.999 and equal >= 0.9999
    assert st_a.trace_calls == st_o.trace_calls and st_a.shadow_rays == st_o.shadow_rays


@pytest.mark.parametrize("spp_grid", [1, 4])
def test_traversal_forms_agree(spp_grid):
    """Every way the library can take a frame through a mesh gives the same bytes and ray counts: level 0 by marks +
    quads (option "packet" 0, rt_scene_set_option), as 64-ray packets (1), packets at the deeper levels too (2); bounce
    levels as per-lane kernels + queues + resuming quads ("large_threshold" 0: every level "large"), as one quad launch
    per level (huge threshold), or as per-lane kernels whose quads serve the rays that reach the mesh ("serve" 1); the
    deep levels one launch each ("tail_rays" 0) or all of them - closest hits, light loops, folds - in one launch
    (k_tail: from level 1 on with a huge threshold, from where the levels hold <= 300 rays with 300).
    The defaults pick among these by samples per pixel and level size; test_frame_parity pins the default to the
    oracle."""
    fs = wl.scene_bunny(10, third="glass2")
    cam, p = reference_camera(320, 240), Params(max_depth=6, aa=wl.aa_grid(spp_grid) if spp_grid > 1 else ())
    frames = {}
    for packet, thr, serve, split, tail in (
            ("0", "0", "0", "1", "0"), ("0", "1000000000", "0", "1", "0"), ("1", "0", "0", "1", "0"),
            ("1", "1000000000", "0", "1", "0"), ("1", "1000000000", "1", "1", "0"), ("1", "1000000000", "0", "0", "0"),
            ("2", "0", "0", "1", "0"), ("2", "1000000000", "0", "1", "0"),
            ("1", "1000000000", "0", "1", "1000000000"), ("1", "1000000000", "0", "1", "300"),
            ("1", "1000000000", "1", "1", "300"), ("0", "0", "0", "1", "300"), ("2", "0", "0", "1", "300")):
        dev = DeviceScene(fs)
        dev.set_option("packet", int(packet))            # how level 0 (1) / every large level (2) goes through the mesh
        dev.set_option("tail_rays", int(tail))           # levels of at most this many rays go to k_tail
        dev.set_option("large_threshold", int(thr))      # rays from which a bounce level is "large"
        dev.set_option("serve", int(serve))              # bounce levels one ray per lane with quad service / one ray per quad
        dev.set_option("split_small", int(split))        # small levels shaded per lane + resuming quads / one quad launch per run
        dev.set_option("tiny_run", 1024)                 # (runs of more hits than this take the split path: the default, 8192, would leave few at this size)
        dev.render(cam, p)                               # (the second frame of a scene knows its level sizes)
        img, st = dev.render(cam, p)
        frames[(packet, thr, serve, split, tail)] = (img, st.trace_calls, st.shadow_rays, list(st.level_rays))
    base = frames[("1", "1000000000", "0", "1", "0")]
    for key, f in frames.items():
        assert np.array_equal(f[0], base[0]), key
        assert f[1:] == base[1:], key
    default_img, st = DeviceScene(fs).render(cam, p)
    assert np.array_equal(default_img, base[0]) and st.trace_calls == base[1]
    # option "run_hits": a run of small levels is shaded as soon as it holds that many hits (default 2^20: batches of
    # frames) instead of when three levels have come together - here: every level by itself, with and without quad service
    for serve in (0, 1):
        dev = DeviceScene(fs)
        dev.set_option("run_hits", 1)
        dev.set_option("serve", serve)
        dev.render(cam, p)
        img, st = dev.render(cam, p)
        assert np.array_equal(img, base[0]) and (st.trace_calls, st.shadow_rays, list(st.level_rays)) == base[1:]


@pytest.mark.parametrize("gpow", [False, True])
def test_fused_levels_agree(oracle_mod, gpow):
    """A scene without meshes runs each recursion level as ONE launch (k_trace_wide: closest hits, child rays and the light
    loops of the hits, the hit record never leaving the registers; option "fuse", on by default). Frames, ray counts and
    level sizes equal those of the two-launch path (option "fuse" 0: k_extend_wide + k_shade_wide) and the oracle's, with
    reflection and refraction down to depth 10, at 1 and 9 samples per pixel, with scene.yaml's integer exponents and with
    the device pow, through the synchronous call and the asynchronous one into a device buffer."""
    import torch
    fs = wl.scene_analytic(third="glass2")
    if gpow:
        for m in fs.materials:
            m.specRefExp = 7.5
    cam = reference_camera(400, 300)
    for p in (Params(), Params(aa=DEFAULT_AA, max_depth=6)):
        fused, plain = DeviceScene(fs), DeviceScene(fs)
        plain.set_option("fuse", 0)
        a, st_a = fused.render(cam, p)
        b, st_b = plain.render(cam, p)
        assert np.array_equal(a, b)
        assert (st_a.trace_calls, st_a.shadow_rays, list(st_a.level_rays)) == (st_b.trace_calls, st_b.shadow_rays, list(st_b.level_rays))
        frame = torch.zeros((300, 400, 3), dtype=torch.uint8, device="cuda")
        for _ in range(2):
            fused.render_device(cam, p, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        fused.status()
        torch.cuda.synchronize()
        assert np.array_equal(frame.cpu().numpy(), a)
        want, st_o = oracle_mod.Oracle(fs).render(cam, p)
        within1, equal, worst = frame_stats(a, want)
        assert within1 >= 0.999 and equal >= 0.9999
        assert st_a.trace_calls == st_o.trace_calls and st_a.shadow_rays == st_o.shadow_rays


@pytest.mark.parametrize("spheres", [False, True])
def test_fused_level0_packets_agree(oracle_mod, spheres):
    """A scene WITH a mesh none of whose materials bounces (BASELINE config 3: Phong + hard shadows): level 0 is the whole
    frame and, as packets, suspends nothing - closest hits and light loops run in one launch (k_trace_wide<0, false, PACKET>;
    option "fuse"). Frames and ray counts equal the two-launch path's (option "fuse" 0, with and without the expensive-groups
    order) and the oracle's, at 1 and 4 samples per pixel, single frames, a batch, and the asynchronous call."""
    import torch
    fs = wl.scene_bunny(10, shadows_only=True, spheres=spheres)
    cam = reference_camera(448, 320)      # 2240 groups of 64 samples: the two-launch path orders them
    for p in (Params(), Params(aa=wl.aa_grid(2))):
        fused, plain = DeviceScene(fs), DeviceScene(fs)
        plain.set_option("fuse", 0)
        a, st_a = fused.render(cam, p)
        for _ in range(3):      # (the second and third frame of `plain` follow the first one's costs)
            b, st_b = plain.render(cam, p)
            assert np.array_equal(a, b)
            assert (st_a.trace_calls, st_a.shadow_rays) == (st_b.trace_calls, st_b.shadow_rays)
        frames = torch.zeros((3, 320, 448, 3), dtype=torch.uint8, device="cuda")
        stream = torch.cuda.Stream()
        fused.render_batch_device([cam] * 3, p, frames.data_ptr(), 320 * 448 * 3, stream=stream.cuda_stream)
        fused.status()
        torch.cuda.synchronize()
        for k in range(3):
            assert np.array_equal(frames[k].cpu().numpy(), a)
        want, st_o = oracle_mod.Oracle(fs).render(cam, p)
        within1, equal, worst = frame_stats(a, want)
        assert within1 >= 0.999 and equal >= 0.9999
        assert st_a.trace_calls == st_o.trace_calls and st_a.shadow_rays == st_o.shadow_rays


def test_expensive_groups_first_agree():
    """Option "order0" (on by default): the level-0 launches of a small one-pass frame hand out their 64-sample groups in
    descending order of what they cost in the scene's previous frame, the most expensive as two halves or four quarters
    (k_order_groups). Only which wave works on which entries changes: frame after frame - each ordered by its predecessor's
    costs, with and without a wait in between, single frames and a batch, after a change of frame size - equals the frames of
    a scene that never orders, and so do the ray counts."""
    import torch
    fs = wl.scene_bunny(10, third="glass2")
    p = Params(max_depth=6)
    plain = DeviceScene(fs)
    plain.set_option("order0", 0)
    dev = DeviceScene(fs)
    stream = torch.cuda.Stream()
    for (w, h) in ((512, 384), (640, 480), (512, 384)):          # 3072 / 4800 groups: the arrays grow, then an order of another size is stale
        cam = reference_camera(w, h)
        want, st_w = plain.render(cam, p)
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        for k in range(7):
            with torch.cuda.stream(stream):
                out.zero_()
                dev.render_device(cam, p, out.data_ptr(), stream=stream.cuda_stream)
                if k % 2:
                    torch.cuda.synchronize()      # with it the next frame finds the order finished, without it maybe not
                stream.synchronize()
            assert np.array_equal(out.cpu().numpy(), want), (w, h, k)
        img, st = dev.render(cam, p)                              # (the timed path: every launch on one stream)
        assert np.array_equal(img, want)
        assert (st.trace_calls, st.shadow_rays, list(st.level_rays)) == (st_w.trace_calls, st_w.shadow_rays, list(st_w.level_rays))
    cams = wl.orbit_cameras(512, 384, 4)
    ref = torch.zeros((4, 384, 512, 3), dtype=torch.uint8, device="cuda")
    out = torch.zeros_like(ref)
    with torch.cuda.stream(stream):
        plain.render_batch_device(cams, p, ref.data_ptr(), 384 * 512 * 3, stream=stream.cuda_stream)
        for k in range(4):
            out.zero_()
            dev.render_batch_device(cams, p, out.data_ptr(), 384 * 512 * 3, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            assert bool(torch.equal(out, ref)), k


@pytest.mark.parametrize("bits", [2, 8, 10])
def test_sorted_light_loops_agree(bits):
    """Option "sort_shade" (off by default: measured slower): the hits of a bounce level sorted by the cell of their hit point
    before their shadow packets. Every hit is still shaded exactly once: same bytes and ray counts as the pool order, with
    coarse and fine cells, on the second frame of a scene (the sort is sized by the previous frame's level sizes) and on a
    deeper frame behind a shallow one (levels that outgrow the estimate are walked in pool order behind the sorted part)."""
    fs = wl.scene_bunny(10, third="glass2")
    cam, p16 = reference_camera(160, 120), Params(max_depth=6, aa=wl.aa_grid(4))
    want, st_w = DeviceScene(fs).render(cam, p16)
    dev = DeviceScene(fs)
    dev.set_option("packet", 2)
    dev.set_option("large_threshold", 0)       # every bounce level runs the per-lane kernels, i.e. as packets
    dev.set_option("sort_shade", 1)
    dev.set_option("sort_bits", bits)
    dev.render(cam, Params(max_depth=2, aa=wl.aa_grid(4)))      # leaves hints for levels 0..2 only
    for _ in range(2):
        img, st = dev.render(cam, p16)
        assert np.array_equal(img, want)
        assert (st.trace_calls, st.shadow_rays, list(st.level_rays)) == (st_w.trace_calls, st_w.shadow_rays, list(st_w.level_rays))
    small = reference_camera(64, 48)                              # far fewer hits than the estimate
    img, st = dev.render(small, p16)
    assert np.array_equal(img, DeviceScene(fs).render(small, p16)[0])
    if bits == 8:       # option "shade_serve_level" (off by default too): those light loops by quad service instead of packets
        dev2 = DeviceScene(fs)
        dev2.set_option("packet", 2)
        dev2.set_option("large_threshold", 0)
        dev2.set_option("shade_serve_level", 1)
        dev2.render(cam, p16)
        img, st = dev2.render(cam, p16)
        assert np.array_equal(img, want) and (st.trace_calls, st.shadow_rays) == (st_w.trace_calls, st_w.shadow_rays)


def test_sphere_roots_exactly_zero(oracle_mod):
    """Sphere::intersect with both roots exactly 0 (reference src/primitives.cpp:188-192 takes neither branch and `dist`
    keeps the caller's value, defined as 0 at its first use here and in the oracle): a miss on the sphere, the plane
    behind it is found, in every traversal form's kernels; the oracle's counter shows the branch was the one taken."""
    from test_oracle_golden import tangent_scene_and_rays
    fs, rays = tangent_scene_and_rays()
    oracle_mod.stale_reads(reset=True)
    h2, d2, n2 = oracle_mod.Oracle(fs).intersect(rays, -1)
    assert oracle_mod.stale_reads(reset=True) == len(rays)
    dev = DeviceScene(fs)
    for form in (0, 1, 2):
        h1, d1, n1 = dev.intersect(rays, -1, form=form)
        assert np.array_equal(h1, h2)
        hit = h2 >= 0
        assert np.array_equal(d1[hit].view(np.uint32), d2[hit].view(np.uint32))
    assert (dev.intersect(rays, 0)[0] == -1).all()


def test_scanned_like_ply_frames(oracle_mod, tmp_path):
    """A .ply with what scanned meshes bring - zero-area faces (NaN normals: trace()'s red exit where a ray hits next to
    them), an unused vertex, a face listed in both orientations - loaded by the host surface, as a reflective mesh in
    the scene.yaml setting: frame and ray counts against the oracle, closest hits bit for bit in every traversal form."""
    from test_oracle_vs_ref import scanned_like_mesh
    from raytracer_amd.scene import FlatScene, Mesh, material
    mesh = Mesh.load_ply(scanned_like_mesh(tmp_path), rotation=(0, 0.5236, 0), scale=1.6, translation=(-0.5, -0.3, -5.5))
    assert np.isnan(np.stack(mesh.v[3:])).any()
    fs = FlatScene()
    fs.add_sphere((1.2, 0.4, -4.0), 0.5, material((0.9, 0.9, 0.9), reflective=True, kr=0.8))
    fs.add_mesh(mesh, material((0.8, 0.7, 0.6), reflective=True, kr=0.4))
    fs.add_plane((0, 1, 0), 2.0, material((0.5, 0.5, 0.9), reflective=True, kr=0.3))
    wl.add_yaml_lights(fs)
    orc = oracle_mod.Oracle(fs)
    dev = DeviceScene(fs)
    cam, p = reference_camera(200, 150), Params(max_depth=5)
    got, st1 = dev.render(cam, p)
    want, st2 = orc.render(cam, p)
    within1, equal, worst = frame_stats(got, want)
    assert (st1.trace_calls, st1.shadow_rays) == (st2.trace_calls, st2.shadow_rays)
    assert within1 >= 0.999 and equal >= 0.9999, (within1, equal, worst)
    rays = np.concatenate([camera_rays(oracle_mod, cam, Params(), 1), edge_rays(fs, 300, 3)])
    h2, d2, n2 = orc.intersect(rays, -1)
    hit = h2 >= 0
    for form in (0, 1, 2):
        h1, d1, n1 = dev.intersect(rays, -1, form=form)
        assert np.array_equal(h1, h2)
        assert np.array_equal(d1[hit].view(np.uint32), d2[hit].view(np.uint32))
        assert np.array_equal(n1[hit].view(np.uint32), n2[hit].view(np.uint32))


def test_nan_normal_red_exit(oracle_mod):
    """trace()'s `return red` exit (reference src/raytracer.cpp:330-332): a reflective mesh whose vertex normals are NaN
    (what calcNormal gives for a degenerate face). The HIP path appends the reflection ray before the light loop has
    decided; colours AND ray counts must still equal the oracle's."""
    from raytracer_amd.scene import FlatScene, Mesh, material
    fs = FlatScene()
    nan = float("nan")
    tri = [[[-3, -2, -6]], [[3, -2, -6.5]], [[0, 3, -5.5]], [[nan, 0, 1]], [[0, nan, 1]], [[0, 0, nan]]]
    fs.add_mesh(Mesh(*[np.array(a, np.float32) for a in tri], np.array([0, 0, -6, 5.0], np.float32)),
                material((0.8, 0.8, 0.8), reflective=True, kr=0.5))
    fs.add_sphere((0.5, 0.2, -3.0), 0.4, material((0.2, 0.9, 0.2)))          # shadows part of the triangle
    fs.add_plane((0, 1, 0), 4.0, material((0.5, 0.5, 0.9), reflective=True, kr=0.3))
    fs.add_light((-2, 5, 0), (1, 1, 1), 30.0)
    fs.add_light((3, 4, -1), (1, 0.9, 0.8), 20.0)
    cam, p = reference_camera(96, 72), Params(max_depth=4)
    rays = camera_rays(oracle_mod, cam, p)
    dev = DeviceScene(fs)
    got, st1 = dev.trace(p, rays)
    want, st2 = oracle_mod.Oracle(fs).trace(p, rays)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.nan_to_num(got, nan=-1.0), np.nan_to_num(want, nan=-1.0))
    assert (got == np.array([1, 0, 0], np.float32)).all(axis=1).any()            # the red exit was taken somewhere
    assert st1.trace_calls == st2.trace_calls and st1.shadow_rays == st2.shadow_rays
    f1, s1 = dev.render(cam, p)
    f2, s2 = oracle_mod.Oracle(fs).render(cam, p)
    assert np.array_equal(f1, f2)
    assert s1.trace_calls == s2.trace_calls and s1.shadow_rays == s2.shadow_rays


def hall_of_mirrors():
    """Everything bounces: two reflective + refractive spheres between mirror walls. The ray tree of a sample has
    hundreds of nodes, far more than the 6 the pool is sized for."""
    fs = wl.FlatScene()
    mats = wl.yaml_materials()
    fs.add_sphere((-1.2, 0.0, -6.0), 1.0, mats["glass2"])
    fs.add_sphere((1.2, 0.3, -7.0), 1.0, mats["glass2"])
    for nrm, d in [((0, 0, 1), 12.0), ((1, 0, 0), 4.0), ((-1, 0, 0), 4.0), ((0, 1, 0), 4.0), ((0, -1, 0), 4.0)]:
        fs.add_plane(nrm, d, mats["mirror"])
    wl.add_yaml_lights(fs)
    return fs


def test_pool_overflow_is_rerun_in_smaller_passes(oracle_mod):
    """Synchronous calls re-run a pass whose ray trees outgrow the pool in smaller pieces; the frame and the ray counts
    still equal the reference's recursion."""
    O = oracle_mod
    fs = hall_of_mirrors()
    cam = reference_camera(96, 72)
    p = Params(max_depth=8)
    dev = DeviceScene(fs)
    got, st1 = dev.render(cam, p)
    want, st2 = O.Oracle(fs).render(cam, p)
    within1, equal, worst = frame_stats(got, want)
    print(f"hall of mirrors: {st1.trace_calls} trace calls for {96 * 72} samples, passes {st1.passes}, retries {st1.retries}, "
          f"within 1 LSB {within1:.6f}")
    assert st1.retries > 0 and st1.passes > 1
    assert st1.trace_calls == st2.trace_calls and st1.shadow_rays == st2.shadow_rays
    assert within1 >= 0.999 and equal >= 0.9999
    # the scene remembers: the same frame again needs no retry
    again, st3 = dev.render(cam, p)
    assert st3.retries == 0 and np.array_equal(again, got)


def test_async_overflow_is_reported_then_fits():
    """An asynchronous frame cannot re-run passes; rt_scene_status reports the overflow, the scene shrinks its passes
    and the frame rendered again equals the synchronous one."""
    import torch
    from raytracer_amd.cabi import RtError
    fs = hall_of_mirrors()
    cam = reference_camera(96, 72)
    p = Params(max_depth=8)
    want, _ = DeviceScene(fs).render(cam, p)
    dev = DeviceScene(fs)
    frame = torch.zeros((72, 96, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()       # torch's fill is not ordered against the scene's own stream
    ok = False
    for attempt in range(12):
        dev.render_device(cam, p, frame.data_ptr())
        try:
            dev.status()
            ok = True
            break
        except RtError as e:
            assert "pool" in str(e)
    assert ok and attempt >= 1
    assert np.array_equal(frame.cpu().numpy(), want)
