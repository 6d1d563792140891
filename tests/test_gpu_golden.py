"""The HIP path against the committed golden vectors of the reference itself (tests/golden), through the C-ABI,
plus the multi-GPU tile path on one GPU, the error behaviour of the boundary, and size-independent properties at
the bench's full size."""
import ctypes as C

import numpy as np
import pytest

from helpers import gold, same_bits, sub
from raytracer_amd import cabi, tiling, workloads as wl
from raytracer_amd.scene import DeviceScene, FlatScene, Params, camera_from_array, cast_rays, reference_camera

pytestmark = pytest.mark.gpu


def test_camera_rays_golden():
    g = gold("camera_rays.npz")
    for ci in range(4):
        cam = camera_from_array(g[f"cam{ci}"])
        assert same_bits(cast_rays(cam, Params(), g[f"xy{ci}"]), g[f"rays{ci}_1"])
        assert same_bits(cast_rays(cam, Params(aa=g["aa9"]), g[f"xy{ci}"]), g[f"rays{ci}_9"])


def test_intersections_golden():
    g = gold("intersect.npz")
    fs = FlatScene.from_arrays(sub(g, "scene_"))
    dev = DeviceScene(fs)
    rays = g["rays"]
    for obj in [-1] + list(range(len(fs.objects))):
        h, d, n = dev.intersect(rays, obj)
        assert np.array_equal(h, g[f"hit_{obj}"]), f"object {obj}: {(h != g[f'hit_{obj}']).sum()} rays differ"
        hit = h >= 0
        assert same_bits(d[hit], g[f"dist_{obj}"][hit]) and same_bits(n[hit], g[f"normal_{obj}"][hit])
    assert np.array_equal(dev.occluded(rays, g["max_dist"]), g["occluded"])
    lone = DeviceScene(FlatScene.from_arrays(sub(g, "lone_")))
    h, d, n = lone.intersect(rays, -1)
    assert np.array_equal(h, g["lone_hit"])
    hit = h >= 0
    assert same_bits(d[hit], g["lone_dist"][hit]) and same_bits(n[hit], g["lone_normal"][hit])


@pytest.mark.parametrize("name", ["analytic", "glass2", "mesh"])
def test_frames_golden(name):
    g = gold("frames.npz")
    fs = FlatScene.from_arrays(sub(g, f"{name}_scene_"))
    dev = DeviceScene(fs)
    cam = camera_from_array(g["cam_small"])
    worst = 1.0
    for key, p in [(f"{name}_d{d}_1", Params(max_depth=d)) for d in (0, 1, 8, 10)] + \
                  [(f"{name}_d10_9", Params(aa=g["aa9"]))]:
        got, _ = dev.render(cam, p)
        d = np.abs(got.astype(np.int16) - g[key].astype(np.int16)).max(axis=2)
        worst = min(worst, float((d <= 1).mean()))
        assert (d <= 1).mean() >= 0.999 and (d == 0).mean() >= 0.9999, key      # north_star's bar; what is measured: every pixel equal
    print(f"{name}: worst within-1-LSB share over 5 frames {worst:.6f}")
    if name != "mesh":
        got, st = dev.render(camera_from_array(g["cam_large"]), Params())
        assert abs(int(got.astype(np.int64).sum()) - int(g[f"{name}_800x600_bytesum"])) <= 480
        assert np.abs(got.astype(np.int64).sum(axis=(1, 2)) - g[f"{name}_800x600_rowsum"]).max() <= 8


def test_tiles_on_one_gpu_equal_the_full_frame():
    """N-part tile render + untile is byte-identical to the single render (SURVEY.md 8e invariant)."""
    import torch
    fs = wl.scene_bunny(10, third="glass2")
    dev = DeviceScene(fs)
    lib = dev.lib
    cam = reference_camera(400, 296)      # 296 = 9*32 + 8: ragged bottom edge; 400 = 12*32 + 16
    p = Params(max_depth=6, aa=((0.5, 0.5),))
    full, _ = dev.render(cam, p)
    W, H = cam.imWidth, cam.imHeight
    for (tw, th, n_parts) in [(32, 32, 3), (64, 16, 8), (32, 32, 1)]:
        slot = lib.rt_tiles_owned(W, H, tw, th, 0, n_parts)
        allt = torch.zeros((n_parts, slot * tw * th * 3), dtype=torch.uint8, device="cuda")
        frame = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
        for part in range(n_parts):
            dev.render_tiles_device(cam, p, (tw, th), part, n_parts, allt[part].data_ptr())
        torch.cuda.synchronize()
        cabi.check(lib, lib.rt_untile_device(0, W, H, tw, th, n_parts, slot, C.c_void_p(allt.data_ptr()),
                                             C.c_void_p(frame.data_ptr()), None))
        torch.cuda.synchronize()
        got = frame.cpu().numpy()
        assert np.array_equal(got, full), (tw, th, n_parts)
        # the host-side mirror agrees with the device scatter
        assert np.array_equal(tiling.untile([a.cpu().numpy() for a in allt], W, H, tw, th), full)


def test_index_arithmetic_paths_agree():
    """Sample index -> frame / pixel / tile: where a pass' indices fit 32 bits the kernels divide by reciprocal
    multiplication (udivmod32), else by the 64-bit expansion (frames beyond 2^32 samples, too large to test).
    Option "index64" (rt_scene_set_option) forces the latter: full frame at 3 samples per pixel (a divisor that is not a power of two), a
    rectangle, tiles in 3 parts and a batch of 3 frames give the same bytes both ways."""
    import torch
    fs = wl.scene_bunny(10, third="glass2")
    cam = reference_camera(400, 296)
    from raytracer_amd.scene import make_camera
    cams = [cam] + [make_camera(400, 296, eye=(0.3 * k, 0.1 * k, 0.2 * k), at=(0.05 * k, 0, -1)) for k in (1, 2)]
    p = Params(max_depth=3, aa=((0.5, 0.5), (0.25, 0.75)))
    W, H = cam.imWidth, cam.imHeight
    got = {}
    for mode in ("fast", "wide"):
        dev = DeviceScene(fs)
        dev.set_option("index64", 1 if mode == "wide" else 0)
        full, st = dev.render(cam, p)
        rect, _ = dev.render(cam, p, rect=(37, 21, 203, 150))
        slot = dev.lib.rt_tiles_owned(W, H, 32, 32, 0, 3)
        allt = torch.zeros((3, slot * 32 * 32 * 3), dtype=torch.uint8, device="cuda")
        batch = torch.zeros((3, H, W, 3), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for part in range(3):
            dev.render_tiles_device(cam, p, (32, 32), part, 3, allt[part].data_ptr())
        dev.render_batch_device(cams, p, batch.data_ptr(), H * W * 3)
        dev.status()
        torch.cuda.synchronize()
        got[mode] = (full, st.trace_calls, rect, allt.cpu().numpy(), batch.cpu().numpy())
    a, b = got["fast"], got["wide"]
    assert a[1] == b[1]
    for k in (0, 2, 3, 4):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a[4][0], a[0]) and not np.array_equal(a[4][1], a[0])


def test_boundary_errors():
    fs = wl.scene_analytic()
    dev = DeviceScene(fs)
    lib = dev.lib
    cam = reference_camera(64, 48)
    out = np.zeros((48, 64, 3), np.uint8)
    p = Params()
    rc = lib.rt_render(dev.h, C.byref(cam), C.byref(p.c), 60, 0, 10, 10, cabi.u8ptr(out), None)
    assert rc == -1 and b"outside" in lib.rt_last_error()
    p16 = Params(max_depth=16)
    assert lib.rt_render(dev.h, C.byref(cam), C.byref(p16.c), 0, 0, 64, 48, cabi.u8ptr(out), None) == -5
    assert lib.rt_render(None, C.byref(cam), C.byref(p.c), 0, 0, 64, 48, cabi.u8ptr(out), None) == -1
    bad = wl.scene_analytic()
    bad.objects[0] = (0, 99, 0)
    h = C.c_void_p()
    assert lib.rt_scene_create(C.byref(bad.c_struct()), 0, C.byref(h)) == -1
    assert lib.rt_scene_create(C.byref(fs.c_struct()), 77, C.byref(h)) == -1
    # empty rectangle and empty scene are fine: nothing written / background everywhere
    got, st = dev.render(cam, p, rect=(5, 5, 0, 0))
    assert not got.any() and st.trace_calls == 0
    empty = DeviceScene(FlatScene())
    got, st = empty.render(cam, p)
    assert (got == np.array([127, 127, 0], np.uint8)).all() and st.trace_calls == 64 * 48 and st.shadow_rays == 0


def _oracle_pixels(O, fs, cam, p, pixels):
    """render()'s value of single pixels from the oracle's trace() of their samples (reference src/raytracer.cpp:436-448:
    px += trace(...) / numRays per sample, `/` = multiply by the float reciprocal, color.h:27-30; then B,G,R truncated).
    For scenes where a ray costs the oracle 0.4 s (10 M faces, every one tested): orc_trace spreads the SAMPLES of a
    pixel over the host threads, orc_render only spreads rows."""
    rays = O.cast_rays("orc", cam, p, pixels)                    # (n, spp, 6)
    rgb, st = O.Oracle(fs).trace(p, rays.reshape(-1, 6))
    rgb = rgb.reshape(len(pixels), p.spp, 3)
    inv = np.float32(1) / np.float32(p.spp)
    px = np.zeros((len(pixels), 3), np.float32)
    for k in range(p.spp):
        px = px + rgb[:, k] * inv
    return (px[:, ::-1] * np.float32(255)).astype(np.uint8), st


# workload -> how the oracle checks it: ("frame",) the whole frame; ("rects", [...]) crops through orc_render;
# ("pixels", [...]) single pixels through orc_trace. Sized to a few CPU-seconds each.
FULL_SIZE = {
    "bunny_1080p": ("rects", [(692, 0, 16, 1080), (1100, 300, 16, 480)]),  # strips through the mesh and the mirror sphere; through the glass sphere
    "c2_analytic_1080p_4spp": ("frame",),
    "c3_bunny_1080p_shadows": ("rects", [(692, 0, 16, 1080)]),
    "c4_bunny_glass_4k_16spp": ("rects", [(1088, 272, 48, 32), (1580, 1390, 32, 32)]),     # in the glass2 sphere; on the mesh
    # 8 pixels on the blobs of the field, then 24 SILHOUETTE pixels: the corner rays of pixel x and x + 1 of that row hit
    # different things (blob / floor; found with the oracle at depth 0), so the 64 samples of pixel x straddle the edge
    "c5_field_10m_8k_64spp": ("pixels", [(3840, 2721), (2560, 3456), (5000, 3000), (3000, 2400), (4400, 3900), (1200, 2900), (3846, 2727), (2553, 3449),
                                         (2813, 2400), (2888, 2400), (2975, 2400), (3054, 2400), (3094, 2400), (3152, 2400), (3251, 2400), (3325, 2400),
                                         (3368, 2400), (3475, 2721), (3543, 2721), (3640, 2721), (3705, 2721), (3801, 2721), (3849, 2721), (3954, 2721),
                                         (2361, 3456), (2462, 3456), (2557, 3456), (2653, 3456), (2768, 3456), (2869, 3456), (4428, 3900), (4488, 3900)]),
}


@pytest.mark.parametrize("workload", list(FULL_SIZE))
def test_full_size_properties(workload):
    """Every BASELINE.json configuration at ITS size (resolution, samples per pixel, depth, triangle count). The oracle
    is too slow for whole frames there, so: determinism, rectangle decomposition == full frame, ray-count identities,
    and a sample of the frame against the oracle (bytes within 1 LSB on >= 99.9 %, ray counts equal)."""
    from oracle import oracle as O
    fs, cam, p = wl.make_workload(workload)
    dev = DeviceScene(fs)
    a, st = dev.render(cam, p)
    b, st2 = dev.render(cam, p)
    assert np.array_equal(a, b) and st.trace_calls == st2.trace_calls and st.shadow_rays == st2.shadow_rays
    W, H = cam.imWidth, cam.imHeight
    assert st.level_rays[0] == W * H * p.spp and st.trace_calls == sum(st.level_rays)
    assert st.shadow_rays % len(fs.lights) == 0 and st.shadow_rays <= len(fs.lights) * st.trace_calls
    assert st.passes == -(-(W * H) // max(64, dev.pass_samples() // p.spp)) and st.retries == 0
    if not any(m.reflective or m.refractive for m in fs.materials):
        assert st.trace_calls == st.level_rays[0]                       # "shadows only": nothing bounces
    parts = np.zeros_like(a)
    rays = 0
    for rect in [(0, 0, W // 2 + 40, H // 2 - 40), (W // 2 + 40, 0, W - (W // 2 + 40), H // 2 - 40), (0, H // 2 - 40, W, H - (H // 2 - 40))]:
        _, s = dev.render(cam, p, rect=rect, out=parts)
        rays += s.trace_calls + s.shadow_rays
    assert np.array_equal(parts, a) and rays == st.trace_calls + st.shadow_rays
    kind = FULL_SIZE[workload]
    if kind[0] == "frame":
        want, so = O.Oracle(fs).render(cam, p, threads=16)
        d = np.abs(a.astype(np.int16) - want.astype(np.int16)).max(axis=2)
        assert so.trace_calls == st.trace_calls and so.shadow_rays == st.shadow_rays
    elif kind[0] == "rects":
        ds = []
        for rect in kind[1]:
            x0, y0, w, h = rect
            want, so = O.Oracle(fs).render(cam, p, rect=rect, threads=16)
            _, sg = dev.render(cam, p, rect=rect)
            assert so.trace_calls == sg.trace_calls and so.shadow_rays == sg.shadow_rays, rect
            ds.append(np.abs(a[y0:y0 + h, x0:x0 + w].astype(np.int16) - want[y0:y0 + h, x0:x0 + w].astype(np.int16)).max(axis=2).ravel())
        d = np.concatenate(ds)
    else:
        want, so = _oracle_pixels(O, fs, cam, p, kind[1])
        got = np.stack([a[y, x] for x, y in kind[1]])
        rays_g = 0
        for x, y in kind[1]:
            _, sg = dev.render(cam, p, rect=(x, y, 1, 1))
            rays_g += sg.trace_calls + sg.shadow_rays
        assert rays_g == so.trace_calls + so.shadow_rays
        assert so.trace_calls > len(kind[1]) * p.spp                    # the sampled pixels do hit something that bounces
        d = np.abs(got.astype(np.int16) - want.astype(np.int16)).max(axis=1)
    print(f"{workload}: {st.trace_calls}+{st.shadow_rays} rays, {st.passes} passes; oracle sample of {d.size} px: "
          f"within 1 LSB {(d <= 1).mean():.6f}, equal {(d == 0).mean():.6f}")
    assert (d <= 1).mean() >= 0.999 and (d == 0).mean() >= 0.9999


@pytest.mark.parametrize("spp_grid", [1, 2])
def test_many_passes_equal_one_pass(spp_grid):
    """The ordinary multi-pass path (a frame of more samples than a pass holds, cut into passes of whole pixels that
    reuse one workspace) against the oracle and against the same frame in one pass: 2^12-sample passes make a 320x240
    frame 19 (1 spp) or 75 (4 spp) passes. Synchronous (stats, one stream) and asynchronous (multi-stream) calls."""
    import torch
    from oracle import oracle as O
    fs = wl.scene_bunny(10, third="glass2")
    cam = reference_camera(320, 240)
    p = Params(max_depth=6, aa=wl.aa_grid(spp_grid) if spp_grid > 1 else ())
    dev = DeviceScene(fs)
    one, st1 = dev.render(cam, p)
    assert st1.passes == 1
    dev.set_pass_samples(1 << 12)
    many, st2 = dev.render(cam, p)
    assert st2.passes == -(-320 * 240 // ((1 << 12) // p.spp)) and st2.retries == 0
    assert np.array_equal(one, many)
    assert (st1.trace_calls, st1.shadow_rays, list(st1.level_rays)) == (st2.trace_calls, st2.shadow_rays, list(st2.level_rays))
    frame = torch.zeros((240, 320, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        dev.render_device(cam, p, frame.data_ptr())
    dev.status()
    assert np.array_equal(frame.cpu().numpy(), one)
    want, so = O.Oracle(fs).render(cam, p, threads=16)
    d = np.abs(many.astype(np.int16) - want.astype(np.int16)).max(axis=2)
    assert so.trace_calls == st2.trace_calls and so.shadow_rays == st2.shadow_rays
    assert (d <= 1).mean() >= 0.999 and (d == 0).mean() >= 0.9999
    with pytest.raises(cabi.RtError):
        dev.set_pass_samples(5000)                                        # not a power of two
    dev.set_pass_samples(0)
    again, st3 = dev.render(cam, p)
    assert st3.passes == 1 and np.array_equal(again, one)


def test_batch_equals_single_frames():
    """rt_render_batch_device / rt_render_tiles_batch_device: n frames with different cameras in one call are, byte for
    byte, the n frames rendered one by one (and the ray counts add up)."""
    import torch
    from raytracer_amd.scene import make_camera
    from raytracer_amd.cabi import rt_stats
    fs = wl.scene_bunny(10, third="glass2")
    dev = DeviceScene(fs)
    W, H = 328, 200
    cams = [make_camera(W, H, eye=(0.3 * k, 0.1 * k, 0.2 * k), at=(0.05 * k, 0, -1)) for k in range(5)]
    p = Params(max_depth=6, aa=((0.5, 0.5),))
    singles, rays = [], 0
    for cam in cams:
        f, st = dev.render(cam, p)
        singles.append(f)
        rays += st.trace_calls + st.shadow_rays
    assert not np.array_equal(singles[0], singles[4])
    stride = W * H * 3 + 256
    buf = torch.zeros((5 * stride,), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    st = rt_stats()
    dev.render_batch_device(cams, p, buf.data_ptr(), stride, stats=st)
    got = buf.cpu().numpy()
    for k in range(5):
        assert np.array_equal(got[k * stride:k * stride + W * H * 3].reshape(H, W, 3), singles[k]), k
    assert st.trace_calls + st.shadow_rays == rays
    # asynchronous form, a rectangle of every frame
    buf.zero_()
    torch.cuda.synchronize()       # torch's fill is not ordered against the scene's own stream
    rect = (40, 16, 200, 120)
    dev.render_batch_device(cams, p, buf.data_ptr(), stride, rect=rect)
    dev.status()
    got = buf.cpu().numpy()
    x0, y0, w, h = rect
    for k in range(5):
        fr = got[k * stride:k * stride + W * H * 3].reshape(H, W, 3)
        assert np.array_equal(fr[y0:y0 + h, x0:x0 + w], singles[k][y0:y0 + h, x0:x0 + w]), k
        fr[y0:y0 + h, x0:x0 + w] = 0
        assert not fr.any()
    # tiles of 3 parts, 5 frames each, untiled per frame
    tw = th = 32
    parts = []
    for part in range(3):
        n = dev.lib.rt_tiles_owned(W, H, tw, th, part, 3)
        tstride = n * tw * th * 3
        t = torch.zeros((5 * tstride,), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        dev.render_tiles_batch_device(cams, p, (tw, th), part, 3, t.data_ptr(), tstride)
        dev.status()
        parts.append(t.cpu().numpy().reshape(5, tstride))
    for k in range(5):
        assert np.array_equal(tiling.untile([pp[k] for pp in parts], W, H, tw, th), singles[k]), k
    # ... and untiled on the device in one launch, from the layout a gather leaves: [part][frame][tile slots]
    n_slot = dev.lib.rt_tiles_owned(W, H, tw, th, 0, 3)
    part_bytes = n_slot * tw * th * 3
    allt = torch.zeros((3, 5, part_bytes), dtype=torch.uint8, device="cuda")
    for part in range(3):
        dev.render_tiles_batch_device(cams, p, (tw, th), part, 3, allt[part].data_ptr(), part_bytes)
    dev.status()
    frames = torch.zeros((5, H, W, 3), dtype=torch.uint8, device="cuda")
    cabi.check(dev.lib, dev.lib.rt_untile_batch_device(0, W, H, tw, th, 3, 5 * n_slot, 5, part_bytes, W * H * 3,
                                                       C.c_void_p(allt.data_ptr()), C.c_void_p(frames.data_ptr()), None))
    torch.cuda.synchronize()
    got = frames.cpu().numpy()
    for k in range(5):
        assert np.array_equal(got[k], singles[k]), k


def test_host_async_batches_equal_device_frames():
    """rt_render_batch_host_async: frames delivered to host memory (page-locked via rt_pinned_alloc, and pageable) by
    the scene's copy stream are, byte for byte, the frames of rt_render; several calls in a row reuse the two device
    frame buffers in turn; a rectangle writes only its bytes."""
    from raytracer_amd.scene import make_camera
    fs = wl.scene_bunny(10, third="glass2")
    dev = DeviceScene(fs)
    W, H = 328, 200
    p = Params(max_depth=6)
    cams = [make_camera(W, H, eye=(0.3 * k, 0.1 * k, 0.2 * k), at=(0.05 * k, 0, -1)) for k in range(6)]
    singles = [dev.render(c, p)[0] for c in cams]
    fb = W * H * 3
    ptr = C.c_void_p()
    cabi.check(dev.lib, dev.lib.rt_pinned_alloc(6 * fb, C.byref(ptr)))
    try:
        host = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(6, H, W, 3))
        host[:] = 0
        for first in (0, 2, 4):          # three calls back to back, two frames each, no wait in between
            dev.render_batch_host_async(cams[first:first + 2], p, ptr.value + first * fb, fb)
        dev.status()
        for k in range(6):
            assert np.array_equal(host[k], singles[k]), k
        host[:] = 0
        rect = (40, 16, 200, 120)
        dev.render_batch_host_async(cams[:3], p, ptr.value, fb, rect=rect)
        dev.status()
        x0, y0, w, h = rect
        for k in range(3):
            assert np.array_equal(host[k][y0:y0 + h, x0:x0 + w], singles[k][y0:y0 + h, x0:x0 + w]), k
            host[k][y0:y0 + h, x0:x0 + w] = 0
        assert not host.any()
    finally:
        dev.status()
        dev.lib.rt_pinned_free(ptr)
    pageable = np.zeros((2, H, W, 3), np.uint8)
    dev.render_batch_host_async(cams[1:3], p, pageable.ctypes.data, fb)
    dev.status()
    assert np.array_equal(pageable[0], singles[1]) and np.array_equal(pageable[1], singles[2])


@pytest.mark.gpu
def test_host_primitive_intersect_golden():
    """Primitive::intersect / getNormal of the C++ host surface (reference src/primitives.h:23-24), which run ONE ray
    against ONE primitive on the device (raytracer.hpp: a one-object scene + rt_intersect), against the reference's own
    answers for the sphere, the plane and the lone triangle of tests/golden/intersect.npz: same bool, and on a hit the
    same bits of dist and normal; getNormal against the reference's three-line bodies."""
    import ctypes as C
    from helpers import gold, same_bits
    from raytracer_amd import cabi
    from raytracer_amd.cabi import fptr
    lib = cabi.load()
    g = gold("intersect.npz")
    rays = g["rays"]

    def call(kind, prm, ray, point=None):
        hit, dist = C.c_int(-7), C.c_float(0)
        nrm, gn = np.zeros(3, np.float32), np.zeros(3, np.float32)
        pt = np.ascontiguousarray(point if point is not None else np.zeros(3), np.float32)
        cabi.check(lib, lib.rt_host_primitive_intersect(kind, fptr(np.ascontiguousarray(prm, np.float32)), fptr(np.ascontiguousarray(ray)),
                                                        fptr(pt), C.byref(hit), C.byref(dist), fptr(nrm), fptr(gn)))
        return hit.value, np.float32(dist.value), nrm, gn

    def pick(hits, n=24):
        yes, no = np.flatnonzero(hits)[:n], np.flatnonzero(~hits)[:n]
        assert len(yes) and len(no)
        return list(yes) + list(no)

    sphere, plane = g["scene_spheres"][0], g["scene_planes"][0]            # objects 0 and 4 of the golden scene
    for kind, prm, obj in ((0, sphere, 0), (1, plane, 4)):
        for i in pick(g[f"hit_{obj}"] >= 0):
            hit, d, n, _ = call(kind, prm, rays[i])
            assert hit == int(g[f"hit_{obj}"][i] >= 0), (kind, i)
            if hit:
                assert same_bits(d, g[f"dist_{obj}"][i]) and same_bits(n, g[f"normal_{obj}"][i]), (kind, i)
    tri = np.concatenate([g["lone_v0"][0], g["lone_v1"][0], g["lone_v2"][0]])
    e1, e2 = g["lone_v1"][0] - g["lone_v0"][0], g["lone_v2"][0] - g["lone_v0"][0]
    flat = np.cross(e1 / np.linalg.norm(e1), e2 / np.linalg.norm(e2))
    flat = (flat / np.linalg.norm(flat)).astype(np.float32)
    for i in list(np.flatnonzero(g["lone_hit"] == 0)[:24]) + list(np.flatnonzero(g["lone_hit"] < 0)[:24]):
        hit, d, n, _ = call(2, tri, rays[i])
        assert hit == int(g["lone_hit"][i] == 0), i
        if hit:       # (the golden triangle carries per-vertex normals, this one the flat normal of the 3-vertex constructor)
            assert same_bits(d, g["lone_dist"][i]) and np.allclose(n, flat, atol=2e-6), i
    # getNormal: Sphere (P - c) * float(1.0 / r) (src/primitives.cpp:198-203, container.h:72-75), Plane (a, b, c), Triangle faceNormal
    P = np.array([0.3, -0.2, -7.1], np.float32)
    _, _, _, gn = call(0, sphere, rays[0], P)
    assert same_bits(gn, (P - sphere[:3]) * np.float32(1.0 / np.float64(sphere[3])))
    _, _, _, gn = call(1, plane, rays[0], P)
    assert same_bits(gn, plane[:3])
    _, _, _, gn = call(2, tri, rays[0], P)
    assert np.allclose(gn, flat, atol=2e-6)                                  # the flat face normal the 3-vertex constructor computes
    assert lib.rt_host_primitive_intersect(9, fptr(sphere), fptr(rays[0]), None, C.byref(C.c_int()), C.byref(C.c_float()), fptr(np.zeros(3, np.float32)), None) != 0
