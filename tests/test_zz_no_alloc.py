"""rt_render_device_noalloc (the entry point that allocates nothing: node pool in a file-scope __device__ array)
against rt_render_device, through the C-ABI: same bytes, on two calls in a row, the second with a smaller input.
Collected last (test_zz_*): it must pass alone in a fresh process and after the other tests."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import cabi, workloads as wl
from raytracer_amd.scene import DeviceScene, Params, reference_camera

pytestmark = pytest.mark.gpu


def _frame(dev, cam, p, rect, noalloc, attempts=None):
    """One asynchronous frame into a device buffer, by the documented protocol: rt_scene_status reports a frame whose
    ray trees outgrew the pool (the scene then uses smaller passes) and the caller renders it again. The no-alloc call
    must leave the device's free memory as it found it (hipMemGetInfo, through torch)."""
    import torch
    buf = torch.zeros((cam.imHeight, cam.imWidth, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()                 # the scene's stream is not ordered against torch's fill
    for attempt in range(10):
        if noalloc:
            free0 = torch.cuda.mem_get_info()[0]
            dev.render_device_noalloc(cam, p, buf.data_ptr(), rect=rect)
            assert torch.cuda.mem_get_info()[0] == free0, "rt_render_device_noalloc changed the device's free memory"
        else:
            dev.render_device(cam, p, buf.data_ptr(), rect=rect)
        try:
            dev.status()                     # waits
            if noalloc:
                assert torch.cuda.mem_get_info()[0] == free0
            if attempts is not None:
                attempts.append(attempt)
            return buf.cpu().numpy()
        except cabi.RtError as e:
            assert "pool" in str(e)
    raise AssertionError("the frame still overflows after 10 attempts")


def test_noalloc_equals_render_device():
    fs = wl.scene_bunny(10, third="glass2")
    dev = DeviceScene(fs)
    p = Params(max_depth=6, aa=((0.5, 0.5),))
    # first call: 400x296 at 2 spp = 236,800 samples = 15 passes of the static pool
    cam = reference_camera(400, 296)
    want = _frame(dev, cam, p, None, False)
    got = _frame(dev, cam, p, None, True)
    assert want.any() and np.array_equal(got, want)
    # second call, smaller input: a rectangle of a smaller frame, one pass; the array still holds the first call's data
    cam2 = reference_camera(160, 120)
    rect = (24, 16, 97, 61)
    want2 = _frame(dev, cam2, p, rect, False)
    got2 = _frame(dev, cam2, p, rect, True)
    assert want2.any() and np.array_equal(got2, want2)
    # another scene on the same GPU shares the array (calls one after the other): still the same bytes
    dev_b = DeviceScene(wl.scene_analytic(third="glass2"))
    p1 = Params()
    want3 = _frame(dev_b, cam2, p1, None, False)
    assert np.array_equal(_frame(dev_b, cam2, p1, None, True), want3)
    assert np.array_equal(_frame(dev, cam, p, None, True), want)


def test_noalloc_rejects_sizes_above_its_maximum():
    dev = DeviceScene(wl.scene_analytic())
    import torch
    cam = reference_camera(1920, 1080)
    buf = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda")
    p4 = Params(aa=((0.5, 0.5), (0.25, 0.75), (0.75, 0.25)))
    rc = dev.lib.rt_render_device_noalloc(dev.h, C.byref(cam), C.byref(p4.c), 0, 0, 1920, 1080, C.c_void_p(buf.data_ptr()), None)
    assert rc == -5 and b"maximum" in dev.lib.rt_last_error()
    # 1920x1080 at 1 sample per pixel is within the maximum (2,073,600 <= 2^21) and equals the ordinary call
    p1 = Params()
    want = _frame(dev, cam, p1, None, False)
    assert np.array_equal(_frame(dev, cam, p1, None, True), want)


def test_noalloc_overflows_at_its_maximum_size():
    """The largest frame the entry point takes (1920x1080 at 1 sample per pixel) on a scene whose ray trees outgrow the
    static pool twice over: every overflow quarters the NO-ALLOC passes only (2^14 -> 2^12 -> 2^10 samples: 2025 passes,
    more than the scene's 1024 pass-counter slots, which an asynchronous frame takes in turn), the frame then equals the
    ordinary call's, and the scene's ordinary passes are what they were."""
    from test_gpu_parity import hall_of_mirrors
    fs = hall_of_mirrors()
    cam, p = reference_camera(1920, 1080), Params(max_depth=8)
    dev = DeviceScene(fs)
    want = _frame(dev, cam, p, None, False)
    ordinary = dev.pass_samples()
    tries = []
    got = _frame(dev, cam, p, None, True, attempts=tries)
    assert tries[0] >= 2, tries              # reported (and learned from) at least twice
    assert np.array_equal(got, want)
    assert dev.pass_samples() == ordinary
    assert np.array_equal(_frame(dev, cam, p, None, False), want)      # the ordinary path still renders as before
