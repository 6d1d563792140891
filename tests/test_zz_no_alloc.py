"""rt_render_device_noalloc (the entry point that allocates nothing: node pool in a file-scope __device__ array)
against rt_render_device, through the C-ABI: same bytes, on two calls in a row, the second with a smaller input.
Collected last (test_zz_*): it must pass alone in a fresh process and after the other tests."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import cabi, workloads as wl
from raytracer_amd.scene import DeviceScene, Params, reference_camera

pytestmark = pytest.mark.gpu


def _frame(dev, cam, p, rect, noalloc):
    """One asynchronous frame into a device buffer, by the documented protocol: rt_scene_status reports a frame whose
    ray trees outgrew the pool (the scene then uses smaller passes) and the caller renders it again."""
    import torch
    buf = torch.zeros((cam.imHeight, cam.imWidth, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()                 # the scene's stream is not ordered against torch's fill
    for attempt in range(10):
        if noalloc:
            dev.render_device_noalloc(cam, p, buf.data_ptr(), rect=rect)
        else:
            dev.render_device(cam, p, buf.data_ptr(), rect=rect)
        try:
            dev.status()                     # waits
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
