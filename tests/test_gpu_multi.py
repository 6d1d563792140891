"""rt_multi_* (one frame on several GPUs of one process: tiles round-robin, peer-copy gather on the first device,
untile) and rt_render_progress, through the C-ABI. The GPU box has one GPU: the device list names it several times,
which takes every step of the path - replicas, per-part tile buffers and streams, cross-stream events, the gather
copies, the untile - except that the copy is device-local instead of a peer copy."""
import numpy as np
import pytest
import torch

from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene, MultiScene, Params, reference_camera

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_multi_frame_equals_single_device_frame(parts):
    fs = wl.scene_bunny(10, third="glass2")
    cam, p = reference_camera(400, 300), Params(max_depth=6, aa=wl.aa_grid(2))
    want, st1 = DeviceScene(fs).render(cam, p)
    multi = MultiScene(fs, [0] * parts)
    assert multi.parts() == parts
    got, stn = multi.render(cam, p)
    assert np.array_equal(got, want)
    assert (stn.trace_calls, stn.shadow_rays) == (st1.trace_calls, st1.shadow_rays)
    assert list(stn.level_rays) == list(st1.level_rays)
    # asynchronous, twice in a row into a device buffer (the second frame reuses the parts' tile buffers), other camera
    cam2 = wl.orbit_cameras(400, 300, 3)[2]
    want2, _ = DeviceScene(fs).render(cam2, p)
    buf = torch.zeros((300, 400, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    multi.render_device(cam, p, buf.data_ptr())
    multi.render_device(cam2, p, buf.data_ptr())
    multi.status()
    assert np.array_equal(buf.cpu().numpy(), want2)
    got3, _ = multi.render(cam, p, want_stats=False)
    assert np.array_equal(got3, want)


def test_multi_rejects_bad_arguments():
    from raytracer_amd import cabi
    fs = wl.scene_bunny(6)
    with pytest.raises(cabi.RtError):
        MultiScene(fs, [])
    with pytest.raises(cabi.RtError):
        MultiScene(fs, [0, 99])           # no such device: the error names it


def test_render_progress_reports_every_pass():
    fs = wl.scene_bunny(8)
    cam, p = reference_camera(320, 240), Params(max_depth=3)
    dev = DeviceScene(fs)
    want, _ = dev.render(cam, p)
    calls = []
    got, _ = dev.render_progress(cam, p, lambda done, total: calls.append((done, total)))
    assert np.array_equal(got, want)
    assert calls == [(320 * 240, 320 * 240)]                       # one pass: one report, at the end
    dev.set_pass_samples(1 << 12)                                  # 4096-sample passes: 19 of them
    calls = []
    got, st = dev.render_progress(cam, p, lambda done, total: calls.append((done, total)))
    assert np.array_equal(got, want) and st.passes == 19
    done = [d for d, _ in calls]
    assert len(calls) == 19 and done == sorted(done) and done[0] == 4096 and calls[-1] == (320 * 240, 320 * 240)
    assert all(t == 320 * 240 for _, t in calls)
    # a rectangle that is not a multiple of 8: reports are in pixels of the rectangle
    calls = []
    dev.render_progress(cam, p, lambda done, total: calls.append((done, total)), rect=(3, 5, 101, 67))
    assert calls[-1] == (101 * 67, 101 * 67) and all(d <= 101 * 67 for d, _ in calls)
