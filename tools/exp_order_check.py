"""Stress check of the expensive-groups-first order: the same batch of frames rendered over and over on one scene (every
call but the first few uses the previous call's order), each compared with the frames of a scene that never orders."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene
name = sys.argv[1] if len(sys.argv) > 1 else "c1_scene_yaml_800x600"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
fs, cam, p = wl.make_workload(name)
W, H = cam.imWidth, cam.imHeight
cams = wl.orbit_cameras(W, H, B)
ref_dev = DeviceScene(fs); ref_dev.set_option("order0", 0)
dev = DeviceScene(fs)
ref = torch.zeros((B, H, W, 3), dtype=torch.uint8, device="cuda")
out = torch.zeros((B, H, W, 3), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    ref_dev.render_batch_device(cams, p, ref.data_ptr(), H * W * 3, stream=st)
torch.cuda.synchronize()
bad = late = 0
mode = os.environ.get("ORDER_CHECK_STREAM", "null")
own = torch.cuda.Stream()
for k in range(n):
    stream = own if mode == "own" else torch.cuda.current_stream()
    with torch.cuda.stream(stream):
        out.zero_()
        dev.render_batch_device(cams, p, out.data_ptr(), H * W * 3, stream=stream.cuda_stream)
        if k % 3 == 0:
            torch.cuda.synchronize()       # some calls find the previous order finished, some do not
        stream.synchronize()
        d = (out != ref).any(dim=3)
        wrong = bool(d.any())
    if wrong:
        torch.cuda.synchronize()
        d2 = (out != ref).any(dim=3)
        if bool(d2.any()):
            bad += 1
            print(f"call {k}: {int(d2.sum())} pixels differ after a device sync (first look: {int(d.sum())})")
        else:
            late += 1
            print(f"call {k}: {int(d.sum())} pixels were not there after the stream sync, all there after a device sync")
print(f"{name} B={B} stream {mode}: {bad} of {n} calls wrong, {late} late")
