"""Where does the time of the host-delivery pipeline go? Render-only, copy-only, and both, with 1..3 scenes in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene
fs, cam, p = wl.make_workload("bunny_1080p")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = cam.imWidth, cam.imHeight
fb = W * H * 3
cams = wl.orbit_cameras(W, H, B)
N = 3
devs = [DeviceScene(fs) for _ in range(N)]
dframes = [torch.zeros((B, H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(N)]
pins = [torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(N)]
streams = [torch.cuda.ExternalStream(d.stream()) for d in devs]
torch.cuda.synchronize()
def t(label, fn, calls=24, n=2):
    for i in range(2 * n): fn(i % n)
    for d in devs: d.status()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(calls): fn(i % n)
    for d in devs: d.status()
    torch.cuda.synchronize()
    print(f"{label:50s} {(time.perf_counter() - t0) / (calls * B) * 1e3:.4f} ms/frame")
for n in (1, 2, 3):
    t(f"render to device, {n} in flight", lambda k: devs[k].render_batch_device(cams, p, dframes[k].data_ptr(), fb), n=n)
def copy_only(k):
    with torch.cuda.stream(streams[k]):
        pins[k].copy_(dframes[k], non_blocking=True)
for n in (1, 2):
    t(f"copy only (torch, scene stream), {n} streams", copy_only, n=n)
def render_copy_same(k):
    devs[k].render_batch_device(cams, p, dframes[k].data_ptr(), fb)
    with torch.cuda.stream(streams[k]):
        pins[k].copy_(dframes[k], non_blocking=True)
for n in (1, 2, 3):
    t(f"render + copy on the scene stream, {n} in flight", render_copy_same, n=n)
for n in (1, 2, 3):
    t(f"rt_render_batch_host_async, {n} in flight", lambda k: devs[k].render_batch_host_async(cams, p, pins[k].data_ptr(), fb), n=n)
for d in devs:
    d.close()
