"""Diagnostic: what one rank of an N-GPU run does per frame (its tiles only, two frames in flight), timed on one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene
fs, cam, p = wl.make_workload(sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p")
devs = [DeviceScene(fs), DeviceScene(fs)]
W, H = cam.imWidth, cam.imHeight
base = None
for n in (1, 2, 4, 8):
    k = devs[0].lib.rt_tiles_owned(W, H, 32, 32, 0, n)
    bufs = [torch.zeros((k * 32 * 32 * 3,), dtype=torch.uint8, device="cuda") for _ in devs]
    def run(steps):
        for i in range(steps):
            devs[i % 2].render_tiles_device(cam, p, (32, 32), 0, n, bufs[i % 2].data_ptr())
        torch.cuda.synchronize()
    run(6)
    t = time.perf_counter(); run(60); ms = (time.perf_counter() - t) / 60 * 1e3
    base = base or ms
    print(f"parts {n}: {ms:.3f} ms/frame per rank -> speedup {base / ms:.2f}x, efficiency {base / ms / n:.2f} (before the gather)")

print("-- host time to issue the launches vs total")
for n in (1, 8):
    k = devs[0].lib.rt_tiles_owned(W, H, 32, 32, 0, n)
    bufs = [torch.zeros((k * 32 * 32 * 3,), dtype=torch.uint8, device="cuda") for _ in devs]
    for i in range(6):
        devs[i % 2].render_tiles_device(cam, p, (32, 32), 0, n, bufs[i % 2].data_ptr())
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(60):
        devs[i % 2].render_tiles_device(cam, p, (32, 32), 0, n, bufs[i % 2].data_ptr())
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(f"parts {n}: issue {t_issue / 60 * 1e3:.3f} ms/frame, total {t_all / 60 * 1e3:.3f} ms/frame")
    for fl in (3, 4):
        ds = devs + [DeviceScene(fs) for _ in range(fl - 2)]
        bs = [torch.zeros((k * 32 * 32 * 3,), dtype=torch.uint8, device="cuda") for _ in ds]
        for i in range(2 * fl):
            ds[i % fl].render_tiles_device(cam, p, (32, 32), 0, n, bs[i % fl].data_ptr())
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(60):
            ds[i % fl].render_tiles_device(cam, p, (32, 32), 0, n, bs[i % fl].data_ptr())
        torch.cuda.synchronize()
        print(f"   {fl} frames in flight: {(time.perf_counter() - t) / 60 * 1e3:.3f} ms/frame")
