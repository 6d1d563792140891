"""One-frame-at-a-time latency of a workload as a function of max_depth: what does each bounce level cost?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene, Params
name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p"
fs, cam, p = wl.make_workload(name)
dev = DeviceScene(fs)
frame = torch.zeros((cam.imHeight, cam.imWidth, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for d in range(0, p.c.max_depth + 1):
    pd = Params(background=tuple(p.c.background), max_depth=d, aa=p.aa)
    ts = []
    for k in range(14):
        t = time.perf_counter()
        dev.render_device(cam, pd, frame.data_ptr())
        dev.status()
        ts.append((time.perf_counter() - t) * 1e3)
    ts = sorted(ts[3:])
    print(f"max_depth {d:2d}: median {ts[len(ts) // 2]:.4f} ms  min {ts[0]:.4f}")
dev.close()
