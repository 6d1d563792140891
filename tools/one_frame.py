"""Renders a few frames of a workload (async path, no stats) - a small target for rocprofv3 counter passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene
name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
fs, cam, p = wl.make_workload(name)
dev = DeviceScene(fs)
frame = torch.zeros((cam.imHeight, cam.imWidth, 3), dtype=torch.uint8, device="cuda")
for _ in range(n):
    dev.render_device(cam, p, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
