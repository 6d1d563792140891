"""Two scenes alternating calls of B frames (what bench.py's timed region does): a target for rocprofv3 --kernel-trace.
two_in_flight.py <B> <calls>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene
B, calls = int(sys.argv[1]), int(sys.argv[2])
fs, cam, p = wl.make_workload("bunny_1080p")
devs = [DeviceScene(fs), DeviceScene(fs)]
W, H = cam.imWidth, cam.imHeight
cams = wl.orbit_cameras(W, H, B)
bufs = [torch.zeros((B, H, W, 3), dtype=torch.uint8, device="cuda") for _ in devs]
torch.cuda.synchronize()
for i in range(calls):
    devs[i % 2].render_batch_device(cams, p, bufs[i % 2].data_ptr(), H * W * 3, stream=devs[i % 2].stream())
for d in devs:
    d.status()
torch.cuda.synchronize()
