"""One rank's share (part 0 of n) of B tiled frames per call, calls one after the other (each waited for): a target for
rocprofv3 --kernel-trace. one_rank.py <parts> <B> <calls>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene
n, B, calls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fs, cam, p = wl.make_workload("bunny_1080p")
dev = DeviceScene(fs)
W, H = cam.imWidth, cam.imHeight
k = dev.lib.rt_tiles_owned(W, H, 32, 32, 0, n)
stride = k * 32 * 32 * 3
buf = torch.zeros((B * stride,), dtype=torch.uint8, device="cuda")
cams = wl.orbit_cameras(W, H, B)
for _ in range(calls):
    dev.render_tiles_batch_device(cams, p, (32, 32), 0, n, buf.data_ptr(), stride)
    dev.status()
torch.cuda.synchronize()
