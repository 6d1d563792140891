"""Per-frame time of one rank's share (1/n of the tiles) when B frames are rendered per call, two calls in flight."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene
name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p"
fs, cam, p = wl.make_workload(name)
devs = [DeviceScene(fs), DeviceScene(fs)]
W, H = cam.imWidth, cam.imHeight
base = None
for n in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2,4,8".split(","))]:
    k = devs[0].lib.rt_tiles_owned(W, H, 32, 32, 0, n)
    stride = k * 32 * 32 * 3
    for B in (1, 2, 4, 8, 16, 32):
        bufs = [torch.zeros((B * stride,), dtype=torch.uint8, device="cuda") for _ in devs]
        cams = [cam] * B
        def run(calls):
            for i in range(calls):
                devs[i % 2].render_tiles_batch_device(cams, p, (32, 32), 0, n, bufs[i % 2].data_ptr(), stride)
            torch.cuda.synchronize()
        run(4)
        calls = max(4, 64 // B)
        t = time.perf_counter(); run(calls); ms = (time.perf_counter() - t) / (calls * B) * 1e3
        for d in devs: d.status()
        if base is None: base = ms
        print(f"parts {n} batch {B:2d}: {ms:.4f} ms/frame per rank", flush=True)
