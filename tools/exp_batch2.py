"""Per-frame time of one rank's share (part 0 of n) when B frames are rendered per call, two calls in flight, with and
without expensive groups first (option order0): exp_batch2.py <workload> <parts,...> <B,...>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene
name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p"
fs, cam, p = wl.make_workload(name)
devs = [DeviceScene(fs), DeviceScene(fs)]
W, H = cam.imWidth, cam.imHeight
for n in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8").split(",")]:
    k = devs[0].lib.rt_tiles_owned(W, H, 32, 32, 0, n)
    stride = k * 32 * 32 * 3
    for B in [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "10").split(",")]:
        bufs = [torch.zeros((B * stride,), dtype=torch.uint8, device="cuda") for _ in devs]
        cams = wl.orbit_cameras(W, H, B)
        def run(calls):
            for i in range(calls):
                devs[i % 2].render_tiles_batch_device(cams, p, (32, 32), 0, n, bufs[i % 2].data_ptr(), stride)
            torch.cuda.synchronize()
        for rep in range(2):
            for order in (0, 1):
                for d in devs: d.set_option("order0", order)
                run(6)
                calls = max(8, 128 // B)
                t = time.perf_counter(); run(calls); ms = (time.perf_counter() - t) / (calls * B) * 1e3
                for d in devs: d.status()
                print(f"parts {n} batch {B:2d} order0 {order}: {ms:.4f} ms/frame per rank ({k * 1024 * B // 64} groups per call)", flush=True)
