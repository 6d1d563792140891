"""One frame at a time, split into horizontal bands that are rendered by several scenes (each with its own pool and
streams) at once: does the latency-bound end of one band's recursion hide under the throughput-bound start of the
next? Prints the median wall time of a frame per number of bands, and checks the bytes against the one-call frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from statistics import median
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene

name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p"
fs, cam, p = wl.make_workload(name)
W, H = cam.imWidth, cam.imHeight
max_scenes = 4
devs = [DeviceScene(fs) for _ in range(max_scenes)]
streams = [torch.cuda.Stream() for _ in range(max_scenes)]
want = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
frame = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
devs[0].render_device(cam, p, want.data_ptr(), stream=streams[0].cuda_stream)
devs[0].status()


def bands(n, stagger):
    """n bands of rows, multiples of 8 high; stagger > 1: the first band smaller, so that its tail starts earlier."""
    rows = [((H * (k + 1)) // n) // 8 * 8 for k in range(n)]
    rows[-1] = H
    if stagger != 1.0 and n == 2:
        rows[0] = int(H / (1 + stagger)) // 8 * 8
    y0, out = 0, []
    for r in rows:
        out.append((0, y0, W, r - y0)); y0 = r
    return out


for n, n_scenes, stagger in ((1, 1, 1.0), (2, 2, 1.0), (2, 2, 1.5), (3, 3, 1.0), (4, 4, 1.0), (4, 2, 1.0), (8, 4, 1.0)):
    bs = bands(n, stagger)
    ts = []
    for k in range(14):
        frame.zero_(); torch.cuda.synchronize()
        t = time.perf_counter()
        for j, rect in enumerate(bs):
            q = j % n_scenes
            if j >= n_scenes:
                devs[q].status()
            devs[q].render_device(cam, p, frame.data_ptr(), rect=rect, stream=streams[q].cuda_stream)
        for q in range(min(n, n_scenes)):
            devs[q].status()
        ts.append((time.perf_counter() - t) * 1e3)
    ok = bool(torch.equal(frame, want))
    print(f"{name} bands {n} scenes {n_scenes} stagger {stagger}: median {median(ts[4:]):.4f} ms min {min(ts[4:]):.4f} identical {ok}", flush=True)
