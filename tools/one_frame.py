"""Renders a few frames of a workload - a small target for rocprofv3 passes. Default: the asynchronous multi-stream path;
third argument "serial": calls with statistics, i.e. every launch alone on one stream between HIP events (what bench.py's
roofline durations are measured on), so that rocprofv3's per-kernel averages can be held against them."""
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
serial = len(sys.argv) > 3 and sys.argv[3] == "serial"
if serial:
    from raytracer_amd.cabi import rt_stats
    ms, cnt = [0.0] * 5, [0] * 5
    for k in range(n + 2):
        st = rt_stats()
        dev.render_device(cam, p, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, stats=st)
        if k >= 2:       # the first frames of a scene do not know their level sizes yet
            for j in range(5):
                ms[j] += st.kernel_ms_by[j]; cnt[j] += st.kernel_launches_by[j]
    names = ("k_extend_wide", "k_shade_wide", "k_extend_quad", "k_shade_quad<0|1>", "k_shade_quad<2>")
    for j in range(5):
        if cnt[j]:
            print(f"HIP events: {names[j]}: {cnt[j]} launches, average {ms[j] / cnt[j] * 1e3:.1f} us")
else:
    B = int(os.environ.get("ONE_FRAME_BATCH", "1"))      # frames per call (the counters of the last CALL are what the PMC scripts sum)
    if B > 1:
        frames = torch.zeros((B, cam.imHeight, cam.imWidth, 3), dtype=torch.uint8, device="cuda")
        for _ in range(n):
            dev.render_batch_device([cam] * B, p, frames.data_ptr(), cam.imHeight * cam.imWidth * 3, stream=torch.cuda.current_stream().cuda_stream)
    else:
        for _ in range(n):
            dev.render_device(cam, p, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            if os.environ.get("ONE_FRAME_SYNC"):      # one frame at a time: wait for it (latency timelines)
                dev.status()
torch.cuda.synchronize()
