"""Per-phase times of k_tail (RT_TAIL_LOG): closest-hit phases, the wait for the light loops, fold phases; in us."""
import os, sys, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_TAIL_LOG"] = "/tmp/tail.log"
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene
fs, cam, p = wl.make_workload(sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p")
dev = DeviceScene(fs)
frame = torch.zeros((cam.imHeight, cam.imWidth, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(4):
    dev.render_device(cam, p, frame.data_ptr())
    dev.status()
raw = open("/tmp/tail.log", "rb").read()
for wg, base in (("E workgroup 0", 0), ("last E workgroup", 64)):
    print(wg)
    rows = [struct.unpack_from("4I", raw, 16 * (base + k)) for k in range(64)]
    rows = [r for r in rows if r[2]]
    t00 = min(r[0] for r in rows)
    for r in sorted(rows, key=lambda r: r[0]):
        tag = "wait for light loops" if r[3] == 99 else (f"fold level {r[3] - 100}" if r[3] >= 100 else f"closest hits level {r[3]}")
        print(f"  {tag:24s} start {(r[0] - t00) / 100:7.1f}  work {((r[1] - r[0]) & 0xFFFFFFFF) / 100:6.1f}  barrier {((r[2] - r[1]) & 0xFFFFFFFF) / 100:6.1f} us")
