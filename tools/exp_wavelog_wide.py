"""Per-workgroup timeline of a level's per-lane kernel (COUNT build): when waves start and end, how many packet steps
they run. usage: exp_wavelog_wide.py <level> <workload> [shade]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
path = f"/tmp/wavelog_{os.getpid()}.bin"
os.environ["RT_WAVE_LOG"] = f"{level}:{path}"
if len(sys.argv) > 3 and sys.argv[3] == "shade":
    os.environ["RT_WAVE_LOG_SHADE"] = "1"
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene, Params
fs, cam, p = wl.make_workload(sys.argv[2] if len(sys.argv) > 2 else "bunny_1080p")
p = Params(background=tuple(p.c.background), max_depth=0, aa=p.aa)
dev = DeviceScene(fs)
dev.render(cam, p)
dev.set_count_work(True)
dev.render(cam, p)
counted = np.fromfile(path, dtype=np.uint32).reshape(-1, 4)
dev.set_count_work(False)
dev.render(cam, p)
dev.render(cam, p)
a = np.fromfile(path, dtype=np.uint32).reshape(-1, 4)      # production kernel: start / end only
a[:, 2] = counted[:, 2]                                     # steps from the counting build
idx = np.nonzero(a[:, 1] != 0)[0]
a = a[idx]
t0 = a[:, 0].min()
start = (a[:, 0] - t0).astype(np.int64) * 0.01
end = (a[:, 1] - t0).astype(np.int64) * 0.01
steps = a[:, 2].astype(np.int64)
life = end - start
print(f"{len(a)} waves, kernel span {end.max():.1f} us; last start {start.max():.1f} us; total steps {steps.sum()}")
print("wave life us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (life.mean(), *np.percentile(life, [50, 90, 99]), life.max()))
h = steps > 0
print(f"waves with traversal: {h.sum()}; life mean {life[h].mean():.1f} p50 {np.percentile(life[h], 50):.1f} p90 {np.percentile(life[h], 90):.1f} max {life[h].max():.1f}; "
      f"steps mean {steps[h].mean():.1f} p90 {np.percentile(steps[h], 90):.0f} max {steps[h].max()}")
print(f"waves without: life mean {life[~h].mean():.1f} p90 {np.percentile(life[~h], 90):.1f}")
A = np.vstack([steps[h], np.ones(h.sum())]).T
coef = np.linalg.lstsq(A, life[h], rcond=None)[0]
print(f"life ~ {coef[0]:.3f} us/step * steps + {coef[1]:.1f} us")
edges = np.linspace(0, end.max(), 21)
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (lo + hi)
    live = ((start <= mid) & (end > mid))
    print(f"t={mid:7.1f} us: resident waves {live.sum():5d} (traversing {(live & h).sum():5d}), started {(start <= mid).sum():6d}")
for w in np.argsort(-life)[:6]:
    print(f"  longest: wg {idx[w]} start {start[w]:.1f} life {life[w]:.1f} steps {steps[w]}")
for w in np.argsort(-end)[:6]:
    print(f"  last to end: wg {idx[w]} start {start[w]:.1f} life {life[w]:.1f} steps {steps[w]}")
