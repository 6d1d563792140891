"""Diagnostic: per-workgroup timeline of one k_extend level (COUNT build): when waves start/end, time in traversal."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
path = f"/tmp/wavelog_{os.getpid()}.bin"
os.environ["RT_WAVE_LOG"] = f"{level}:{path}"
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene
fs, cam, p = wl.make_workload(sys.argv[2] if len(sys.argv) > 2 else "bunny_1080p")
dev = DeviceScene(fs)
dev.render(cam, p)
dev.set_count_work(True)
dev.render(cam, p)
a = np.fromfile(path, dtype=np.uint32).reshape(-1, 4)
a = a[a[:, 1] != 0]
t0 = a[:, 0].min()
start = (a[:, 0] - t0).astype(np.int64) * 0.01
end = (a[:, 1] - t0).astype(np.int64) * 0.01
trav = a[:, 2] * 0.01
cyc = a[:, 3].astype(np.int64)
batches = (a[:, 2] > 0).astype(np.int64)
steps = np.ones(len(a), np.int64)
life = end - start
print(f"level {level}: {len(a)} waves, kernel span {end.max():.1f} us; last start {start.max():.1f} us")
print("wave life us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (life.mean(), *np.percentile(life, [50, 90, 99]), life.max()))
h = batches > 0
print(f"waves with traversal: {h.sum()}; their life: mean {life[h].mean():.1f} p50 {np.percentile(life[h],50):.1f} p90 {np.percentile(life[h],90):.1f} p99 {np.percentile(life[h],99):.1f} max {life[h].max():.1f}")
print(f"trav us/wave: mean {trav[h].mean():.1f} p90 {np.percentile(trav[h],90):.1f} max {trav[h].max():.1f}; node steps/wave mean {steps[h].mean():.1f} p90 {np.percentile(steps[h],90):.0f} max {steps[h].max()}")
print(f"us per node step (trav/steps) mean {(trav[h]/np.maximum(steps[h],1)).mean():.3f}")
# concurrency over time
edges = np.linspace(0, end.max(), 26)
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (lo + hi)
    live = ((start <= mid) & (end > mid))
    print(f"t={mid:7.1f} us: resident waves {live.sum():5d} (with traversal {(live & h).sum():5d}), started so far {(start <= mid).sum():6d}")
worst = np.argsort(-life)[:5]
for w in worst:
    print(f"  worst wave: start {start[w]:.1f} life {life[w]:.1f} trav {trav[w]:.1f} core cycles {cyc[w]} -> {cyc[w]/life[w]:.0f} MHz")
sel = life > 5
print("core clock over waves living > 5 us: mean %.0f MHz, min %.0f, max %.0f" % ((cyc[sel]/life[sel]).mean(), (cyc[sel]/life[sel]).min(), (cyc[sel]/life[sel]).max()))
os.unlink(path)
