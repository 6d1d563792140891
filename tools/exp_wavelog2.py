"""Diagnostic (library built with -DRT_EXPERIMENTS): start and end of every wave of one level's per-lane launch in the
RELEASE instantiation of the kernel (no work counters: two clock reads and one store per wave), launches one at a time.
    RT_AMD_LIB=tools/bin/librt_amd_exp.so [RT_WAVE_LOG_SHADE=1] python3 tools/exp_wavelog2.py <level> <workload>"""
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
dev.render(cam, p)
a = np.fromfile(path, dtype=np.uint32).reshape(-1, 4)
a = a[a[:, 1] != 0]
t0 = a[:, 0].min()
start = (a[:, 0] - t0).astype(np.int64) * 0.01
end = (a[:, 1] - t0).astype(np.int64) * 0.01
life = end - start
print(f"level {level}: {len(a)} waves logged, span {end.max():.1f} us; last start {start.max():.1f} us")
print("wave life us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (life.mean(), *np.percentile(life, [50, 90, 99]), life.max()))
edges = np.linspace(0, end.max(), 31)
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (lo + hi)
    live = ((start <= mid) & (end > mid))
    print(f"t={mid:7.1f} us: resident waves {live.sum():5d}, started so far {(start <= mid).sum():6d}, mean life of those resident {life[live].mean() if live.any() else 0:6.1f}")
os.unlink(path)
