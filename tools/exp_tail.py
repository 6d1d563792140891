"""Diagnostic: per-segment wave cycles of the deep (sparse) levels = counters at max_depth 10 minus at max_depth 4."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene, Params
fs, cam, p = wl.make_workload("bunny_1080p")
dev = DeviceScene(fs)
dev.set_count_work(True)
res = {}
for d in (4, 10):
    pp = Params(max_depth=d)
    dev.render(cam, pp)
    _, st = dev.render(cam, pp)
    w = dev.get_work()
    res[d] = (np.array(list(w.segment_cycles), dtype=np.int64), w.nodes_visited, w.tris_tested, w.wave_node_steps, list(st.level_rays[:11]))
a, b = res[4], res[10]
print("levels", b[4])
print("seg cycles levels 5..10 [fetch,closest,shadow,shade,append,bvh]:", (b[0] - a[0]).tolist())
print("nodes", b[1] - a[1], "tris", b[2] - a[2], "wave node steps", b[3] - a[3])
n = sum(b[4][5:])
print("rays", n, "per-ray cycles", ((b[0] - a[0]) / max(n, 1)).round().tolist())
