"""Diagnostic: traversal work counters per recursion level and kernel (differences of frames at max_depth d and d-1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene, Params
name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p"
fs, cam, p = wl.make_workload(name)
dev = DeviceScene(fs)
dev.set_count_work(True)
F = ["nodes_visited", "tris_tested", "wave_node_steps", "wave_tri_steps", "trav_ticks_sum", "batches", "trav_waves", "wave_ticks_sum"]
prev = {k: np.zeros(len(F), np.int64) for k in range(5)}
prays = np.zeros(2, np.int64)
for d in range(0, 3):
    pp = Params(max_depth=d, aa=p.aa)
    _, st = dev.render(cam, pp)
    rays = np.array([st.trace_calls, st.shadow_rays], np.int64)
    for k, kn in ((0, "extend wide"), (2, "extend quad"), (1, "shade wide "), (3, "shade quad "), (4, "shade park ")):
        w = dev.get_work(k)
        cur = np.array([getattr(w, f) for f in F], np.int64)
        dl = cur - prev[k]; prev[k] = cur
        r = rays[k & 1] - prays[k & 1]
        print(f"L{d} {kn}: rays {r} | quad nodes {dl[0]} ({dl[0]/max(r,1):.2f}/ray) tris {dl[1]} | wave steps node {dl[2]} tri {dl[3]}"
              f" | batches {dl[5]} in {dl[6]} waves, {dl[2]/max(dl[5],1):.1f} node + {dl[3]/max(dl[5],1):.1f} tri steps/batch,"
              f" {dl[4]*0.01/max(dl[5],1):.2f} us/batch, trav max/wave {w.trav_ticks_max*0.01:.1f} us (cumulative)"
              f" | wave life sum {dl[7]*0.01/1e3:.1f} ms, max {w.wave_ticks_max*0.01:.1f} us (cumulative)")
    prays = rays
