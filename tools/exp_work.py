"""Diagnostic: traversal work counters per recursion level (differences of frames rendered at max_depth d and d-1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import workloads as wl
from raytracer_amd.scene import DeviceScene, Params
name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p"
fs, cam, p = wl.make_workload(name)
dev = DeviceScene(fs)
dev.set_count_work(True)
prev = np.zeros(6, np.int64)
for d in range(0, 4):
    pp = Params(max_depth=d, aa=p.aa)
    _, st = dev.render(cam, pp)
    w = dev.get_work()
    cur = np.array([w.nodes_visited, w.tris_tested, w.wave_node_steps, w.wave_tri_steps, st.trace_calls, st.shadow_rays], np.int64)
    dl = cur - prev
    prev = cur
    rays = dl[4] + dl[5]
    print(f"skip={os.environ.get('RT_DEBUG_SKIP','0')} level {d}: trace {dl[4]} shadow {dl[5]} | lane nodes {dl[0]} ({dl[0]/max(rays,1):.1f}/ray) tris {dl[1]} ({dl[1]/max(rays,1):.1f}/ray)"
          f" | wave node steps {dl[2]} tri steps {dl[3]} | lanes/step nodes {dl[0]/max(dl[2],1):.1f} tris {dl[1]/max(dl[3],1):.1f}")
