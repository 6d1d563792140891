"""Life times of the waves of the level-0 kernels (COUNT instantiations: rt_set_count_work), per kernel kind."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import workloads as wl
from raytracer_amd.cabi import rt_stats
from raytracer_amd.scene import DeviceScene, Params
name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p"
fs, cam, p = wl.make_workload(name)
p0 = Params(background=tuple(p.c.background), max_depth=0, aa=p.aa)
dev = DeviceScene(fs)
frame = torch.zeros((cam.imHeight, cam.imWidth, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for prm in (p, p0):
    dev.set_count_work(False)
    st = rt_stats(); dev.render_device(cam, prm, frame.data_ptr(), stats=st)
    dev.set_count_work(True)
    st = rt_stats(); dev.render_device(cam, prm, frame.data_ptr(), stats=st)
    print("depth", prm.c.max_depth, "kernel ms by kind", [round(x, 4) for x in st.kernel_ms_by], list(st.kernel_launches_by))
    n_waves = cam.imWidth * cam.imHeight * p.spp // 64
    for k in range(5):
        w = dev.get_work(k)
        if w.wave_ticks_sum:
            print(f"  kind {k}: wave life sum {w.wave_ticks_sum / 100:.0f} us, max {w.wave_ticks_max / 100:.1f} us, mean over {n_waves} level-0 waves {w.wave_ticks_sum / 100 / n_waves:.2f} us; "
                  f"nodes {w.nodes_visited} tris {w.tris_tested} wave_node_steps {w.wave_node_steps} wave_tri_steps {w.wave_tri_steps}")
