#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY. Generates tests/golden/*.npz from the REAL reference objects (oracle/_ref/libref.so,
built by oracle/Makefile from /root/reference). Runs only where /root/reference exists; the fixtures it writes are
data (inputs + the reference's outputs), committed, and are what pins the C restatement and the host surface on
machines without the reference.

    python oracle/make_golden.py
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from raytracer_amd import meshgen, workloads as wl  # noqa: E402
from raytracer_amd.cabi import fptr, rt_camera, rt_plane  # noqa: E402
from raytracer_amd.scene import (DEFAULT_AA, FlatScene, Mesh, Params, camera_to_array, material)  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ROT, SCALE, TRANS = (0.0, 0.5235987755982988, 0.0), 20.0, (-1.5, -1.0, -6.0)   # scene.yaml:99-101


def ref_camera(W, H, eye=(0, 0, 0), up=(0, 1, 0), at=(0, 0, -1), fov=60.0, rot=None, move=None):
    import ctypes as C
    lib = O.ref()
    cam = rt_camera()
    f = lambda v: fptr(np.asarray(v, np.float32)) if v is not None else None
    aspect = float(np.float32(W) / np.float32(H))
    lib.ref_camera_make(f(eye), f(up), f(at), aspect, fov, W, H, f(rot), f(move), C.byref(cam))
    return cam


def edge_rays(fs):
    """Rays that exercise the quirks: inside spheres, parallel to planes, zero components, non-unit, NaN,
    through mesh vertices and edge midpoints."""
    r = []
    for s in fs.spheres:
        c = np.array(s.center, np.float32)
        r.append([*c, 0, 0, -1])                         # origin at the centre
        r.append([*(c + [0, 0, s.radius * 0.5]), 0, 1, 0])
        r.append([*(c + [0, 0, 3 * s.radius]), 0, 0, -1])   # head-on
        r.append([*(c + [s.radius, 0, 3 * s.radius]), 0, 0, -1])   # grazing
        r.append([*(c + [0, 0, 3 * s.radius]), 0, 0, -2.5])  # non-unit direction
    r += [[0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, -1], [0, 0, 0, 0, 0, 1], [0, -4, 0, 1, 0, 0],
          [0, 0, 0, float("nan"), 0, -1], [0, 0, 0, 0, 0, 0], [1, 1, 1, 1e-20, 1, -1e-20]]
    if fs.meshes:
        m = fs.meshes[0]
        o = np.array([-0.5, 0, -0.25], np.float32)
        v0, v1 = m.v[0][::7], m.v[1][::7]
        for p in np.concatenate([v0, (v0 + v1) * np.float32(0.5)]):
            d = (p - o).astype(np.float32)
            r.append([*o, *d])                            # unnormalised, through a vertex / an edge midpoint
            dn = d / np.float32(np.linalg.norm(d))
            r.append([*o, *dn])
    return np.array(r, np.float32)


def random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-3.5, 3.5, (n, 3)).astype(np.float32)
    o[:, 2] = rng.uniform(-11, 0, n).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    unit = rng.random(n) < 0.7
    d[unit] /= np.linalg.norm(d[unit], axis=1, keepdims=True).astype(np.float32)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def main():
    import ctypes as C
    os.makedirs(GOLD, exist_ok=True)
    O.build()
    if not O.have_ref():
        raise SystemExit("oracle/_ref/libref.so missing: /root/reference is needed to generate golden vectors")
    lib = O.ref()

    # ---------- host surface ----------
    ka = np.zeros(16, np.float32)
    lib.ref_known_answers(fptr(ka))
    rng = np.random.default_rng(1)
    cam_in, cam_out = [], []
    cases = [dict(W=800, H=600, move=(-0.5, 0, -0.25)), dict(W=1920, H=1080, move=(-0.5, 0, -0.25)),
             dict(W=640, H=360, eye=(1, 2, 3), up=(0.1, 2, 0.2), at=(0.3, -0.2, -1.5), fov=47.5),
             dict(W=333, H=777, up=(0, 3, 0), at=(1, 0, -1), rot=(0.1, -0.2, 0.3), fov=75.0),
             dict(W=3840, H=2160, rot=(0, 0, float(np.pi / 16)), move=(0.25, 0.5, 1.0))]
    for c in cases:
        cam = ref_camera(**c)
        cam_in.append([*c.get("eye", (0, 0, 0)), *c.get("up", (0, 1, 0)), *c.get("at", (0, 0, -1)),
                       c.get("fov", 60.0), c["W"], c["H"], *(c.get("rot") or (np.nan,) * 3),
                       *(c.get("move") or (np.nan,) * 3)])
        cam_out.append(camera_to_array(cam))
    normals = rng.normal(size=(40, 3)).astype(np.float32)
    normals[:5] = [[0, 0, 1], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]]
    dists = rng.uniform(-5, 12, 40).astype(np.float32)
    planes = np.zeros((40, 4), np.float32)
    for i in range(40):
        p = rt_plane()
        lib.ref_plane_make(fptr(np.ascontiguousarray(normals[i])), float(dists[i]), C.byref(p))
        planes[i] = [p.a, p.b, p.c, p.d]
    angles = rng.uniform(-3.2, 3.2, (24, 3)).astype(np.float32)
    angles[0] = ROT
    rots = np.zeros((24, 9), np.float32)
    for i in range(24):
        lib.ref_mat3_rotation(float(angles[i, 0]), float(angles[i, 1]), float(angles[i, 2]), fptr(rots[i]))

    # ---------- PLY loader + transform chain ----------
    V, F = meshgen.blob(3, seed=5)
    tmp = os.path.join(tempfile.gettempdir(), f"golden_{os.getpid()}.ply")
    meshgen.write_ply(tmp, V, F)
    ply_text = np.frombuffer(open(tmp, "rb").read(), dtype=np.uint8)
    raw, raw_bb = O.ref_load_ply(tmp, transform=False)
    xf, xf_bb = O.ref_load_ply(tmp, ROT, SCALE, TRANS)
    os.unlink(tmp)
    bb2 = np.zeros(4, np.float32)
    lib.ref_bbvol(fptr(xf[0]), fptr(xf[1]), fptr(xf[2]), xf[0].shape[0], fptr(bb2))
    np.savez_compressed(os.path.join(GOLD, "host.npz"), known_answers=ka, cam_in=np.array(cam_in, np.float64),
                        cam_out=np.array(cam_out), plane_normals=normals, plane_dists=dists, planes=planes,
                        rot_angles=angles, rot_mats=rots, ply_text=ply_text, ply_rot=np.array(ROT, np.float32),
                        ply_scale=np.float64(SCALE), ply_trans=np.array(TRANS, np.float32),
                        ply_raw=np.stack(raw), ply_raw_bb=raw_bb, ply_xf=np.stack(xf), ply_xf_bb=xf_bb,
                        bbvol_of_xf=bb2)

    # ---------- G1 camera rays ----------
    g1 = {}
    for ci, c in enumerate(cases[:4]):
        cam = ref_camera(**c)
        W, H = c["W"], c["H"]
        xy = np.stack([rng.integers(0, W, 150), rng.integers(0, H, 150)], axis=1)
        xy = np.concatenate([xy, [[0, 0], [W - 1, H - 1], [0, H - 1], [W - 1, 0]]]).astype(np.uint32)
        g1[f"cam{ci}"] = camera_to_array(cam)
        g1[f"xy{ci}"] = xy
        g1[f"rays{ci}_1"] = O.cast_rays("ref", cam, Params(), xy)
        g1[f"rays{ci}_9"] = O.cast_rays("ref", cam, Params(aa=DEFAULT_AA), xy)
    g1["aa9"] = np.array(DEFAULT_AA, np.float32)
    np.savez_compressed(os.path.join(GOLD, "camera_rays.npz"), **g1)

    # ---------- G2/G3 intersections ----------
    fs = wl.scene_bunny(5, third="glass2")          # 3 spheres, 500-face mesh, 5 planes
    lone = FlatScene()                              # a scene with an ungated one-face model (lone Triangle) as well
    tri = [[[-2, -1, -6]], [[1, -1.5, -7]], [[-0.5, 2, -5]], [[0.1, 0.2, 0.9]], [[-0.3, 0.1, 1.1]], [[0, 0.5, 0.7]]]
    lone.add_mesh(Mesh(*[np.array(a, np.float32) for a in tri], np.array([0, 0, 0, -1], np.float32)),
                  material((1, 0, 0)))
    lone.add_sphere((0, 0, -8), 1.0, material((0, 1, 0)))
    cam = ref_camera(96, 72, move=(-0.5, 0, -0.25))
    cam_rays = O.cast_rays("ref", cam, Params(), [(x, y) for y in range(72) for x in range(96)]).reshape(-1, 6)
    rays = np.concatenate([cam_rays, random_rays(4000, 3), edge_rays(fs)])
    g2 = {("scene_" + k): v for k, v in fs.to_arrays().items()}
    g2.update({("lone_" + k): v for k, v in lone.to_arrays().items()})
    g2["rays"] = rays
    R = O.Reference(fs)
    for obj in [-1] + list(range(len(fs.objects))):
        h, d, n = R.intersect(rays, obj)
        g2[f"hit_{obj}"], g2[f"dist_{obj}"], g2[f"normal_{obj}"] = h, d, n
    md = rng.uniform(0.05, 14, rays.shape[0]).astype(np.float32)
    g2["max_dist"] = md
    g2["occluded"] = R.occluded(rays, md)
    RL = O.Reference(lone)
    h, d, n = RL.intersect(rays, -1)
    g2["lone_hit"], g2["lone_dist"], g2["lone_normal"] = h, d, n
    np.savez_compressed(os.path.join(GOLD, "intersect.npz"), **g2)

    # ---------- G5 frames ----------
    g5 = {}
    scenes = {"analytic": wl.scene_analytic(), "glass2": wl.scene_analytic(third="glass2"),
              "mesh": wl.scene_bunny(5, third="glass2")}
    cam_s = ref_camera(160, 120, move=(-0.5, 0, -0.25))
    cam_l = ref_camera(800, 600, move=(-0.5, 0, -0.25))
    g5["cam_small"], g5["cam_large"] = camera_to_array(cam_s), camera_to_array(cam_l)
    g5["aa9"] = np.array(DEFAULT_AA, np.float32)
    for name, sc in scenes.items():
        g5.update({f"{name}_scene_{k}": v for k, v in sc.to_arrays().items()})
        R = O.Reference(sc)
        for depth in (0, 1, 8, 10):
            g5[f"{name}_d{depth}_1"] = R.render(cam_s, Params(max_depth=depth))[0]
        g5[f"{name}_d10_9"] = R.render(cam_s, Params(aa=DEFAULT_AA))[0]
        rgb, _ = R.trace(Params(), cam_rays[::5])
        g5[f"{name}_trace_rgb"] = rgb
        if name != "mesh":
            big = R.render(cam_l, Params())[0]
            g5[f"{name}_800x600_bytesum"] = np.int64(big.astype(np.int64).sum())
            g5[f"{name}_800x600_rowsum"] = big.astype(np.int64).sum(axis=(1, 2))
    big9 = O.Reference(scenes["analytic"]).render(cam_l, Params(aa=DEFAULT_AA))[0]
    g5["analytic_800x600_9spp_bytesum"] = np.int64(big9.astype(np.int64).sum())
    g5["trace_rays"] = cam_rays[::5]
    np.savez_compressed(os.path.join(GOLD, "frames.npz"), **g5)
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))
    print("known answers (SURVEY 8c): analytic", int(g5["analytic_800x600_bytesum"]), "9spp",
          int(g5["analytic_800x600_9spp_bytesum"]), "glass2", int(g5["glass2_800x600_bytesum"]))


if __name__ == "__main__":
    main()
