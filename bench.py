#!/usr/bin/env python3
"""bench.py - Mrays/s and ms/frame of the render hot path on N MI355X (one process per GPU).

A "step" is one full frame of the workload (default: the reference's scene.yaml "bunny scene" at 1920x1080 with
its own reflective materials, depth 10, 1 ray per pixel; the mesh is the 69,620-face procedural stand-in for
bun_zipper.ply, which does not ship with the reference). The scene is resident in HBM before the timed region;
the frame stays in HBM (device framebuffer).

N = 1: rt_render_device renders the whole frame.
N > 1: strong scaling of the same frame. 32x32-pixel tiles are dealt round-robin to the ranks (scene replicated),
       each rank renders its tiles with rt_render_tiles_device, the tile buffers are gathered to rank 0 with
       torch.distributed (backend nccl = RCCL over xGMI) and rank 0 scatters them into the row-major frame.

Rays = trace() invocations + shadow rays, counted on the device.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: FP32 vector peak


def algorithmic_work(work, rays, closest_rays, n_lights):
    """bytes/ray = B_node N_node + 48 N_tri + 16 N_sphere + 16 N_plane + 48 with B_node = 128 (one
    4-wide node = 4 child boxes, i.e. two of the survey's 64-B two-box nodes);
    flops/ray = 96 N_node + 47 N_tri + 35 N_sphere + 13 N_plane + S (S = 17 + 60 per light for closest-hit rays)."""
    b = 128 * work.nodes_visited + 48 * work.tris_tested + 16 * work.spheres_tested + 16 * work.planes_tested + 48 * rays
    f = (96 * work.nodes_visited + 47 * work.tris_tested + 35 * work.spheres_tested + 13 * work.planes_tested
         + closest_rays * (17 + 60 * n_lights))
    return b, f


def cpu_baseline(fs, cam, params, dev, np):
    """The reference's own objects (oracle/_ref, kind 'reference') or the C restatement ('port') on a bounded sample:
    8 full-height strips, 4 pixels wide, evenly spread over the frame (rows under OpenMP schedule(dynamic,10) as the
    reference's render())."""
    from oracle import oracle as O
    kind = "reference" if O.have_ref() else "port"
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")), 16)
    backend = O.Reference(fs) if kind == "reference" else O.Oracle(fs)
    W, H = cam.imWidth, cam.imHeight

    def run(width):
        strips = [(min(W - width, int(W * (k + 0.5) / 8)), 0, width, H) for k in range(8)]
        rays, secs, same = 0, 0.0, True
        for rect in strips:
            g, st = dev.render(cam, params, rect=rect)
            rays += st.trace_calls + st.shadow_rays
            t = time.time()
            c, _ = backend.render(cam, params, rect=rect, threads=cores)
            secs += time.time() - t
            x0, y0, w, h = rect
            d = np.abs(g[:, x0:x0 + w].astype(np.int16) - c[:, x0:x0 + w].astype(np.int16)).max(axis=2)
            same = same and float((d <= 1).mean()) >= 0.999
        return rays, secs, same

    if fs.num_faces > 1000000:
        # the reference tests EVERY face per ray (src/primitives.cpp:376-382): at 10 M faces a few thousand rays are all
        # that fit the time bound, so the sample is one 6x6 crop on the middle of the field at 1 sample per pixel
        from raytracer_amd.scene import Params
        p1 = Params(background=tuple(params.c.background), max_depth=params.c.max_depth, aa=())
        rect = (W // 2 - 3, int(0.63 * H) - 3, 6, 6)
        g, st = dev.render(cam, p1, rect=rect)
        t = time.time()
        c, _ = backend.render(cam, p1, rect=rect, threads=cores)
        secs = time.time() - t
        x0, y0, w, h = rect
        d = np.abs(g[y0:y0 + h, x0:x0 + w].astype(np.int16) - c[y0:y0 + h, x0:x0 + w].astype(np.int16)).max(axis=2)
        rays = st.trace_calls + st.shadow_rays
        return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": kind,
                "sample": f"6x6 px crop on the middle of the field at 1 spp ({rays} rays, {secs:.1f} s CPU wall)",
                "gpu_matches_cpu_on_sample": bool(float((d <= 1).mean()) >= 0.999)}
    width = 4
    rays, secs, same = run(width)                    # probe, then size the sample to about 15 s of CPU wall time
    target = int(width * 15.0 / max(secs, 1e-3))
    if target > width:
        width = max(4, min(W // 8, 160, target))
        rays, secs, same = run(width)
    return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": kind,
            "sample": f"8 strips of {width}x{H} px of the same frame, evenly spread ({rays} rays, {secs:.1f} s CPU wall)",
            "gpu_matches_cpu_on_sample": bool(same)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", default="bunny_1080p")
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="library calls in flight (each in its own scene workspace and stream); 0 = 2, or 1 when one frame is several passes already")
    ap.add_argument("--frames-per-batch", type=int, default=32,
                    help="frames (steps) rendered per library call, rt_render_batch_device: every launch then carries "
                         "that many frames' rays (1..32; 1 = one call per frame)")
    ap.add_argument("--bvh", default="host", choices=("host", "device"),
                    help="BVH builder: the host's SAH tree (default) or the GPU's linear BVH (RT_SCENE_BVH_DEVICE)")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="rehearsal only: a 1-rank job (launched through torch.distributed.run) takes the N-rank path - tiles, "
                         "RCCL gather to itself, untile - so that the exchange code runs on the real backend with one GPU")
    ap.add_argument("--watchdog", type=int, default=0, help="dump all Python stacks and exit after this many seconds")
    args = ap.parse_args()
    if args.watchdog:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    import numpy as np
    import torch
    import torch.distributed as dist

    from raytracer_amd import cabi, workloads as wl
    from raytracer_amd.cabi import rt_stats
    from raytracer_amd.scene import DeviceScene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    multi = world > 1 or args.rehearse_exchange        # the N-rank code path
    if multi:
        if args.backend == "nccl":
            # No device_id: the RCCL communicator is then created by the first collective (the first warmup call's gather),
            # i.e. AFTER the scenes have created their streams. Created before them (device_id = eager), RCCL's own streams
            # shift the scenes' streams onto other hardware queues and every frame takes 1.5x as long (one rank, exchange
            # included: 0.46 ms against 0.31; tools/exp_rehearse2.sh). BENCH_EAGER_PG=1 brings the eager order back.
            if os.environ.get("BENCH_EAGER_PG"):
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group("nccl")
        else:
            dist.init_process_group(args.backend)

    fs, cam, params = wl.make_workload(args.workload)
    t0 = time.time()
    dev = DeviceScene(fs, device=local, bvh=args.bvh)
    build_s = time.time() - t0
    lib = dev.lib
    W, H = cam.imWidth, cam.imHeight
    tw = th = args.tile

    # Batches and calls in flight. A frame ends in a chain of small, latency-bound launches (deep bounce levels, fold)
    # that leave most of the chip idle, and a small frame - or one rank's share of a frame - is bound by the host's
    # launch rate. So (1) the steps are rendered B frames per library call (rt_render_batch_device: the frames' rays share
    # every launch) and (2) two calls are in flight, each with its own scene workspace, stream and framebuffers. Every
    # frame is still rendered completely and compared with the census frame below.
    # ... as many as fill about one 8 M-sample pass of this rank (a batch of large frames would only add passes)
    # a frame of several 8 M-sample passes fills the chip by itself: a second call in flight only competes for the caches
    # (4K 16 spp glass scene: 55 ms one call at a time, 67 ms with two)
    in_flight = args.frames_in_flight or (1 if (W * H * params.spp) // world >= 2 * (8 << 20) else 2)
    # ... and at least two calls per slot inside the timed region, so that the exchange of one call (gather, untile)
    # overlaps the rendering of the next whatever --steps is
    B = max(1, min(32, args.frames_per_batch, (8 << 20) // max(1, (W * H * params.spp) // world), args.steps // (2 * in_flight)))
    cams = [cam] * B
    frame_bytes = H * W * 3

    class Slot:
        pass
    slots = []
    for k in range(in_flight):
        sl = Slot()
        sl.dev = dev if k == 0 else DeviceScene(fs, device=local, bvh=args.bvh)
        # the scene's own stream, seen by torch: renders, the RCCL gather and the untile are ordered on it
        sl.stream = torch.cuda.ExternalStream(sl.dev.stream(), device=torch.device("cuda", local))
        sl.frames = torch.zeros((B, H, W, 3), dtype=torch.uint8, device="cuda")
        sl.frame = sl.frames[0]
        sl.last_b = 0
        if multi:
            sl.n_slot = lib.rt_tiles_owned(W, H, tw, th, 0, world)       # part 0 owns the most tiles
            sl.part_bytes = sl.n_slot * tw * th * 3                       # one frame's tiles of one part
            sl.mine = torch.zeros((B * sl.part_bytes,), dtype=torch.uint8, device="cuda")
            sl.allt = torch.zeros((world * B * sl.part_bytes,), dtype=torch.uint8, device="cuda") if rank == 0 else None
            sl.gathered = list(sl.allt.chunk(world)) if rank == 0 else None     # the gather lands directly in the untile input
        slots.append(sl)
    stream = slots[0].stream.cuda_stream
    frame = slots[0].frame

    # ray census of one frame (untimed), with the traversal work counters on
    st = rt_stats()
    work_k = None
    if rank == 0:
        dev.set_count_work(True)
        dev.render_device(cam, params, frame.data_ptr(), stream=stream, stats=st)
        work_k = [dev.get_work(k) for k in range(5)]
        work_hits = st.shadow_rays // max(len(fs.lights), 1)      # shaded hits (each shoots one ray per light)
        dev.set_count_work(False)
        ref_frame = frame.clone()
    rays_per_frame = st.trace_calls + st.shadow_rays

    def step(i, b=1):
        """Call number i: the next b (<= B) frames, on slot i % slots."""
        sl = slots[i % len(slots)]
        sl.last_b = b
        h = sl.stream.cuda_stream
        with torch.cuda.stream(sl.stream):
            if not multi:
                sl.dev.render_batch_device(cams[:b], params, sl.frames.data_ptr(), frame_bytes, stream=h)
                return
            sl.dev.render_tiles_batch_device(cams[:b], params, (tw, th), rank, world, sl.mine.data_ptr(), sl.part_bytes, stream=h)
            if "gather" in os.environ.get("BENCH_SKIP", ""):     # diagnostics: time the path without its exchange / untile
                pass
            elif args.backend == "nccl":
                dist.gather(sl.mine, sl.gathered, dst=0)        # ONE exchange for the b frames
            else:       # gloo rehearsal: gloo gathers host tensors only
                sl.stream.synchronize()
                host = sl.mine.cpu()
                hg = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, hg, dst=0)
                if rank == 0:
                    sl.allt.copy_(torch.cat(hg))
            if rank == 0 and "untile" not in os.environ.get("BENCH_SKIP", ""):       # part q's tiles of frame k start at tile slot q * B * n_slot + k * n_slot; one launch
                cabi.check(lib, lib.rt_untile_batch_device(local, W, H, tw, th, world, B * sl.n_slot, b, sl.part_bytes,
                                                           frame_bytes, C.c_void_p(sl.allt.data_ptr()),
                                                           C.c_void_p(sl.frames.data_ptr()), C.c_void_p(h)))

    def run_steps(n):
        """n frames as ceil(n / B) calls."""
        done, call = 0, 0
        while done < n:
            b = min(B, n - done)
            step(call, b)
            done += b
            call += 1

    def sync():
        if multi and "barrier" not in os.environ.get("BENCH_SKIP", ""):
            dist.barrier(device_ids=[local]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    def overflowed():
        """Did an asynchronous frame of any slot (on any rank) run out of ray pool? The scenes adapt (smaller passes)."""
        bad = 0
        for sl in slots:
            try:
                sl.dev.status()
            except cabi.RtError:
                bad = 1
        if multi and "barrier" not in os.environ.get("BENCH_SKIP", ""):
            f = torch.tensor([bad], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(f, op=dist.ReduceOp.MAX)
            bad = int(f.item())
        return bad != 0

    # untimed: W warmup steps, repeated (at most 8 times) while a scene is still learning its pass size
    adapt_rounds = 0
    for adapt_rounds in range(8):
        # (rounded up to whole batches on every slot, so that no workspace grows inside the timed region)
        run_steps(-(-args.warmup // (B * len(slots))) * B * len(slots))
        sync()
        if not overflowed():
            break
    t = time.perf_counter()
    run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t
    if overflowed():
        raise SystemExit("bench.py: a timed frame ran out of ray pool; the measurement is void")
    if multi and "barrier" not in os.environ.get("BENCH_SKIP", ""):
        e = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        elapsed = float(e.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "Mrays/s", "value": rays_per_frame * args.steps / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "ms_per_frame": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "width": W, "height": H, "spp": params.spp,
                       "max_depth": params.c.max_depth, "triangles": fs.num_faces, "objects": len(fs.objects),
                       "lights": len(fs.lights), "rays_per_frame": rays_per_frame,
                       "trace_calls": st.trace_calls, "shadow_rays": st.shadow_rays,
                       "level_rays": [int(x) for x in st.level_rays[:params.c.max_depth + 1]],
                       "parallelism": "1 GPU" if world == 1 else f"{world} GPUs: {tw}x{th} tiles round-robin, RCCL gather",
                       "mesh": "procedural geodesic blob (no .ply ships with the reference)",
                       "scene_build_s": round(build_s, 3), "bvh": args.bvh,
                       "bvh_build_s": round(dev.bvh_info().build_seconds, 4)},
        }
        identical = all(bool(torch.equal(sl.frames[k], ref_frame)) for sl in slots for k in range(sl.last_b))
        out["frame_identical_to_1gpu"] = identical
        out["config"]["frames_in_flight"] = len(slots)
        out["config"]["frames_per_batch"] = B
        out["config"]["pool_adapt_rounds"] = adapt_rounds
        if not multi and (len(slots) > 1 or B > 1):
            # latency view: the same frames one at a time (one workspace, one stream)
            t = time.perf_counter()
            for _ in range(args.steps):
                with torch.cuda.stream(slots[0].stream):
                    dev.render_device(cam, params, frame.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            out["one_frame_at_a_time"] = {"ms_per_frame": (time.perf_counter() - t) / args.steps * 1e3}
        if not multi:
            # The frame's time is in four kernels: k_extend_wide / k_extend_quad (closest hits: analytic objects one ray
            # per lane, meshes one ray per quad) and k_shade_wide / k_shade_quad (shadow rays + Phong). A call with stats
            # runs them back to back on ONE stream with HIP events around every launch (the untimed path overlaps the
            # shade kernels with deeper extend launches, which would blur per-kernel durations).
            reps = 10 if W * H * params.spp <= 64 << 20 else 2      # a stats call times at most 2048 launches of a frame
            names = ("k_extend_wide", "k_shade_wide", "k_extend_quad", "k_shade_quad<0|1>", "k_shade_quad<2>")
            NK = len(names)
            ms = [0.0] * NK
            n = [0] * NK
            frame_ms = 0.0
            for _ in range(reps):
                s2 = rt_stats()
                dev.render_device(cam, params, frame.data_ptr(), stream=stream, stats=s2)
                for k in range(NK):
                    ms[k] += s2.kernel_ms_by[k]; n[k] += s2.kernel_launches_by[k]
                frame_ms += s2.kernel_ms
            per_kernel = {}
            for ki, name in enumerate(names):
                if not n[ki]:
                    continue
                wk = work_k[ki]
                # rays whose record the kernel reads or writes (48 B each in the survey's formula): a wide kernel sees
                # every ray of its kind, a quad kernel the rays it traverses (16 per batch at most)
                rays_k = (st.trace_calls if ki == 0 else st.shadow_rays) if ki < 2 else min(16 * wk.batches, st.trace_calls if ki == 2 else st.shadow_rays)
                b, f = algorithmic_work(wk, rays_k, work_hits if ki == 1 else 0, len(fs.lights))
                lp = n[ki] / reps
                avg = ms[ki] / n[ki]
                per_kernel[name] = {
                    "avg_launch_ms": avg, "launches_per_frame": lp, "ms_per_frame_serial": ms[ki] / reps,
                    "algorithmic_bytes_per_launch": b / lp, "GB/s": b / lp / (avg * 1e-3) / 1e9,
                    "TFLOP/s": f / lp / (avg * 1e-3) / 1e12,
                    "work_per_frame": {"bvh_nodes": wk.nodes_visited, "tri_tests": wk.tris_tested,
                                       "sphere_tests": wk.spheres_tested, "plane_tests": wk.planes_tested,
                                       "wave_node_steps": wk.wave_node_steps, "wave_tri_steps": wk.wave_tri_steps,
                                       "batches": wk.batches}}
            # HBM bytes per launch from the PMC passes of tools/exp_traffic.sh (FETCH_SIZE x2 + WRITE_SIZE, separate runs;
            # committed under profiles/r*/traffic_<workload>.json). bench.py cannot collect counters itself.
            import glob
            traffic = {}
            tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"traffic_{args.workload}.json")))
            if tf:
                traffic = {k: v["hbm_bytes_per_launch"] for k, v in json.load(open(tf[-1]))["kernels"].items()}
            for name in per_kernel:     # all launches of the kernel, whatever its template arguments
                tot = [(v, json.load(open(tf[-1]))["kernels"][k]["launches"]) for k, v in traffic.items() if k.split("<")[0] == name] if tf else []
                per_kernel[name]["hbm_bytes_per_launch_pmc"] = (sum(v * l for v, l in tot) / sum(l for _, l in tot)) if tot else None
            # Roofline of the dominant LAUNCH: a frame at max_depth 0 runs exactly the level-0 launches (the heaviest of
            # the frame), so its counters and its HIP-event durations belong to single launches, which rocprofv3 lists
            # under the same templated names.
            from raytracer_amd.scene import Params
            p0 = Params(background=tuple(params.c.background), max_depth=0, aa=params.aa)
            dev.set_count_work(True)
            s0 = rt_stats()
            dev.render_device(cam, p0, frame.data_ptr(), stream=stream, stats=s0)
            work0 = [dev.get_work(k) for k in range(NK)]
            dev.set_count_work(False)
            ms0, n0 = [0.0] * NK, [0] * NK
            for _ in range(reps):
                s2 = rt_stats()
                dev.render_device(cam, p0, frame.data_ptr(), stream=stream, stats=s2)
                for k in range(NK):
                    ms0[k] += s2.kernel_ms_by[k]; n0[k] += s2.kernel_launches_by[k]
            dev.render_device(cam, params, frame.data_ptr(), stream=stream)      # leave the full frame in the buffer
            torch.cuda.synchronize()
            # (a scene whose level 0 is traversed as packets has no quad launches at level 0: the per-lane kernels'
            # PACKET instantiations do the whole level)
            pk = "true" if (fs.num_faces and not n0[2]) else "false"
            rocprof_names = (f"k_extend_wide<0, false, {pk}>", f"k_shade_wide<false, false, {pk}>", "k_extend_quad<0, false, false>",
                             "k_shade_quad<1, false, false>", "k_shade_quad<2, false, false>")
            dk = max(range(NK), key=lambda k: ms0[k] / max(n0[k], 1))
            avg = ms0[dk] / n0[dk]
            lp = n0[dk] / reps
            rays_k = (s0.trace_calls if dk == 0 else s0.shadow_rays) if dk < 2 else min(16 * work0[dk].batches, s0.trace_calls if dk == 2 else s0.shadow_rays)
            b, f = algorithmic_work(work0[dk], rays_k, s0.shadow_rays // max(len(fs.lights), 1) if dk == 1 else 0, len(fs.lights))
            gbs, tfl = b / lp / (avg * 1e-3) / 1e9, f / lp / (avg * 1e-3) / 1e12
            out["roofline"] = {
                "bound": "hbm", "kernel": rocprof_names[dk], "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": traffic.get(rocprof_names[dk]),
                "avg_launch_ms": avg, "launches_per_frame": lp, "algorithmic_bytes_per_launch": b / lp,
                "units_per_launch": {"rays": rays_k, "bvh_nodes": work0[dk].nodes_visited, "tri_tests": work0[dk].tris_tested,
                                     "sphere_tests": work0[dk].spheres_tested, "plane_tests": work0[dk].planes_tested},
                "valu": {"achieved": tfl, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / VALU_PEAK_TFLOPS},
                "level0_launch_ms": {rocprof_names[k]: ms0[k] / n0[k] for k in range(NK) if n0[k]},
                "serial_frame_ms": frame_ms / reps, "kernels": per_kernel,
            }
            # What the launch really moves over HBM (PMC bytes / the same duration). `achieved` follows SURVEY.md 8d to
            # the letter: every node a RAY visits is priced at 128 B. A packet fetches a node once for its 64 rays,
            # through the scalar cache, and the trees of this workload live in L2, so `frac` can exceed 1: those bytes
            # never reach HBM and the HBM roofline does not bind the kernel.
            tr = traffic.get(rocprof_names[dk])
            if tr:
                rate = tr / (avg * 1e-3) / 1e9
                out["roofline"]["hbm_measured"] = {"achieved": rate, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rate / HBM_PEAK_GBS}
            # What does bind the frame: VALU issue. SQ_INSTS_VALU of all launches of one frame (PMC pass committed under
            # profiles/, tools/exp_pmc_sq.sh) x 4 cycles per wave64 instruction / (1024 SIMDs x 2.4 GHz), against the
            # measured time per frame.
            pf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_sq_{args.workload}.json")))
            if pf:
                n_valu = json.load(open(pf[-1]))["valu_instructions_per_frame"]
                busy_ms = n_valu * 4 / (1024 * 2.4e9) * 1e3
                out["roofline"]["valu_issue"] = {"valu_instructions_per_frame": n_valu, "cycles_per_instruction": 4, "simds": 1024,
                                                 "clock_GHz": 2.4, "busy_ms_per_frame": busy_ms, "frac": busy_ms / out["ms_per_step"]}
            # the boundary as the reference has it (rt_render: host framebuffer, synchronous): PCIe-inclusive, never `value`
            host_frame = np.zeros((H, W, 3), np.uint8)
            dev.render(cam, params, out=host_frame, want_stats=False)
            t = time.perf_counter()
            for _ in range(5):
                dev.render(cam, params, out=host_frame, want_stats=False)
            host_ms = (time.perf_counter() - t) / 5 * 1e3
            out["host_framebuffer"] = {"ms_per_frame": host_ms, "Mrays/s": rays_per_frame / host_ms / 1e3,
                                       "note": "rt_render into a host buffer (frame copied over PCIe, call synchronous)"}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(fs, cam, params, dev, np)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier(device_ids=[local]) if args.backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
