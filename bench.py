#!/usr/bin/env python3
"""bench.py - Mrays/s and ms/frame of the render hot path on N MI355X (one process per GPU).

A "step" is one full frame of the workload (default: the reference's scene.yaml "bunny scene" at 1920x1080 with
its own reflective materials, depth 10, 1 ray per pixel; the mesh is the 69,620-face procedural stand-in for
bun_zipper.ply, which does not ship with the reference). The scene is resident in HBM before the timed region.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child
processes, before anything touches a GPU) and relays rank 0's JSON line; under torch.distributed.run the ranks are
already there and WORLD_SIZE must equal --gpus.

N = 1: rt_render_batch_device renders whole frames into a device framebuffer.
N > 1: strong scaling of the same frames. 32x32-pixel tiles are dealt round-robin to the ranks (scene replicated;
       the reference's own model is the tile pool of src/raytracer.cpp:583-617), each rank renders its tiles, the
       tile buffers are gathered to rank 0 with torch.distributed (backend nccl = RCCL over xGMI) and rank 0 scatters
       them into the row-major frames.

What the JSON line's times are:
  value / ms_per_step / ms_per_frame_batched_throughput   K frames, device framebuffer, batches of B frames with
        DIFFERENT cameras per call, two calls in flight: whole-job throughput with everything resident in HBM
  ms_per_frame          SURVEY.md 8(d) to the letter: wall time of one synchronous rt_render of the full image into a
                        HOST buffer (D2H included), median of >= 10 frames after 2 warm-ups
  ms_per_frame_latency  one rt_render_device call + wait, device framebuffer (median)
  ms_per_frame_host     frames delivered to page-locked HOST memory by the pipelined path
                        (rt_render_batch_host_async, two scenes alternating: the copy of one call overlaps the next)
Rays = trace() invocations + shadow rays, counted on the device.
"""
import argparse
import ctypes as C
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: FP32 vector peak (= 1024 SIMDs x 2.4 GHz x 64 lanes x 2 / 2 cycles)
SIMDS, CLOCK_GHZ = 1024, 2.4
# A wave64 VALU instruction issues in 2 cycles on a CDNA4 SIMD-32 (cdna_hip_programming.md section 1); measured with
# tools/valu_rate.hip: v_fma/v_mul/v_add reach that with several waves per SIMD, v_min/v_max/v_cmp/v_cndmask/DPP/
# v_readlane take 4, v_rcp/v_sqrt 8 (profiles/r2/valu_rate.txt). The peak below is the 2-cycle rate.
VALU_ISSUE_PEAK_GINST = SIMDS * CLOCK_GHZ / 2.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", default="bunny_1080p")
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed run: no latency/host/roofline/CPU legs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="library calls in flight (each in its own scene workspace and stream); 0 = 2, or 1 when one frame is several passes already")
    ap.add_argument("--frames-per-batch", type=int, default=32,
                    help="frames (steps) rendered per library call, rt_render_batch_device: every launch then carries "
                         "that many frames' rays (1..32; 1 = one call per frame)")
    ap.add_argument("--batch-samples", type=int, default=64 << 20,
                    help="a batch holds at most this many level-0 samples (frames x pixels x samples per pixel, per rank)")
    ap.add_argument("--batch-plan", default="",
                    help="experiments: frames per call of the timed region, one entry per call (e.g. 14,6); overrides --frames-per-batch")
    ap.add_argument("--same-camera", action="store_true", help="experiments: every frame of a batch from the reference camera")
    ap.add_argument("--bvh", default="host", choices=("host", "device"),
                    help="BVH builder: the host's SAH tree (default) or the GPU's linear BVH (RT_SCENE_BVH_DEVICE)")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="rehearsal only: a 1-rank job (launched through torch.distributed.run) takes the N-rank path - tiles, "
                         "RCCL gather to itself, untile - so that the exchange code runs on the real backend with one GPU")
    ap.add_argument("--rehearse-launcher", action="store_true",
                    help="rehearsal only (no GPU needed): the ranks rendezvous, all-gather their ranks and rank 0 prints "
                         "what it saw; checks that --gpus N really is N processes")
    ap.add_argument("--ply", default=None,
                    help="a mesh file (ASCII .ply, e.g. the Stanford bun_zipper.ply the reference's scene.yaml:98-104 names) in place "
                         "of the procedural blob wherever the workload's scene holds scene.yaml's mesh object")
    ap.add_argument("--scene", default=None,
                    help="a scene file in the reference's schema (src/scene.cpp:32-178) in place of the workload's scene; frame "
                         "size, samples per pixel and depth stay the workload's")
    ap.add_argument("--watchdog", type=int, default=0, help="dump all Python stacks and exit after this many seconds")
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N without a launcher: N child processes of this script, one per GPU, rendezvous on 127.0.0.1. The parent
    never touches a GPU (no re-exec of a process that has). Rank 0's stdout is relayed; exit code = the worst child's."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [c for c in codes if c != 0]
    sys.exit(bad[0] if bad else 0)


def algorithmic_work(work, rays, closest_rays, n_lights):
    """bytes/ray = B_node N_node + 48 N_tri + 16 N_sphere + 16 N_plane + 48 with B_node = 128 (one
    4-wide node = 4 child boxes, i.e. two of the survey's 64-B two-box nodes);
    flops/ray = 96 N_node + 47 N_tri + 35 N_sphere + 13 N_plane + S (S = 17 + 60 per light for closest-hit rays).
    N_* must be PER-RAY counts (a scene created with RT_SCENE_NO_PACKETS: every ray traversed by itself)."""
    b = 128 * work["nodes"] + 48 * work["tris"] + 16 * work["spheres"] + 16 * work["planes"] + 48 * rays
    f = (96 * work["nodes"] + 47 * work["tris"] + 35 * work["spheres"] + 13 * work["planes"]
         + closest_rays * (17 + 60 * n_lights))
    return b, f


def cpu_baseline(fs, cam, params, dev, np):
    """The reference's own objects (oracle/_ref, kind 'reference') or the C restatement ('port') on a bounded sample:
    8 full-height strips, evenly spread over the frame (rows under OpenMP schedule(dynamic,10) as the reference's
    render())."""
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
        # that fit the time bound, so the sample is one 16x16 crop on the middle of the field at 1 sample per pixel
        from raytracer_amd.scene import Params
        p1 = Params(background=tuple(params.c.background), max_depth=params.c.max_depth, aa=())
        rect = (W // 2 - 8, int(0.63 * H) - 8, 16, 16)
        g, st = dev.render(cam, p1, rect=rect)
        t = time.time()
        c, _ = backend.render(cam, p1, rect=rect, threads=cores)
        secs = time.time() - t
        x0, y0, w, h = rect
        d = np.abs(g[y0:y0 + h, x0:x0 + w].astype(np.int16) - c[y0:y0 + h, x0:x0 + w].astype(np.int16)).max(axis=2)
        rays = st.trace_calls + st.shadow_rays
        return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": kind,
                "sample": f"16x16 px crop on the middle of the field at 1 spp ({rays} rays, {secs:.1f} s CPU wall)",
                "gpu_matches_cpu_on_sample": bool(float((d <= 1).mean()) >= 0.999)}
    width = 4
    rays, secs, same = run(width)                    # probe, then size the sample to about 15 s of CPU wall time
    target = int(width * 15.0 / max(secs, 1e-3))
    if target > width:
        width = max(4, min(W // 8, target))
        rays, secs, same = run(width)
    return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": kind,
            "sample": f"8 strips of {width}x{H} px of the same frame, evenly spread ({rays} rays, {secs:.1f} s CPU wall)",
            "gpu_matches_cpu_on_sample": bool(same)}


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def main():
    args = parse_args()
    if args.watchdog:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        launch_ranks(args)                           # does not return
    world = int(env_world or "1")
    if world != args.gpus and not args.rehearse_exchange:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, "
                         f"or without a launcher (bench.py then starts the ranks itself)\n")
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.rehearse_launcher:
        dist.init_process_group("gloo")
        seen = [None] * world
        dist.all_gather_object(seen, (rank, os.getpid()))
        if rank == 0:
            print(json.dumps({"rehearse_launcher": True, "n_gpus": args.gpus, "rccl_ranks": dist.get_world_size(),
                              "ranks_seen": sorted(r for r, _ in seen), "distinct_processes": len({p for _, p in seen})}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    from raytracer_amd import cabi, workloads as wl
    from raytracer_amd.cabi import rt_stats
    from raytracer_amd.scene import DeviceScene, Params

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

    fs, cam, params = wl.make_workload(args.workload, ply=args.ply, scene_file=args.scene)
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
    # frame is still rendered completely and compared with its camera's census frame below.
    # ... as many as fill one pass of this rank (--batch-samples: 2^26 level-0 samples, the scenes' default pass on an
    # MI355X; bunny_1080p per frame: 4 per call 0.287 ms, 8: 0.274, 16: 0.267, 32: 0.258 - tools/exp_batchsize.sh; a batch of
    # large frames would only add passes)
    # a frame of several 8 M-sample passes fills the chip by itself: a second call in flight only competes for the caches
    # (4K 16 spp glass scene: 55 ms one call at a time, 67 ms with two)
    in_flight = args.frames_in_flight or (1 if (W * H * params.spp) // world >= 2 * (8 << 20) else 2)
    # ... and at least one call per slot inside the timed region, so that the calls in flight overlap (and, with several
    # ranks, the exchange of one call - gather, untile - the rendering of the other) whatever --steps is
    # (one GPU: the fewest calls of at most 32 frames, evenly sized - a short run is better off as ONE large call than as
    # one small call per slot: --steps 20 as 2 x 10 0.240-0.244 ms per frame, as 1 x 20 0.235, tools/exp_steps20b.sh)
    per_call = args.steps // in_flight if multi else -(-args.steps // max(1, -(-args.steps // 32)))
    B = max(1, min(32, args.frames_per_batch, args.batch_samples // max(1, (W * H * params.spp) // world), per_call))
    plan = [int(x) for x in args.batch_plan.split(",") if x]
    if plan:
        B = max(plan)
    # the frames of a batch are DIFFERENT pictures: camera k on a small circle through the reference camera (camera 0)
    cams = [cam] * B if args.same_camera else wl.orbit_cameras(W, H, B)
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
    torch.cuda.synchronize()       # the fills above ran on torch's stream, the renders run on the scenes' own
    stream = slots[0].stream.cuda_stream
    frame = slots[0].frame

    # ray census (untimed): one frame per camera of the batch, each kept as that camera's reference frame
    st = rt_stats()
    cam_rays, ref_frames = [], []
    if rank == 0:
        for k in range(B):
            if k and args.same_camera:
                cam_rays.append(cam_rays[0]); ref_frames.append(ref_frames[0])
                continue
            sk = rt_stats()
            dev.render_device(cams[k], params, frame.data_ptr(), stream=stream, stats=sk)
            cam_rays.append(sk.trace_calls + sk.shadow_rays)
            ref_frames.append(frame.clone())
            if k == 0:
                st = sk
        torch.cuda.synchronize()
    rays_per_frame = st.trace_calls + st.shadow_rays       # of the reference camera

    def step(i, b=1, marks=None):
        """Call number i: the next b (<= B) frames, on slot i % slots. marks (a list): events on the slot's stream before
        the render, after it, after the gather and after the untile (the untimed phase diagnostics of an N-rank run)."""
        sl = slots[i % len(slots)]
        sl.last_b = b
        h = sl.stream.cuda_stream

        def mark():
            if marks is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(sl.stream)
                marks.append(e)
        with torch.cuda.stream(sl.stream):
            if not multi:
                sl.dev.render_batch_device(cams[:b], params, sl.frames.data_ptr(), frame_bytes, stream=h)
                return
            mark()
            sl.dev.render_tiles_batch_device(cams[:b], params, (tw, th), rank, world, sl.mine.data_ptr(), sl.part_bytes, stream=h)
            mark()
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
                mark()
                cabi.check(lib, lib.rt_untile_batch_device(local, W, H, tw, th, world, B * sl.n_slot, b, sl.part_bytes,
                                                           frame_bytes, C.c_void_p(sl.allt.data_ptr()),
                                                           C.c_void_p(sl.frames.data_ptr()), C.c_void_p(h)))
                mark()
            else:
                mark()
                mark()

    def run_steps(n, plan=()):
        """n frames as ceil(n / B) calls; returns the rays of those frames (rank 0 knows them)."""
        done, call, rays = 0, 0, 0
        while done < n:
            b = min(plan[call % len(plan)] if plan else B, n - done)
            step(call, b)
            rays += sum(cam_rays[:b]) if cam_rays else 0
            done += b
            call += 1
        return rays

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
    sync()
    t = time.perf_counter()
    timed_rays = run_steps(args.steps, plan)
    sync()
    elapsed = time.perf_counter() - t
    if overflowed():
        raise SystemExit("bench.py: a timed frame ran out of ray pool; the measurement is void")
    rccl_ranks = 1
    phases = None
    if multi:
        rccl_ranks = dist.get_world_size()
        # Where an N-rank frame's time goes (untimed, after the measurement): ONE call at a time, events on the slot's stream
        # around the rank's render, the gather and rank 0's untile; per frame, max and mean over the ranks. With calls in
        # flight the phases of one call overlap those of the other, so these sum to more than ms_per_step.
        per = [0.0, 0.0, 0.0]
        n_diag = 4
        for k in range(n_diag):
            marks = []
            step(k, B, marks)
            sync()
            for j in range(3):
                per[j] += marks[j].elapsed_time(marks[j + 1]) / (B * n_diag)
        mine = torch.tensor(per + [float(lib.rt_tiles_owned(W, H, tw, th, rank, world))], dtype=torch.float64,
                            device="cuda" if args.backend == "nccl" else "cpu")
        allp = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        allp = torch.stack(allp).cpu()
        phases = {name: {"max": float(allp[:, j].max()), "mean": float(allp[:, j].mean())}
                  for j, name in enumerate(("render_ms", "gather_ms", "untile_ms"))}
        phases["untile_ms"] = {"rank0": float(allp[0, 2])}
        phases["tiles_per_rank"] = [int(x) for x in allp[:, 3]]
        phases["note"] = ("per frame, one call at a time (no overlap between calls): render = this rank's tiles, gather = the "
                          "exchange to rank 0 as the rank's stream sees it (a peer's includes waiting for rank 0), untile on rank 0")
        if "barrier" not in os.environ.get("BENCH_SKIP", ""):
            e = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(e, op=dist.ReduceOp.MAX)
            elapsed = float(e.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "Mrays/s", "value": timed_rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "ms_per_frame_batched_throughput": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "rccl_ranks": rccl_ranks,
            "config": {"workload": args.workload, "width": W, "height": H, "spp": params.spp,
                       "max_depth": params.c.max_depth, "triangles": fs.num_faces, "objects": len(fs.objects),
                       "lights": len(fs.lights), "rays_per_frame": rays_per_frame,
                       "rays_per_frame_by_camera": cam_rays,
                       "trace_calls": st.trace_calls, "shadow_rays": st.shadow_rays,
                       "level_rays": [int(x) for x in st.level_rays[:params.c.max_depth + 1]],
                       "parallelism": "1 GPU" if world == 1 else f"{world} GPUs: {tw}x{th} tiles round-robin, RCCL gather",
                       "mesh": (f"scene file {args.scene}" if args.scene else f"{args.ply} (--ply)" if args.ply else
                                "procedural geodesic blob (no .ply ships with the reference)"),
                       "scene_build_s": round(build_s, 3), "bvh": args.bvh,
                       "bvh_build_s": round(dev.bvh_info().build_seconds, 4),
                       "cameras": "reference camera" if args.same_camera else
                                  f"{B} per batch: the reference camera and {B - 1} on a circle of radius 0.05 through it",
                       "value_is": "device-framebuffer throughput of batched frames (K steps / wall time of the timed region)"},
        }
        identical = all(bool(torch.equal(sl.frames[k], ref_frames[k])) for sl in slots for k in range(sl.last_b))
        out["frame_identical_to_1gpu"] = identical
        if phases is not None:
            out["phases"] = phases
        out["config"]["frames_in_flight"] = len(slots)
        out["config"]["frames_per_batch"] = B
        out["config"]["pool_adapt_rounds"] = adapt_rounds
        if not multi and not args.no_extras:
            extras(args, out, dev, slots, fs, cam, cams, params, st, rays_per_frame, cam_rays, ref_frames, np, torch)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier(device_ids=[local]) if args.backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


def extras(args, out, dev, slots, fs, cam, cams, params, st, rays_per_frame, cam_rays, ref_frames, np, torch):
    """One GPU only: latency and host-framebuffer figures, per-kernel table, roofline, CPU baseline."""
    from raytracer_amd.cabi import rt_stats
    from raytracer_amd.scene import DeviceScene, Params
    W, H = cam.imWidth, cam.imHeight
    B = len(cams)
    frame_bytes = H * W * 3
    stream = slots[0].stream.cuda_stream
    frame = slots[0].frame
    big = W * H * params.spp > (64 << 20)           # seconds per frame: fewer repetitions
    n_lat = 3 if big else 12

    # ---- latency: one call, one frame, wait (device framebuffer) ----
    ts = []
    for k in range(n_lat + 2):
        t = time.perf_counter()
        dev.render_device(cam, params, frame.data_ptr(), stream=stream)
        dev.status()
        ts.append((time.perf_counter() - t) * 1e3)
    out["ms_per_frame_latency"] = median(ts[2:])
    # ---- SURVEY.md 8(d) to the letter: synchronous rt_render of the full image into a host buffer ----
    pinned = torch.empty((max(B, 1), H, W, 3), dtype=torch.uint8).pin_memory()
    host_frame = pinned[0].numpy()
    ts = []
    for k in range(n_lat + 2):
        t = time.perf_counter()
        dev.render(cam, params, out=host_frame, want_stats=False)
        ts.append((time.perf_counter() - t) * 1e3)
    out["ms_per_frame"] = median(ts[2:])
    host_ok = bool(np.array_equal(host_frame, ref_frames[0].cpu().numpy()))
    pageable = np.zeros((H, W, 3), np.uint8)
    ts = []
    for k in range(min(n_lat, 5) + 1):
        t = time.perf_counter()
        dev.render(cam, params, out=pageable, want_stats=False)
        ts.append((time.perf_counter() - t) * 1e3)
    # ---- pipelined host delivery: batches into page-locked memory, the copy of one call under the next call's kernels ----
    # (ONE scene: it renders into two device buffers in turn and copies on a stream of its own, so call n+1's kernels run
    # over call n's copy. Two scenes alternating were measured slower, 0.44 against 0.41 ms per frame, three 0.51: their
    # ten streams share the process's four hardware queues; tools/exp_host_pipe.py)
    pins = [pinned, torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory()]
    n_host = max(4 * B, min(args.steps, 64 if not big else 2 * B))

    def host_steps(n):
        done, call, rays = 0, 0, 0
        while done < n:
            b = min(B, n - done)
            dev.render_batch_host_async(cams[:b], params, pins[call % 2].data_ptr(), frame_bytes)
            rays += sum(cam_rays[:b])
            done += b
            call += 1
        return rays
    host_steps(4 * B)
    for sl in slots:
        sl.dev.status()
    t = time.perf_counter()
    host_rays = host_steps(n_host)
    for sl in slots:
        sl.dev.status()
    host_el = time.perf_counter() - t
    out["ms_per_frame_host"] = host_el / n_host * 1e3
    host_ok = host_ok and all(bool(np.array_equal(pins[0][k].numpy(), ref_frames[k].cpu().numpy())) for k in range(B))
    out["host_framebuffer"] = {
        "ms_per_frame_pipelined": out["ms_per_frame_host"], "Mrays/s_pipelined": host_rays / host_el / 1e6, "frames": n_host,
        "ms_per_frame_sync_pinned": out["ms_per_frame"], "ms_per_frame_sync_pageable": median(ts[1:]),
        "frames_identical_to_device_frames": host_ok,
        "note": "PCIe-inclusive, never `value`: pipelined = rt_render_batch_host_async into page-locked memory, calls "
                "alternating on the scenes in flight; sync = rt_render, one frame per call, wall time of the call"}

    # ---- per-kernel table of one frame: launches back to back on ONE stream with HIP events around each ----
    reps = 10 if not big else 2      # a stats call times at most 2048 launches of a frame
    names = ("k_extend_wide", "k_shade_wide", "k_extend_quad", "k_shade_quad<0|1>", "k_shade_quad<2>")
    NK = len(names)

    def timed_frames(prm, n):
        ms, cnt, total = [0.0] * NK, [0] * NK, 0.0
        for _ in range(n):
            s2 = rt_stats()
            dev.render_device(cam, prm, frame.data_ptr(), stream=stream, stats=s2)
            for k in range(NK):
                ms[k] += s2.kernel_ms_by[k]; cnt[k] += s2.kernel_launches_by[k]
            total += s2.kernel_ms
        return ms, cnt, total
    ms, n, frame_ms = timed_frames(params, reps)
    per_kernel = {names[k]: {"avg_launch_ms": ms[k] / n[k], "launches_per_frame": n[k] / reps, "ms_per_frame_serial": ms[k] / reps}
                  for k in range(NK) if n[k]}

    # ---- roofline of the dominant LAUNCH. A frame at max_depth 0 runs exactly the level-0 launches
    # (the heaviest of the frame): one per kernel and PASS, so HIP-event durations, work counters and PMC counters there
    # belong to single launches, which rocprofv3 lists under the same templated names. ----
    p0 = Params(background=tuple(params.c.background), max_depth=0, aa=params.aa)
    ms0, n0, _ = timed_frames(p0, reps)
    # per-RAY algorithmic work (the nodes / triangles each ray's OWN traversal visits): the same depth-0
    # frame on a scene that never forms packets, with the counting kernels. Extend work = kinds 0 + 2, shade = 1 + 3 + 4.
    devq = DeviceScene(fs, device=dev.device, bvh=args.bvh, packets=False)
    devq.set_count_work(True)
    s0 = rt_stats()
    devq.render_device(cam, p0, frame.data_ptr(), stats=s0)
    wq = [devq.get_work(k) for k in range(NK)]
    devq.close()
    # what the packets do instead - node / triangle steps per WAVE (one scalar fetch serves 64 rays) - and the clock the
    # launch runs at: the counting instantiation of the same kernels stamps s_memtime (shader cycles) and s_memrealtime
    # (100 MHz) at the start and the end of every wave (MI355X_MICROARCH.md: clock = d s_memtime / d s_memrealtime x 100 MHz)
    dev.set_count_work(True)
    wp = None
    for _ in range(3):
        sp = rt_stats()
        dev.render_device(cam, p0, frame.data_ptr(), stream=stream, stats=sp)
        wp = [dev.get_work(k) for k in range(NK)]
    dev.set_count_work(False)
    dev.render_device(cam, params, frame.data_ptr(), stream=stream)      # leave the full frame in the buffer
    dev.status()

    def summed(ws):
        return {"nodes": sum(w.nodes_visited for w in ws), "tris": sum(w.tris_tested for w in ws),
                "spheres": sum(w.spheres_tested for w in ws), "planes": sum(w.planes_tested for w in ws)}
    per_ray = {0: summed([wq[0], wq[2]]), 1: summed([wq[1], wq[3], wq[4]])}
    # (a scene whose level 0 is traversed as packets has no quad launches at level 0: the per-lane kernels'
    # PACKET instantiations do the whole level)
    pk = "true" if (fs.num_faces and not n0[2]) else "false"
    # (the per-lane kernels' last template argument: k_extend_wide - whether it hands out level 0's groups in the order of the
    # previous frame's costs; k_shade_wide - 0 any level, 1 level 0, 2 level 0 in that order. Which instantiation a
    # workload's level 0 runs is read from the counters file below)
    rocprof_names = [f"k_extend_wide<0, false, {pk}, false>", f"k_shade_wide<false, false, {pk}, {1 if pk == 'true' else 0}>",
                     "k_extend_quad<0, false, false>", "k_shade_quad<1, false, false>", "k_shade_quad<2, false, false>"]
    rocprof_alt = [[f"k_extend_wide<0, false, {pk}, true>"], [f"k_shade_wide<false, false, {pk}, {m}>" for m in (0, 1, 2)]]
    # (a scene without meshes runs a level as ONE launch, k_trace_wide: closest hits and light loops; its work is both kinds')
    fused0 = bool(n0[0] and not n0[1] and s0.shadow_rays)
    if fused0:
        rocprof_names[0] = f"k_trace_wide<0, false, {pk}, false>"
        rocprof_alt[0] = ["k_trace_wide<0, true, false, false>", "k_trace_wide<0, false, true, true>"]
    dk = max(range(2), key=lambda k: ms0[k] / max(n0[k], 1))      # level 0 = the two per-lane launches
    avg_ms = ms0[dk] / n0[dk]
    launches = n0[dk] / reps                                       # per depth-0 frame = its passes
    rays_k = s0.trace_calls if dk == 0 else s0.shadow_rays
    hits = s0.shadow_rays // max(len(fs.lights), 1)
    b_frame, f_frame = algorithmic_work(per_ray[dk], rays_k, hits if dk == 1 else 0, len(fs.lights))
    units = dict(per_ray[dk])
    if fused0:
        b2, f2 = algorithmic_work(per_ray[1], s0.shadow_rays, hits, len(fs.lights))
        b_frame, f_frame, rays_k = b_frame + b2, f_frame + f2, s0.trace_calls + s0.shadow_rays
        units = {k: per_ray[0][k] + per_ray[1][k] for k in per_ray[0]}
    b_alg, f_alg = b_frame / launches, f_frame / launches         # the census is of the whole frame: ONE launch is a pass of it
    clock_mhz = 100.0 * wp[dk].wave_cycles_sum / max(wp[dk].wave_ticks_sum, 1)
    if not 500.0 < clock_mhz < 3000.0:
        clock_mhz = None
    peak_tfl = VALU_PEAK_TFLOPS * (clock_mhz / (CLOCK_GHZ * 1e3)) if clock_mhz else VALU_PEAK_TFLOPS
    alg_tfl = f_alg / (avg_ms * 1e-3) / 1e12
    alg_gbs = b_alg / (avg_ms * 1e-3) / 1e9
    # PMC counters of the same launch, collected by tools/collect_counters.sh (bench.py cannot run rocprofv3 on itself):
    # newest round first, and only if they were collected from THESE kernels (hash of sources + flags)
    from raytracer_amd.buildinfo import kernels_hash
    def round_of(path):
        try:
            return int(os.path.basename(os.path.dirname(path))[1:])
        except ValueError:
            return -1
    ctr_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"counters_{args.workload}.json")), key=round_of)
    ctr, ctr_src, stale = None, None, None
    if ctr_files:
        cj = json.load(open(ctr_files[-1]))
        stale = cj.get("kernels_hash") != kernels_hash()
        d0 = cj.get("depth0", {})
        for k in range(2):
            for alt in rocprof_alt[k]:
                if d0.get(alt, {}).get("launches", 0) > d0.get(rocprof_names[k], {}).get("launches", 0):
                    rocprof_names[k] = alt
        ctr = d0.get(rocprof_names[dk])
        if ctr:      # a frame of several passes launches the kernel once per pass: counters per LAUNCH
            nl = max(ctr.get("launches", 1.0), 1.0)
            ctr = {k: v / nl for k, v in ctr.items()}
        ctr_src = {"file": os.path.relpath(ctr_files[-1], ROOT), "commit": cj.get("commit"), "kernels_hash": cj.get("kernels_hash"),
                   "kernels_hash_now": kernels_hash(), "kernel": rocprof_names[dk],
                   "note": "counters come from this committed file (tools/collect_counters.sh), durations, work counts and the "
                           "clock are measured live; stale = the kernels or their build flags changed since it was collected"}
    fresh = ctr if (ctr and not stale) else None
    traffic = fresh["hbm_bytes"] if fresh else None
    roof = {"kernel": rocprof_names[dk], "bound": "valu_fp32",
            "achieved": alg_tfl, "peak": peak_tfl, "unit": "TFLOP/s", "frac": alg_tfl / peak_tfl, "traffic": traffic,
            "definition": "achieved = SURVEY.md 8(d) flops/ray x the per-ray census of ONE launch / that launch's average duration "
                          "(HIP events on the scene's stream); peak = 1024 SIMDs x 32 lanes x 2 flop x the shader clock measured in "
                          "the launch (157.3 TFLOP/s at the 2400 MHz of MI355X_MICROARCH.md; no MFMA on this path: nothing is a "
                          "dense contraction); traffic = HBM bytes per launch from the PMC counters (2 x FETCH_SIZE + WRITE_SIZE)",
            "shader_clock_mhz": clock_mhz, "peak_at_2400_mhz": VALU_PEAK_TFLOPS, "frac_of_peak_at_2400_mhz": alg_tfl / VALU_PEAK_TFLOPS,
            "avg_launch_ms": avg_ms, "launches_per_frame_at_depth0": launches,
            "units_per_launch": dict({k: v / launches for k, v in units.items()}, rays=rays_k / launches,
                                     note="per RAY: each ray traversed by itself (RT_SCENE_NO_PACKETS census of the same frame), "
                                          "divided by the frame's launches of this kernel (one per pass)"),
            "packet_steps_per_launch": {"wave_node_steps": wp[dk].wave_node_steps / launches, "wave_tri_steps": wp[dk].wave_tri_steps / launches,
                                        "lane_node_visits": wp[dk].nodes_visited / launches, "lane_tri_tests": wp[dk].tris_tested / launches,
                                        "note": "what the launch executes: one scalar fetch and one step per WAVE for its 64 rays"},
            "algorithmic": {"flops_per_launch": f_alg, "TFLOP/s": alg_tfl, "bytes_per_launch": b_alg, "GB/s": alg_gbs,
                            "note": "SURVEY.md 8(d) formulas x per-ray counts / launches / launch duration. The bytes are served by "
                                    "L2 and the scalar cache (nodes, leaves, object table: see traffic), so their rate is not an HBM "
                                    "rate and is not priced against the HBM peak"},
            "level0_launch_ms": {rocprof_names[k]: ms0[k] / n0[k] for k in range(NK) if n0[k]},
            "serial_frame_ms": frame_ms / reps, "kernels": per_kernel, "counters_from": ctr_src, "counters_stale": stale}
    if fresh:
        clk = (clock_mhz or CLOCK_GHZ * 1e3) * 1e6
        ginst = fresh["SQ_INSTS_VALU"] / (avg_ms * 1e-3) / 1e9
        wc = max(fresh.get("SQ_WAVE_CYCLES", 1), 1)
        rate = traffic / (avg_ms * 1e-3) / 1e9
        roof["valu_issue"] = {
            "valu_instructions_per_launch": fresh["SQ_INSTS_VALU"], "salu_instructions_per_launch": fresh.get("SQ_INSTS_SALU"),
            "G_wave64_instructions_per_s": ginst, "peak_G_per_s": SIMDS * clk / 2.0 / 1e9, "utilisation": ginst / (SIMDS * clk / 2.0 / 1e9),
            "lane_ops_per_algorithmic_flop": fresh["SQ_INSTS_VALU"] * 64.0 / f_alg,
            "wait_inst_share_of_wave_cycles": fresh.get("SQ_WAIT_INST_ANY", 0) / wc,
            "wait_any_share_of_wave_cycles": fresh.get("SQ_WAIT_ANY", 0) / wc,
            "l2_hit_rate": fresh.get("TCC_HIT_sum", 0) / max(fresh.get("TCC_HIT_sum", 0) + fresh.get("TCC_MISS_sum", 0), 1),
            "note": "EXECUTED wave64 VALU instructions / duration against one per 2 cycles per SIMD at the measured clock: how busy "
                    "the issue slots are, overhead included - a diagnostic, not the roofline fraction"}
        roof["hbm_measured"] = {"bytes_per_launch": traffic, "GB/s": rate, "of_hbm_peak": rate / HBM_PEAK_GBS,
                                "x_algorithmic_bytes": traffic / b_alg}
    # ---- the same launch in the BATCHED configuration `value` is measured in: a timed call (HIP events around every launch,
    # one stream) of B frames at max_depth 0 = the level-0 launches of a batch. Its rays are the cameras' census rays. ----
    if len(cams) > 1 and cam_rays:
        sb = rt_stats()
        dev.render_batch_device(cams, p0, slots[0].frames.data_ptr(), frame_bytes, stream=stream, stats=sb)
        dev.render_batch_device(cams, p0, slots[0].frames.data_ptr(), frame_bytes, stream=stream, stats=sb)
        if sb.kernel_launches_by[dk]:
            ms_b = sb.kernel_ms_by[dk] / sb.kernel_launches_by[dk]
            scale = (sum(cam_rays) / cam_rays[0]) / sb.kernel_launches_by[dk]      # rays of the batch per launch, in frames of camera 0
            tfl_b = f_frame * scale / (ms_b * 1e-3) / 1e12
            roof["batched_launch"] = {"frames_per_call": len(cams), "launches": sb.kernel_launches_by[dk], "avg_launch_ms": ms_b,
                                      "ms_per_frame": sb.kernel_ms_by[dk] / len(cams), "TFLOP/s": tfl_b, "frac": tfl_b / peak_tfl,
                                      "note": "the dominant kernel's launch when a call carries a batch of frames (what `value` times): "
                                              "algorithmic flops of camera 0's census x the batch's rays / camera 0's rays"}
        # ... and every kernel kind of a whole batched call (full depth), launches back to back on one stream
        sf = rt_stats()
        dev.render_batch_device(cams, params, slots[0].frames.data_ptr(), frame_bytes, stream=stream, stats=sf)
        dev.render_batch_device(cams, params, slots[0].frames.data_ptr(), frame_bytes, stream=stream, stats=sf)
        roof["batched_call_serial"] = {"frames_per_call": len(cams), "ms_per_frame": sf.kernel_ms / len(cams),
                                       "ms_per_frame_by_kernel": {names[k]: sf.kernel_ms_by[k] / len(cams) for k in range(NK) if sf.kernel_launches_by[k]},
                                       "launches_by_kernel": {names[k]: sf.kernel_launches_by[k] for k in range(NK) if sf.kernel_launches_by[k]},
                                       "note": "one call of B frames with HIP events around every launch (one stream, nothing overlaps); "
                                               "k_extend_quad / k_shade_quad rows include the quad-service launches of the bounce levels; "
                                               "the remainder of ms_per_frame is folds, resolve and frame set-up"}
        dev.render_batch_device(cams[:1], params, slots[0].frames.data_ptr(), frame_bytes, stream=stream)
        dev.status()
    out["roofline"] = roof
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(fs, cam, params, dev, np)
        out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]


if __name__ == "__main__":
    main()
