"""The N > 1 path on CPU: two gloo ranks each "render" the tiles they own (the oracle stands in for the device,
writing the same tile-major layout rt_render_tiles_device produces), rank 0 gathers with torch.distributed and
scatters the tiles back; the frame must be byte-identical to the single-rank frame."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import gold, sub
from raytracer_amd import cabi, tiling
from raytracer_amd.scene import FlatScene, Params, camera_from_array


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tw, th, result_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    g = gold("frames.npz")
    fs = FlatScene.from_arrays(sub(g, "mesh_scene_"))
    cam = camera_from_array(g["cam_small"])
    W, H = cam.imWidth, cam.imHeight
    orc = O.Oracle(fs)
    slot = tiling.tiles_owned(W, H, tw, th, 0, world)
    mine = np.zeros((slot, th, tw, 3), np.uint8)
    scratch = np.zeros((H, W, 3), np.uint8)
    for j, x0, y0, w, h in tiling.owned_tile_rects(W, H, tw, th, rank, world):
        orc.render(cam, Params(), rect=(x0, y0, w, h), out=scratch)
        mine[j, :h, :w] = scratch[y0:y0 + h, x0:x0 + w]
    t = torch.from_numpy(mine.reshape(-1))
    gathered = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gathered, dst=0)
    if rank == 0:
        frame = tiling.untile([x.numpy() for x in gathered], W, H, tw, th)
        np.save(result_path, frame)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_gather_is_byte_identical(tmp_path, oracle_mod):
    tw, th = 24, 16           # 160x120 -> 7 x 8 tiles, ragged right edge (160 = 6*24 + 16)
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), tw, th, out), nprocs=2, join=True)
    frame = np.load(out)
    assert np.array_equal(frame, gold("frames.npz")["mesh_d10_1"])


def test_tiles_owned_matches_the_library():
    lib = cabi.load()
    for (W, H, tw, th) in [(1920, 1080, 32, 32), (160, 120, 24, 16), (7, 5, 8, 8), (800, 600, 64, 16)]:
        for n in (1, 2, 3, 4, 8):
            total = 0
            for p in range(n):
                k = lib.rt_tiles_owned(W, H, tw, th, p, n)
                assert k == tiling.tiles_owned(W, H, tw, th, p, n)
                total += k
            tx, ty = tiling.tile_grid(W, H, tw, th)
            assert total == tx * ty
    assert lib.rt_tiles_owned(10, 10, 8, 8, 3, 2) == 0       # part >= n_parts
    assert lib.rt_tiles_owned(10, 10, 0, 8, 0, 1) == 0


def _run_bench(*argv, env_extra=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_n_starts_n_ranks():
    """`python bench.py --gpus N` (the driver's form, no launcher) must be N processes: the launcher path, rehearsed
    without a GPU (--rehearse-launcher: rendezvous over gloo, all-gather of the ranks, rank 0 reports)."""
    r, d = _run_bench("--gpus", "3", "--rehearse-launcher")
    assert r.returncode == 0, r.stderr[-2000:]
    assert d["n_gpus"] == 3 and d["rccl_ranks"] == 3 and d["ranks_seen"] == [0, 1, 2] and d["distinct_processes"] == 3
    # a launcher that disagrees with --gpus is an error, not a silent 1-rank run
    r, d = _run_bench("--gpus", "4", "--rehearse-launcher", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and d is None


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """The whole N-rank script with two ranks sharing GPU 0 (gloo carries the gather: RCCL needs a GPU per rank):
    tiles dealt to two processes, gathered, untiled; the frames must equal the single-GPU census frames."""
    r, d = _run_bench("--gpus", "2", "--backend", "gloo", "--single-device", "--steps", "8", "--warmup", "4",
                      "--workload", "c1_scene_yaml_800x600", "--frames-per-batch", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["frame_identical_to_1gpu"] is True
    assert d["value"] > 0 and d["config"]["parallelism"].startswith("2 GPUs")
    # where an N-rank frame's time goes: per-phase times (max / mean over the ranks) and the tiles each rank owns
    ph = d["phases"]
    assert ph["render_ms"]["max"] >= ph["render_ms"]["mean"] > 0 and ph["gather_ms"]["max"] > 0 and ph["untile_ms"]["rank0"] > 0
    assert len(ph["tiles_per_rank"]) == 2 and sum(ph["tiles_per_rank"]) == 25 * 19      # 800x600 in 32x32 tiles
