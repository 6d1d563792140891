"""CSV output of tools/collect_counters.sh -> gpurun_out/counters_<workload>.json.

Per kernel name (with template arguments, as rocprofv3 prints it), for the frame at max_depth 0 (`depth0`: one launch
per kernel = the level-0 launches, the heaviest of a frame) and for one full frame (`frame`): launches, summed
counters, HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; the x2 is the guide's gfx950 correction: 128-B read
requests are tallied at 64 B), summed kernel-trace durations. SQ counters are sums over the chip.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

W = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p"


def short(n):
    m = re.match(r'(?:void )?(k_\w+)(<.*>)?', n.split('(')[0].strip())
    return (m.group(1) + (m.group(2) or '')) if m else n[:32]


def frames_of(rows, id_key, name_key):
    """dispatch id -> frame ordinal (0-based), frames start at k_set_frame_params."""
    ids = sorted({int(r[id_key]) for r in rows})
    name = {int(r[id_key]): short(r[name_key]) for r in rows}
    frame, out = -1, {}
    for i in ids:
        if name[i].startswith('k_set_frame_params'):
            frame += 1
        out[i] = frame
    return out, name


res = {"depth0": collections.defaultdict(lambda: collections.defaultdict(float)),
       "frame": collections.defaultdict(lambda: collections.defaultdict(float))}
WHICH = {2: "depth0", 4: "frame"}      # (frames_for_counters.py: full, full, depth 0, full, full)
for p in ("sq1", "sq2", "fetch", "write", "tcc"):
    f = glob.glob(f'gpurun_out/ctr_{p}/**/p_counter_collection.csv', recursive=True)
    if not f:
        print("missing pass", p)
        continue
    rows = list(csv.DictReader(open(f[0])))
    fr, name = frames_of(rows, 'Dispatch_Id', 'Kernel_Name')
    seen = set()
    for r in rows:
        i = int(r['Dispatch_Id'])
        w = WHICH.get(fr[i])
        if not w or not name[i].startswith('k_'):
            continue
        k = res[w][name[i]]
        k[r['Counter_Name']] += float(r['Counter_Value'])
        if p == "sq1" and (i, w) not in seen:
            seen.add((i, w))
            k["launches"] += 1
f = glob.glob('gpurun_out/ctr_trace/**/p_kernel_trace.csv', recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    frame = -1
    for r in rows:
        n = short(r['Kernel_Name'])
        if n.startswith('k_set_frame_params'):
            frame += 1
        w = WHICH.get(frame)
        if w and n.startswith('k_'):
            res[w][n]["trace_ns"] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            res[w][n]["trace_launches"] += 1
for w in res:
    for k, v in res[w].items():
        v["hbm_bytes"] = 2.0 * 1024.0 * v.get("FETCH_SIZE", 0.0) + 1024.0 * v.get("WRITE_SIZE", 0.0)
stamp = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".commit_stamp")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd.buildinfo import kernels_hash  # noqa: E402
out = {"workload": W, "commit": open(stamp).read().strip() if os.path.exists(stamp) else "unknown",
       "kernels_hash": kernels_hash(),      # of the kernel sources + build flags these counters were collected from
       "collected_by": "tools/collect_counters.sh (rocprofv3 --pmc, one pass per counter group; --kernel-trace for trace_ns)",
       "note": "depth0 = the frame at max_depth 0 (one launch per kernel: per-launch numbers); frame = one full frame; "
               "hbm_bytes = 2*FETCH_SIZE + WRITE_SIZE in bytes",
       "depth0": res["depth0"], "frame": res["frame"]}
json.dump(out, open(f'gpurun_out/counters_{W}.json', 'w'), indent=1, sort_keys=True)
for w in ("depth0", "frame"):
    print(f"== {W} {w}")
    for k, v in sorted(res[w].items(), key=lambda kv: -kv[1].get("trace_ns", 0)):
        ln = max(v.get("launches", 1), 1)
        print(f"{k:42s} x{int(ln):4d} {v.get('trace_ns', 0) / 1e3:9.1f} us  VALU {v.get('SQ_INSTS_VALU', 0) / 1e6:8.2f}M SALU {v.get('SQ_INSTS_SALU', 0) / 1e6:8.2f}M "
              f"SMEM {v.get('SQ_INSTS_SMEM', 0) / 1e6:7.2f}M wait_inst/wave_cyc {v.get('SQ_WAIT_INST_ANY', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1):.2f} "
              f"wait_any/wave_cyc {v.get('SQ_WAIT_ANY', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1):.2f} HBM {v['hbm_bytes'] / 1e6:9.1f} MB "
              f"L2 hit {v.get('TCC_HIT_sum', 0) / max(v.get('TCC_HIT_sum', 0) + v.get('TCC_MISS_sum', 0), 1):.3f}")
