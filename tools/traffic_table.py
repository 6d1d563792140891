"""Per-kernel HBM bytes of the last frame of the tools/exp_traffic.sh passes -> gpurun_out/traffic_<workload>.json.
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B, guide's HBM section)."""
import csv, glob, json, re, sys, collections
wl = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p"
def short(n):
    m = re.match(r'(?:void )?(k_\w+)(<[^>]*>)?', n)     # kernel with its template arguments, as rocprofv3 names it
    return (m.group(1) + (m.group(2) or '')) if m else n[:24]
out = collections.defaultdict(lambda: {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f'gpurun_out/pmc_{c}/**/p_counter_collection.csv', recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r['Counter_Name'] == c]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    name = {int(r['Dispatch_Id']): short(r['Kernel_Name']) for r in rows}
    res = [i for i in ids if name[i].startswith('k_resolve')]
    lo = res[-2] + 1
    seen = set()
    for r in rows:
        i = int(r['Dispatch_Id'])
        if i < lo or not name[i].startswith('k_'): continue
        v = float(r['Counter_Value']) * 1024.0
        k = out[name[i]]
        if c == "FETCH_SIZE":
            k["fetch_bytes"] += 2.0 * v
            if i not in seen: k["launches"] += 1; seen.add(i)
        else:
            k["write_bytes"] += v
for k, v in out.items():
    v["hbm_bytes_per_launch"] = (v["fetch_bytes"] + v["write_bytes"]) / max(v["launches"], 1)
    print(k, v)
json.dump({"workload": wl, "kernels": out, "note": "one frame; FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes"},
          open(f'gpurun_out/traffic_{wl}.json', 'w'), indent=1)
