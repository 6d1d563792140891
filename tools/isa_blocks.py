"""Basic blocks of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only), with the instruction classes of each:
    python tools/isa_blocks.py rt_kernels.s 'k_shade_wide<false, false, true>'
prints label, VALU / SALU / SMEM / VMEM / LDS counts, and where the block branches to (to find the loops)."""
import re, subprocess, sys
path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
syms = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
names = subprocess.run(["c++filt"], input="\n".join(s for _, s in syms), capture_output=True, text=True).stdout.split("\n")
start = end = None
for k, ((i, s), n) in enumerate(zip(syms, names)):
    if want in n:
        start, end = i, (syms[k + 1][0] if k + 1 < len(syms) else len(lines)); break
assert start is not None, "kernel not found"
blocks, cur = [], {"label": "entry", "c": {}, "br": [], "line": start}
def cls(op):
    if op.startswith("v_"): return "VALU"
    if op.startswith(("s_load", "s_buffer_load")): return "SMEM"
    if op.startswith(("s_waitcnt", "s_nop", "s_branch", "s_cbranch", "s_endpgm", "s_barrier", "s_setprio", "s_sleep")): return "ctl"
    if op.startswith("s_"): return "SALU"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    if op.startswith("ds_"): return "LDS"
    return "other"
for i in range(start + 1, end):
    l = lines[i].split(";")[0].strip()
    if not l or l.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur); cur = {"label": l[:-1], "c": {}, "br": [], "line": i}
        continue
    op = l.split()[0]
    if op == "s_endpgm": cur["c"]["ctl"] = cur["c"].get("ctl", 0) + 1; break
    k = cls(op); cur["c"][k] = cur["c"].get(k, 0) + 1
    if op.startswith(("s_branch", "s_cbranch")): cur["br"].append(l.split()[-1])
    if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"): cur["c"]["lane"] = cur["c"].get("lane", 0) + 1
    if "_f64" in op: cur["c"]["f64"] = cur["c"].get("f64", 0) + 1
blocks.append(cur)
order = {b["label"]: k for k, b in enumerate(blocks)}
tot = {}
for k, b in enumerate(blocks):
    back = [t for t in b["br"] if order.get(t, 1 << 30) <= k]
    for x, v in b["c"].items(): tot[x] = tot.get(x, 0) + v
    print(f'{b["line"]+1:7d} {b["label"]:12s} VALU {b["c"].get("VALU",0):4d} (lane {b["c"].get("lane",0):3d} f64 {b["c"].get("f64",0):3d}) SALU {b["c"].get("SALU",0):4d} SMEM {b["c"].get("SMEM",0):3d} VMEM {b["c"].get("VMEM",0):3d} LDS {b["c"].get("LDS",0):3d}  -> {" ".join(b["br"])}{"   <== BACK " + " ".join(back) if back else ""}')
print("total", tot)
