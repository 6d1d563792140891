import csv, json, re, sys
rows = list(csv.DictReader(open('gpurun_out/prof_lv/b_kernel_trace.csv')))
d = json.load(open('gpurun_out/lv.json'))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.match(r'(?:void )?(k_\w+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:30]
# frames of the untimed loop: find k_resolve launches; the frame = kernels between two resolves. Take the frame
# before the roofline/stat frames: index steps+warmup-1 after the census frame (frame 0 is the census).
res = [i for i, r in enumerate(rows) if 'k_resolve' in r['Kernel_Name']]
k = d['steps'] + d['warmup']          # census frame is res[0]; timed frames follow
lo, hi = res[k - 1] + 1, res[k]
t0 = int(rows[lo]['Start_Timestamp'])
print('ms/frame', round(d['ms_per_frame'], 4), 'rays', d['config']['level_rays'])
for r in rows[lo:hi + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{short(r['Kernel_Name']):28s} start {(s - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}  end {(e - t0) / 1e3:8.1f}")
