"""Timeline of the LAST frame in a rocprofv3 kernel trace of tools/one_frame.py (ONE_FRAME_SYNC=1): start, duration, end of
every launch relative to the frame's first launch. Usage: lat_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.match(r'(?:void )?(k_\w+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:30]
res = [i for i, r in enumerate(rows) if 'k_resolve' in r['Kernel_Name']]
lo, hi = res[-2] + 1, res[-1]
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{short(r['Kernel_Name']):34s} grid {int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0)):9d} start {(s - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}  end {(e - t0) / 1e3:8.1f}")
