"""Timeline of the LAST batched call in a rocprofv3 kernel trace of bench.py: every launch between the k_resolve of at
least MIN_GRID threads that ends that call and the k_resolve before it. Usage: batch_timeline.py <kernel_trace.csv> [min_grid]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * 2073600
def grid(r):
    return int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0))
def short(n):
    m = re.match(r'(?:void )?(k_\w+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:30]
big = [i for i, r in enumerate(rows) if 'k_resolve' in r['Kernel_Name'] and grid(r) >= min_grid]
hi = big[-1]
lo = max(i for i, r in enumerate(rows[:hi]) if 'k_resolve' in r['Kernel_Name']) + 1
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{short(r['Kernel_Name']):40s} grid {grid(r):10d} start {(s - t0) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}  end {(e - t0) / 1e3:8.1f}")
