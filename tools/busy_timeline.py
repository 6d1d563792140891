"""Where a kernel trace's wall time goes: for the window between the starts of the 3rd and the last k_resolve of at least
MIN_GRID threads, the time during which exactly the named kinds of kernels were running. busy_timeline.py <trace.csv> [min_grid]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 10 * 2073600
def grid(r): return int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0))
def short(n):
    m = re.match(r'(?:void )?(k_\w+)', n)
    return m.group(1) if m else 'other'
big = sorted(int(r['Start_Timestamp']) for r in rows if 'k_resolve' in r['Kernel_Name'] and grid(r) >= min_grid)
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t0, t1 = big[skip], big[-1]
ev = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    s, e = max(s, t0), min(e, t1)
    if e > s:
        k = short(r['Kernel_Name'])
        ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
active = defaultdict(int)
last = t0
by_set = defaultdict(int)
for t, d, k in ev:
    if t > last:
        key = '+'.join(sorted(x for x, c in active.items() if c > 0)) or 'idle'
        by_set[key] += t - last
        last = t
    active[k] += d
total = t1 - t0
print(f"window {total / 1e3:.1f} us, calls {len(big) - 1 - skip}")
for key, v in sorted(by_set.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{100 * v / total:5.1f} %  {v / 1e3:9.1f} us  {key}")
