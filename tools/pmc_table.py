import csv, glob, re, collections
def short(n):
    m = re.match(r'(?:void )?(k_\w+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:24]
tab = collections.OrderedDict()
for g in 'abcd':
    f = glob.glob(f'gpurun_out/pmc_{g}/**/p_counter_collection.csv', recursive=True)
    if not f: continue
    rows = list(csv.DictReader(open(f[0])))
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    # last frame = dispatches after the second-to-last k_resolve
    name = {int(r['Dispatch_Id']): short(r['Kernel_Name']) for r in rows}
    res = [i for i in ids if name[i].startswith('k_resolve')]
    lo = res[-2] + 1 if len(res) >= 2 else ids[0]
    for r in rows:
        i = int(r['Dispatch_Id'])
        if i < lo: continue
        key = (i - lo, name[i])
        tab.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
cols = []
for v in tab.values():
    for c in v:
        if c not in cols: cols.append(c)
print('idx kernel ' + ' '.join(cols))
for (i, n), v in sorted(tab.items()):
    if 'fold' in n: continue
    print(i, n, ' '.join(f"{v.get(c, float('nan')):.4g}" for c in cols))
