# SQ counters of the last frame's launches (tools/one_frame.py), per kernel: where do the waves' cycles go?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-c3_bunny_1080p_shadows}
rm -rf gpurun_out/pmc_sq
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_sq -o p -- python3 tools/one_frame.py $W 3 > gpurun_out/pmc_sq.log 2>&1 || { tail -5 gpurun_out/pmc_sq.log; exit 1; }
rm -rf gpurun_out/pmc_sq2
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d gpurun_out/pmc_sq2 -o p -- python3 tools/one_frame.py $W 3 > gpurun_out/pmc_sq2.log 2>&1 || { tail -5 gpurun_out/pmc_sq2.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, re
for d in ('pmc_sq','pmc_sq2'):
    f = glob.glob(f'gpurun_out/{d}/**/p_counter_collection.csv', recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    name = {int(r['Dispatch_Id']): r['Kernel_Name'] for r in rows}
    res = [i for i in ids if name[i].startswith('k_resolve')]
    lo = res[-2] + 1
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        i = int(r['Dispatch_Id'])
        if i < lo: continue
        m = re.match(r'(?:void )?(k_\w+)(<[^>]*>)?', name[i])
        acc[(m.group(1) + (m.group(2) or '')) if m else name[i][:24]][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in acc.items():
        print(k, {c: int(x) for c, x in v.items()})
PY
