#!/bin/bash
# Instruction counts of the level-0 launches for several library builds (tools/exp_ab_builds.sh build ...), one rocprofv3
# --pmc pass per library: LIBS="a b" WORKLOADS="bunny_1080p" bash tools/ab_counters.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in ${WORKLOADS:-bunny_1080p}; do for l in $LIBS; do
  lib=$PWD/tools/bin/librt_amd_$l.so; [ "$l" = tree ] && lib=$PWD/raytracer_amd/csrc/librt_amd.so
  rm -rf gpurun_out/abc_$l
  RT_AMD_LIB=$lib timeout -s KILL ${LIM:-200} rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
     --output-format csv -d gpurun_out/abc_$l -o p -- python3 tools/frames_for_counters.py $w > gpurun_out/abc_$l.log 2>&1 || { echo "pass $l failed"; tail -5 gpurun_out/abc_$l.log; exit 1; }
  python3 - $w $l <<'PY'
import csv, glob, sys, collections, re
w, l = sys.argv[1:3]
f = glob.glob(f'gpurun_out/abc_{l}/**/p_counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    m = re.match(r'(?:void )?(k_\w+)(<.*>)?', n.split('(')[0].strip())
    return (m.group(1) + (m.group(2) or '')) if m else n[:32]
ids = sorted({int(r['Dispatch_Id']) for r in rows}); name = {int(r['Dispatch_Id']): short(r['Kernel_Name']) for r in rows}
frame, fr = -1, {}
for i in ids:
    if name[i].startswith('k_set_frame_params'): frame += 1
    fr[i] = frame
res = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    i = int(r['Dispatch_Id'])
    if fr[i] == 1 and name[i].startswith(('k_extend', 'k_shade')): res[name[i]][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in res.items():
    print(w, l, k, 'VALU %.2fM SALU %.2fM SMEM %.2fM wave_cyc %.1fM busy %.2fM wait_inst %.2f active_valu %.2f' % (
        c['SQ_INSTS_VALU'] / 1e6, c['SQ_INSTS_SALU'] / 1e6, c['SQ_INSTS_SMEM'] / 1e6, c['SQ_WAVE_CYCLES'] / 1e6, c['SQ_BUSY_CYCLES'] / 1e6,
        c['SQ_WAIT_INST_ANY'] / max(c['SQ_WAVE_CYCLES'], 1), c['SQ_ACTIVE_INST_VALU'] / max(c['SQ_WAVE_CYCLES'], 1)), flush=True)
PY
  rm -rf gpurun_out/abc_$l
done; done
