#!/bin/bash
# Counters of the bounce-level kernels (quad service / lane traversal) of a steady-state frame for several library builds:
# two rocprofv3 --pmc passes per library, each a run of its own (SQ; TA + TCP).
#   LIBS="a b" WORKLOADS="c4_bunny_glass_4k_16spp" bash tools/ab_counters2.sh      (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in ${WORKLOADS:-c4_bunny_glass_4k_16spp}; do for l in $LIBS; do
  lib=$PWD/tools/bin/librt_amd_$l.so; [ "$l" = tree ] && lib=$PWD/raytracer_amd/csrc/librt_amd.so
  n=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
              "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
    n=$((n+1)); rm -rf gpurun_out/abc_${l}_$n
    RT_AMD_LIB=$lib timeout -s KILL ${LIM:-200} rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/abc_${l}_$n -o p -- python3 tools/frames_for_counters.py $w > gpurun_out/abc_${l}_$n.log 2>&1 || { echo "pass $l $n failed"; tail -5 gpurun_out/abc_${l}_$n.log; exit 1; }
  done
  python3 - $w $l <<'PY'
import csv, glob, sys, collections, re
w, l = sys.argv[1:3]
res = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for n in (1, 2):
    f = glob.glob(f'gpurun_out/abc_{l}_{n}/**/p_counter_collection.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    def short(nm):
        m = re.match(r'(?:void )?(k_\w+)(<.*>)?', nm.split('(')[0].strip())
        return (m.group(1) + (m.group(2) or '')) if m else nm[:32]
    ids = sorted({int(r['Dispatch_Id']) for r in rows}); name = {int(r['Dispatch_Id']): short(r['Kernel_Name']) for r in rows}
    frame, fr = -1, {}
    for i in ids:
        if name[i].startswith('k_set_frame_params'): frame += 1
        fr[i] = frame
    seen = set()
    for r in rows:
        i = int(r['Dispatch_Id'])
        if fr[i] == 3 and name[i].startswith(('k_extend', 'k_shade')):
            res[name[i]][r['Counter_Name']] += float(r['Counter_Value'])
            if n == 1 and i not in seen: seen.add(i); cnt[name[i]] += 1
for k, c in sorted(res.items()):
    print(w, l, k, 'x%d' % cnt[k], 'VALU %.1fM SALU %.1fM VMEM_RD %.1fM wave_cyc %.0fM wait_inst %.2f active_valu %.2f | TA_BUSY_avr %.1fM of GUI_ACTIVE %.1fM, TA rd wavefronts %.1fM, TCP accesses %.1fM, TA-data stall %.0fM, pending stall %.0fM, TCC rd req %.1fM' % (
        c['SQ_INSTS_VALU'] / 1e6, c['SQ_INSTS_SALU'] / 1e6, c['SQ_INSTS_VMEM_RD'] / 1e6, c['SQ_WAVE_CYCLES'] / 1e6,
        c['SQ_WAIT_INST_ANY'] / max(c['SQ_WAVE_CYCLES'], 1), c['SQ_ACTIVE_INST_VALU'] / max(c['SQ_WAVE_CYCLES'], 1),
        c['TA_BUSY_avr'] / 1e6, c['GRBM_GUI_ACTIVE'] / 1e6, c['TA_FLAT_READ_WAVEFRONTS_sum'] / 1e6, c['TCP_TOTAL_CACHE_ACCESSES_sum'] / 1e6,
        c['TCP_TCP_TA_DATA_STALL_CYCLES_sum'] / 1e6, c['TCP_PENDING_STALL_CYCLES_sum'] / 1e6, c['TCP_TCC_READ_REQ_sum'] / 1e6), flush=True)
PY
  rm -rf gpurun_out/abc_${l}_1 gpurun_out/abc_${l}_2
done; done
