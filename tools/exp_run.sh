#!/bin/bash
# Run length of the workgroup -> group mapping (group_of) against frame time.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for r in ${RUNS:-31 2 4 6 8}; do
  for w in ${WORKLOADS:-bunny_1080p c5_field_10m_1080p_4spp c4_bunny_glass_4k_16spp}; do
    RT_DEBUG_SKIP=$((r << 8)) timeout -k 10 300 python -u bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 --watchdog 280 > gpurun_out/run_${r}_$w.json 2> gpurun_out/run_${r}_$w.err
    python - $r $w <<'PY'
import json,sys; d=json.loads(open('gpurun_out/run_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
l0=d['roofline']['level0_launch_ms']
print('run_log2', sys.argv[1], sys.argv[2], 'ms/frame %.4f'%d['ms_per_step'], 'one at a time %.4f'%d['one_frame_at_a_time']['ms_per_frame'], 'serial %.3f'%d['roofline']['serial_frame_ms'], {k.split('<')[0]+k[k.index('<'):k.index('<')+2]: round(v,3) for k,v in l0.items()}, d['frame_identical_to_1gpu'], flush=True)
PY
  done
done
