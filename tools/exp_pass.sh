#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for k in ${KS:-21 22 23 24}; do
  for w in ${WORKLOADS:-c4_bunny_glass_4k_16spp c5_field_10m_1080p_4spp}; do
  RT_PASS_LOG2=$k timeout -k 10 300 python -u bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 --watchdog 280 > gpurun_out/pass_${k}_$w.json 2> gpurun_out/pass_${k}_$w.err
  python - $k $w <<'PY'
import json,sys; d=json.loads(open('gpurun_out/pass_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
print('pass log2', sys.argv[1], sys.argv[2], 'ms/frame %.4f'%d['ms_per_step'], 'one at a time %.4f'%d['one_frame_at_a_time']['ms_per_frame'], d['frame_identical_to_1gpu'], flush=True)
PY
  done
done
