#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for q in ${QS:-4 8 16}; do for f in ${FS:-2 3}; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 60 --warmup 6 --frames-in-flight $f --watchdog 280 > gpurun_out/hq_${q}_$f.json 2> gpurun_out/hq_${q}_$f.err
  python - $q $f <<'PY'
import json,sys; d=json.loads(open('gpurun_out/hq_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
print('hw queues', sys.argv[1], 'frames in flight', sys.argv[2], 'ms/frame %.4f'%d['ms_per_step'], 'one at a time %.4f'%d['one_frame_at_a_time']['ms_per_frame'] if 'one_frame_at_a_time' in d else '', flush=True)
PY
done; done
