#!/bin/bash
# shade streams per scene x frames in flight
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for s in 1 2 3; do for f in 1 2 3 4; do
  RT_SHADE_STREAMS=$s timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 60 --warmup 6 --frames-in-flight $f --watchdog 280 > gpurun_out/st_${s}_$f.json 2> gpurun_out/st_${s}_$f.err
  python - $s $f <<'PY'
import json,sys; d=json.loads(open('gpurun_out/st_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
print('shade streams', sys.argv[1], 'frames in flight', sys.argv[2], 'ms/frame %.4f'%d['ms_per_step'], d['frame_identical_to_1gpu'], flush=True)
PY
done; done
