#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for f in ${FIF:-1 2 3 4}; do
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 64 --warmup 16 --frames-in-flight $f --watchdog 280 > gpurun_out/fif_$f.json 2> gpurun_out/fif_$f.err
  python - $f <<'PY'
import json,sys; d=json.loads(open('gpurun_out/fif_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('frames in flight', sys.argv[1], 'ms/frame %.4f'%d['ms_per_step'], d['frame_identical_to_1gpu'], flush=True)
PY
done
