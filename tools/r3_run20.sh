#!/bin/bash
# RT_RUN_HITS: the driver's short run (one call of 20 frames) and the default run, per value.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for v in ${VALS:-0 1000000}; do
  for st in "20 5" "128 32"; do set -- $st
  RT_RUN_HITS=$v timeout -k 10 200 python bench.py --no-cpu-baseline --gpus 1 --workload ${WL:-bunny_1080p} --steps $1 --warmup $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RT_RUN_HITS=$v steps $1', 'ms/step', round(d['ms_per_step'],4), round(d['value']), d['frame_identical_to_1gpu'], flush=True)" || exit 1
  done
done; done
