#!/bin/bash
# The driver's short run against the same 20 timed frames behind a warm-up of the same call shape.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do for wu in 5 20 40; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --gpus 1 --steps 20 --warmup $wu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup $wu', 'ms/step', round(d['ms_per_step'],4), round(d['value']), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done
