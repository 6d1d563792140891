#!/bin/bash
# threads per workgroup of the per-lane kernels: the driver's short run and the default run, repeated
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do for v in 64 128; do
  for st in "20 5" "128 32"; do set -- $st
  RT_WIDE_BLOCK=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --gpus 1 --steps $1 --warmup $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RT_WIDE_BLOCK=$v steps $1', 'ms/step', round(d['ms_per_step'],4), flush=True)" || exit 1
  done
done; done
