#!/bin/bash
# expensive groups first: the new parity test, the stress check, and the bench of the workload whose batches are ordered too
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread -k "expensive_groups or frame_parity or schedule" 2>&1 | tail -3 || exit 1
ORDER_CHECK_STREAM=own timeout -k 10 250 python3 tools/exp_order_check.py c1_scene_yaml_800x600 32 40 2>&1 | tail -2
ORDER_CHECK_STREAM=own timeout -k 10 250 python3 tools/exp_order_check.py bunny_1080p 1 60 2>&1 | tail -2
for rep in 1 2 3 4 5 6; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --workload c1_scene_yaml_800x600 --steps 64 --warmup 32 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c1 batched', round(d['ms_per_step'],4), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done
