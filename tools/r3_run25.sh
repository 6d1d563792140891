#!/bin/bash
# large frames: one call at a time against two scenes in flight
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout -k 10 500 python bench.py "$@" --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['ms_per_step'],4), round(d['value']), d['frame_identical_to_1gpu'], flush=True)" || exit 1; }
for rep in 1 2; do
run --workload c4_bunny_glass_4k_16spp --steps 16 --warmup 4 --frames-in-flight 1
run --workload c4_bunny_glass_4k_16spp --steps 16 --warmup 4 --frames-in-flight 2
done
run --workload c5_field_10m_8k_64spp --steps 2 --warmup 2 --frames-in-flight 1
run --workload c5_field_10m_8k_64spp --steps 4 --warmup 2 --frames-in-flight 2
