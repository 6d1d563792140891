#!/bin/bash
# small batches of whole frames: bounce levels by quads or by quad service (where the pass-size rule should switch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout -k 10 300 python bench.py --warmup 24 "$@" --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$RT_SERVE', '$*', '->', round(d['ms_per_step'],4), flush=True)" || exit 1; }
for b in 3 4 6; do for v in 0 1; do export RT_SERVE=$v
run --frames-per-batch $b --steps 96
done; done
for v in 0 1; do export RT_SERVE=$v
run --workload c1_scene_yaml_800x600 --frames-per-batch 12 --steps 96
done
