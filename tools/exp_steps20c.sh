#!/bin/bash
# the driver's short run again, after run_hits: one call of 20 against two calls in flight
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup 5 "$@" --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['ms_per_step'],4), flush=True)" || exit 1; }
for rep in 1 2; do
run
run --frames-in-flight 2 --batch-plan 10,10
run --frames-in-flight 2 --batch-plan 12,8
run --frames-in-flight 2 --batch-plan 14,6
run --frames-in-flight 2 --batch-plan 16,4
done
