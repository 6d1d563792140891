#!/bin/bash
# calls in flight x frames per call after run_hits
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout -k 10 300 python bench.py --steps ${STEPS:-128} --warmup 32 "$@" --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['ms_per_step'],4), flush=True)" || exit 1; }
for rep in 1 2; do
run --frames-in-flight 1
run --frames-in-flight 2
run --frames-in-flight 3
run --frames-in-flight 2 --frames-per-batch 16
run --frames-in-flight 3 --frames-per-batch 16
done
