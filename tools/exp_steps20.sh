#!/bin/bash
# the driver's short run (--steps 20 --warmup 5): frames per call x calls in flight
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for fb in ${FBS:-4 5 7 10}; do for fl in ${FLS:-2 3}; do
  timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup 5 --frames-per-batch $fb --frames-in-flight $fl --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per call $fb in flight $fl ->', d['config']['frames_per_batch'], round(d['ms_per_step'],4), flush=True)" || exit 1
done; done
