#!/bin/bash
# Frames per call (batch size in samples) for the default workload, now that a pass may hold 2^26 samples.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for bs in ${SIZES:-8388608 16777216 33554432 67108864}; do for rep in 1 2; do
  timeout -k 10 300 python -u bench.py --workload ${WL:-bunny_1080p} --steps ${STEPS:-256} --warmup 64 --batch-samples $bs --no-extras --watchdog 280 2>gpurun_out/bs.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch samples $bs', 'B', d['config']['frames_per_batch'], round(d['ms_per_step'],4), 'ms', round(d['value']/1e3,2), 'Grays/s', d['frame_identical_to_1gpu'], flush=True)" || { tail -3 gpurun_out/bs.err; exit 1; }
done; done
