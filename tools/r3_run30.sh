#!/bin/bash
# host-side schedule knobs measured again at the round's final kernels (default run and the driver's short run)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { env "$1" timeout -k 10 200 python bench.py --no-cpu-baseline --gpus 1 --steps $2 --warmup $3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 steps $2', 'ms/step', round(d['ms_per_step'],4), 'latency', round(d.get('ms_per_frame_latency',0),4), d['frame_identical_to_1gpu'], flush=True)" || exit 1; }
for kv in X=0 RT_SHADE_STREAMS=2 RT_SHADE_STREAMS=1 RT_WIDE_BLOCK=128 RT_WIDE_BLOCK=256 RT_WIDE_GRID=14336 RT_WIDE_GRID=3584 RT_CHUNK_LOG2=2 RT_CHUNK_LOG2=6; do
  run $kv 128 32; run $kv 20 5
done
