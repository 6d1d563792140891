#!/bin/bash
# the exchange path on one rank with RCCL initialised: hardware queues per process
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for q in ${QS:-4 6 8 12 16}; do
  GPU_MAX_HW_QUEUES=$q RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 300 python bench.py --gpus 1 --rehearse-exchange --no-cpu-baseline ${@} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hw queues $q', round(d['ms_per_step'],4), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done
