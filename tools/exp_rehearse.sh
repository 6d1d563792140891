#!/bin/bash
# cost of the exchange path on one rank (RCCL gather to itself, untile): BENCH_SKIP leaves parts out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for skip in none untile gather gather,untile; do
  BENCH_SKIP=$skip timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --rehearse-exchange --no-cpu-baseline ${@} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip $skip', round(d['ms_per_step'],4), d['config']['frames_per_batch'], flush=True)" || exit 1
done
