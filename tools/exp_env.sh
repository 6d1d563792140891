#!/bin/bash
# one environment variable, several values, several workloads: VAR=RT_SPLIT_SMALL VALS="0 1" WORKLOADS="..." bash tools/exp_env.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in ${WORKLOADS:-bunny_1080p}; do for rep in 1 2; do for x in $VALS; do
  st=${STEPS:-128}; wu=${WARMUP:-32}
  case $w in c5_field_10m_8k_64spp) st=2; wu=2;; c4_bunny_glass_4k_16spp) st=16; wu=4;; esac
  env $VAR=$x timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps $st --warmup $wu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $VAR=$x', 'batched', round(d['ms_per_step'],4), 'sync+D2H', round(d.get('ms_per_frame',0),4), 'latency', round(d.get('ms_per_frame_latency',0),4), 'host-pipelined', round(d.get('ms_per_frame_host',0),4), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done; done
