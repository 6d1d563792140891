#!/bin/bash
# sorted light loops within windows of the pool order
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in ${WORKLOADS:-c5_field_10m_8k_64spp}; do for cfg in ${CFGS:-0:8:32 1048576:8:32 1048576:8:20 1048576:8:16 1048576:7:14 1048576:6:12 1048576:8:18}; do
  IFS=: read v b win <<< "$cfg"
  st=16; wu=8; case $w in c5_field_10m_8k_64spp) st=2; wu=4;; esac
  RT_SORT_WINDOW=$win RT_SORT_BITS=$b RT_SORT_SHADE=$v timeout -k 10 400 python bench.py --no-cpu-baseline --no-extras --workload $w --steps $st --warmup $wu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w sort_shade=$v bits=$b window=$win', round(d['ms_per_step'],3), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done
