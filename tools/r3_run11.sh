#!/bin/bash
# sorted light loops (hits of bounce levels sorted by position before their shadow packets): threshold and key-width sweep
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in ${WORKLOADS:-c5_field_10m_8k_64spp c4_bunny_glass_4k_16spp}; do for v in ${VALUES:-0 1048576}; do for b in ${BITS:-8 10}; do
  [ $v = 0 ] && [ $b != 8 ] && continue
  st=16; wu=8; case $w in c5_field_10m_8k_64spp) st=2; wu=4;; esac
  RT_SORT_BITS=$b RT_SORT_SHADE=$v timeout -k 10 400 python bench.py --no-cpu-baseline --no-extras --workload $w --steps $st --warmup $wu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w sort_shade=$v bits=$b', round(d['ms_per_step'],3), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done; done
