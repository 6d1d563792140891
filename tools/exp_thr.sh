#!/bin/bash
# small/large level threshold (RT_LARGE_THR, rays) per workload
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in ${WORKLOADS:-c5_field_10m_8k_64spp c4_bunny_glass_4k_16spp}; do for t in ${THRS:-131072 32768 8192 2048}; do
  st=${STEPS:-16}; [ $w = c5_field_10m_8k_64spp ] && st=2
  RT_LARGE_THR=$t timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps $st --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w thr $t', round(d['ms_per_step'],3), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done
