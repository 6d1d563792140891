#!/bin/bash
# packets below level 0: the light loops of the large bounce levels by quad service instead of packets (RT_SHADE_SERVE_LV)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in ${WORKLOADS:-c5_field_10m_8k_64spp c4_bunny_glass_4k_16spp}; do for v in 0 1 2; do
  st=16; wu=4; case $w in c5_field_10m_8k_64spp) st=2; wu=2;; esac
  RT_SHADE_SERVE_LV=$v timeout -k 10 400 python bench.py --no-cpu-baseline --no-extras --workload $w --steps $st --warmup $wu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w shade_serve_lv=$v', round(d['ms_per_step'],3), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done
