#!/bin/bash
# RT_PACKET (0 quads only, 1 packets at level 0, 2 packets at every large level) per workload
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in ${WORKLOADS:-bunny_1080p c1_scene_yaml_800x600 c5_field_10m_1080p_4spp}; do for p in ${MODES:-1 2}; do
  st=${STEPS:-64}; [ $w = c5_field_10m_8k_64spp ] && st=2
  RT_PACKET=$p timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w --steps $st --warmup ${WARMUP:-8} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w packet $p', round(d['ms_per_step'],4), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done
