#!/bin/bash
# expensive groups first (RT_ORDER0): one frame at a time, level-0 launch durations and frame latency with and without
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in ${WORKLOADS:-bunny_1080p c3_bunny_1080p_shadows c1_scene_yaml_800x600 c5_field_10m_1080p_4spp c2_analytic_1080p_4spp}; do for rep in 1 2; do for v in 0 1; do
  RT_ORDER0=$v timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps 64 --warmup 32 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$w order0=$v', 'batched', round(d['ms_per_step'],4), 'latency', round(d.get('ms_per_frame_latency',0),4), 'sync+D2H', round(d.get('ms_per_frame',0),4), 'level0', {k.split('<')[0]: round(v,4) for k,v in r['level0_launch_ms'].items()}, 'frac', round(r['frac'],4), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done; done
