#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in bunny_1080p c4_bunny_glass_4k_16spp c5_field_10m_1080p_4spp; do
 st=64; wu=32; case $w in c4_bunny_glass_4k_16spp) st=16; wu=4;; esac
 for g in 7168 6144 5120 3584 2048; do
  RT_WIDE_GRID=$g timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps $st --warmup $wu --workload $w 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w wide_grid $g', 'batched', round(d['ms_per_step'],4), d['frame_identical_to_1gpu'], flush=True)" || exit 1
 done
done
