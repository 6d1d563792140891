#!/bin/bash
# RT_WIDE_WAVES_SMALL (waves per workgroup of the per-lane launches of passes below the quad-service size): 0 (one) against 4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in 0 4; do
  echo "== RT_WIDE_WAVES_SMALL=$v"
  RT_WIDE_WAVES_SMALL=$v timeout -k 10 300 python3 tools/exp_batch2.py bunny_1080p 8,4 10 2>/dev/null | grep "order0 1" || exit 1
  RT_WIDE_WAVES_SMALL=$v timeout -k 10 300 python bench.py --warmup 24 --frames-per-batch 3 --steps 96 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('3 frames per call ->', round(d['ms_per_step'],4), flush=True)" || exit 1
done
WORKLOADS="bunny_1080p c1_scene_yaml_800x600 c5_field_10m_1080p_4spp" VAR=RT_WIDE_WAVES_SMALL VALS="0 4" STEPS=64 bash tools/exp_env.sh
