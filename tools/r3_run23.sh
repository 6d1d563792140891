#!/bin/bash
# RT_RUN_HITS per value: one rank's share of tiled batches (parts 8 / 4 / 2, 10 frames per call) and one frame at a time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in ${VALS:-0 250000 1048576}; do
  echo "== RT_RUN_HITS=$v"
  RT_RUN_HITS=$v RT_ORDER0=1 timeout -k 10 300 python3 tools/exp_batch2.py bunny_1080p 8,4,2 10 2>/dev/null | grep "order0 1" || exit 1
done
WORKLOADS="bunny_1080p c1_scene_yaml_800x600" VAR=RT_RUN_HITS VALS="${VALS:-0 250000 1048576}" STEPS=64 bash tools/exp_env.sh
