#!/bin/bash
# one rank's share of ten tiled frames per call (two calls in flight): bounce levels by quads or by quad service
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in -1 0 1; do
  echo "== RT_SERVE=$v"
  RT_SERVE=$v timeout -k 10 300 python3 tools/exp_batch2.py bunny_1080p 8,4,2 10 2>/dev/null | grep "order0 1" || exit 1
done
