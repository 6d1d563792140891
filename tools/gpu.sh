#!/bin/bash
# Local wrapper: stamps the commit the tree was built from (the GPU box has no .git), then runs a command on the box.
#   tools/gpu.sh [--timeout S] -- <command>
cd "$(dirname "$0")/.."
c=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- raytracer_amd include || c="$c+uncommitted"
echo "$c" > tools/.commit_stamp
exec /usr/local/graft/bin/gpurun "$@"
