#!/bin/bash
# Kernel timeline of the last pass of a frame of a large workload (one frame per call).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
w=${WL:-c4_bunny_glass_4k_16spp}
mkdir -p gpurun_out; rm -rf gpurun_out/lt
ONE_FRAME_SYNC=1 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/lt -o p --output-format csv -- python3 tools/one_frame.py $w ${N:-3} > gpurun_out/lt.log 2>&1 || { tail -5 gpurun_out/lt.log; exit 1; }
python3 tools/lat_timeline.py $(find gpurun_out/lt -name '*kernel_trace.csv' | head -1)
