#!/bin/bash
# Kernel timeline of one call of 20 frames, the scene's own stream (tools/one_frame.py), calls one after the other.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/bt2
ONE_FRAME_BATCH=${B:-20} ONE_FRAME_SYNC=1 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/bt2 -o p --output-format csv -- python3 tools/one_frame.py bunny_1080p 4 > gpurun_out/bt2.log 2>&1 || { tail -5 gpurun_out/bt2.log; exit 1; }
python3 tools/batch_timeline.py $(find gpurun_out/bt2 -name '*kernel_trace.csv' | head -1)
