#!/bin/bash
# Kernel timeline of one rank's call of B tiled frames (part 0 of N).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/rk
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/rk -o p --output-format csv -- python3 tools/one_rank.py ${N:-8} ${B:-10} 5 > gpurun_out/rk.log 2>&1 || { tail -5 gpurun_out/rk.log; exit 1; }
python3 tools/batch_timeline.py $(find gpurun_out/rk -name '*kernel_trace.csv' | head -1) 1000000
