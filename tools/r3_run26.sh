#!/bin/bash
# Which kernels run together, and for how long, when two calls of 32 frames are in flight.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/tf
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/tf -o p --output-format csv -- python3 tools/two_in_flight.py ${B:-32} ${CALLS:-10} > gpurun_out/tf.log 2>&1 || { tail -5 gpurun_out/tf.log; exit 1; }
python3 tools/busy_timeline.py $(find gpurun_out/tf -name "*kernel_trace.csv" | head -1) 20736000 ${SKIP:-2}
