#!/bin/bash
# Kernel timeline of one frame at a time (the call waits for its frame), per RT_TAIL value.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out
for t in ${TAILS:-0 768}; do
  rm -rf gpurun_out/lat_$t
  RT_TAIL=$t ONE_FRAME_SYNC=1 timeout -k 10 120 rocprofv3 --kernel-trace -d gpurun_out/lat_$t -o p --output-format csv -- python3 tools/one_frame.py ${WL:-bunny_1080p} 8 > gpurun_out/lat_$t.log 2>&1 || { tail -5 gpurun_out/lat_$t.log; exit 1; }
  f=$(find gpurun_out/lat_$t -name '*kernel_trace.csv' | head -1)
  echo "== RT_TAIL=$t"; python3 tools/lat_timeline.py $f
done
