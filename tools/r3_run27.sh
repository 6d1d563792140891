#!/bin/bash
# k_resolve's duration per library build (one call of 20 frames under rocprofv3 --kernel-trace --stats)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for l in $LIBS; do
  rm -rf gpurun_out/rs
  RT_AMD_LIB=$PWD/tools/bin/librt_amd_$l.so ONE_FRAME_BATCH=20 ONE_FRAME_SYNC=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/rs -o p --output-format csv -- python3 tools/one_frame.py bunny_1080p 6 > gpurun_out/rs.log 2>&1 || { tail -5 gpurun_out/rs.log; exit 1; }
  echo "== $l"; grep -E "k_resolve|k_fold\(" $(find gpurun_out/rs -name '*kernel_stats.csv' | head -1) | cut -c1-120
done
