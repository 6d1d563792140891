#!/bin/bash
# lane traversal (one ray per lane in the serve kernels): parity through the in-tree build and through a build whose
# short stack always overflows (every ray falls back to quad service), then A/B against the build without it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -4 || exit 1
RT_AMD_LIB=$PWD/tools/bin/librt_amd_lanefb.so timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread -k "forms or traversal or schedule" 2>&1 | tail -4 || exit 1
LIBS="nolane tree" WORKLOADS="${WORKLOADS:-bunny_1080p c5_field_10m_1080p_4spp c4_bunny_glass_4k_16spp}" bash tools/exp_ab_builds.sh run
