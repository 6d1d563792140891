#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
LIBS="nt0 nt1" WORKLOADS="bunny_1080p c4_bunny_glass_4k_16spp c5_field_10m_1080p_4spp c5_field_10m_8k_64spp" bash tools/exp_ab_builds.sh run 2>&1 | tee gpurun_out/r3_ab_nt.txt
