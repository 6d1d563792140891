#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS="cur w8 w8s" WORKLOADS="bunny_1080p c5_field_10m_1080p_4spp" bash tools/exp_ab_builds.sh run 2>&1 | tee gpurun_out/r3_ab_w8.txt
