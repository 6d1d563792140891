#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_tests5.txt 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r3_tests5.txt
WORKLOADS="bunny_1080p c1_scene_yaml_800x600 c2_analytic_1080p_4spp c3_bunny_1080p_shadows c5_field_10m_1080p_4spp c4_bunny_glass_4k_16spp" LIM=300 bash tools/r3_run2.sh
