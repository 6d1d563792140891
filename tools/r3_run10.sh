#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
LIBS="${LIBS:-base planercp}" WORKLOADS="${WORKLOADS:-bunny_1080p c2_analytic_1080p_4spp}" bash tools/exp_ab_builds.sh run
