#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_tests1.txt 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r3_tests1.txt
