#!/bin/bash
# local: build a waves-per-SIMD variant, then run it on the GPU.  usage: exp_wpe.sh "-DRT_WPE_SQ=5" [workloads...]
set -e
cd /root/repo
touch raytracer_amd/csrc/rt_kernels.hip
make -C raytracer_amd/csrc EXTRA="$1" 2>&1 | grep -E "error|shared" || true
shift
/usr/local/graft/bin/gpurun --timeout 900 -- "WORKLOADS='${*:-bunny_1080p c5_field_10m_1080p_4spp}' RUNS=3 bash tools/exp_run.sh" 2>&1 | grep "^run_log2" | sed 's/run_log2 3 //'
