#!/bin/bash
# local: build a waves-per-SIMD variant, then run the default bench + c5 1080p on the GPU.  usage: exp_wpe2.sh "-DRT_WPE_SQ=5"
set -e
cd /root/repo
touch raytracer_amd/csrc/rt_kernels.hip
make -C raytracer_amd/csrc EXTRA="$1" 2>&1 | grep -E "error|shared" || true
/usr/local/graft/bin/gpurun --timeout 900 -- 'for w in bunny_1080p c5_field_10m_1080p_4spp c4_bunny_glass_4k_16spp; do timeout -k 10 300 python -u bench.py --workload $w --no-cpu-baseline --steps 64 --warmup 16 > gpurun_out/wpe_$w.json 2> gpurun_out/wpe_$w.err; python3 -c "
import json,sys; d=json.loads(open(\"gpurun_out/wpe_$w.json\").read().strip().splitlines()[-1]); print(\"$w\", \"ms/frame %.4f\" % d[\"ms_per_step\"], \"one at a time %.4f\" % d[\"one_frame_at_a_time\"][\"ms_per_frame\"])"; done' 2>&1 | grep "ms/frame"
