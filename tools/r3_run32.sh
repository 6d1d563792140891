#!/bin/bash
# waves per workgroup of a lone frame's per-lane launches beyond four (a build with launch bounds of 512 threads)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
for cfg in "b256 4" "b512 4" "b512 8"; do set -- $cfg
  RT_AMD_LIB=$PWD/tools/bin/librt_amd_$1.so RT_WIDE_WAVES_SMALL=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 64 --warmup 32 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 waves $2', 'batched', round(d['ms_per_step'],4), 'sync+D2H', round(d.get('ms_per_frame',0),4), 'latency', round(d.get('ms_per_frame_latency',0),4), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done
