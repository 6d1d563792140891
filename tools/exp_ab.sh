#!/bin/bash
# A/B of two builds of the library on one box: tools/tmp/lib<A>.so, tools/tmp/lib<B>.so (ms per frame batched, and one frame at a time)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${WORKLOAD:-c4_bunny_glass_4k_16spp}
for rep in 1 2; do for v in "$@"; do
  cp tools/tmp/lib$v.so raytracer_amd/csrc/librt_amd.so
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $W --steps ${STEPS:-16} --warmup ${WARMUP:-4} --frames-per-batch ${FPB:-32} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W lib $v', round(d['ms_per_step'],4), (d.get('one_frame_at_a_time') or {}).get('ms_per_frame'), d['config']['pool_adapt_rounds'], d['config']['frames_in_flight'], d['config']['frames_per_batch'], d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done
