#!/bin/bash
# per-kernel serial times of one frame (rt_stats) for a build of the library: tools/tmp/lib<V>.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${WORKLOAD:-c4_bunny_glass_4k_16spp}
for v in "$@"; do
  cp tools/tmp/lib$v.so raytracer_amd/csrc/librt_amd.so
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $W --steps ${STEPS:-4} --warmup ${WARMUP:-2} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$W lib $v', round(d['ms_per_step'],3), 'serial', round(r['serial_frame_ms'],3))
for k,x in r['kernels'].items(): print('   ', k, x['launches_per_frame'], round(x['ms_per_frame_serial'],3))" || exit 1
done
