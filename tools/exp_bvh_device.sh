#!/bin/bash
# GPU-built linear BVH against the host's SAH tree: build time and frame time.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for w in ${WORKLOADS:-bunny_1080p c5_field_10m_1080p_4spp}; do for b in host device; do
  timeout -k 10 300 python -u bench.py --workload $w --bvh $b --no-cpu-baseline --steps ${STEPS:-64} --warmup 16 --no-extras --watchdog 280 > gpurun_out/bd_${b}_$w.json 2> gpurun_out/bd_${b}_$w.err
  python - $b $w <<'PY'
import json,sys; d=json.loads(open('gpurun_out/bd_%s_%s.json'%(sys.argv[1],sys.argv[2])).read().strip().splitlines()[-1])
c=d['config']
print(sys.argv[2], 'bvh', sys.argv[1], 'ms/frame %.4f'%d['ms_per_step'], 'bvh build %.4f s'%c['bvh_build_s'], 'scene create %.3f s'%c['scene_build_s'], 'identical', d['frame_identical_to_1gpu'], flush=True)
PY
done; done
