#!/bin/bash
# Frames whose passes fill the chip: one call at a time against two in flight (their bounce tails overlap the other's level 0).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for w in ${WORKLOADS:-c4_bunny_glass_4k_16spp c5_field_10m_8k_64spp}; do for f in 1 2; do
  st=16; wu=4; case $w in c5_field_10m_8k_64spp) st=4; wu=2;; esac
  timeout -k 10 500 python -u bench.py --workload $w --steps $st --warmup $wu --frames-in-flight $f --no-extras --watchdog 480 2>gpurun_out/inflight.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w in flight $f', round(d['ms_per_step'],2), 'ms', round(d['value']/1e3,2), 'Grays/s', d['frame_identical_to_1gpu'], flush=True)" || { tail -3 gpurun_out/inflight.err; exit 1; }
done; done
