#!/bin/bash
# c5 (10 M triangles, 8K, 64 spp): pass size x the level size from which a bounce level runs as packets.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for pl in ${PASSES:-23 25 26}; do for thr in ${THRS:-1048576 262144}; do
  RT_PASS_LOG2=$pl RT_LARGE_THR=$thr timeout -k 10 400 python -u bench.py --workload ${WL:-c5_field_10m_8k_64spp} --steps 2 --warmup 2 --frames-in-flight 1 --no-extras --watchdog 380 2>gpurun_out/c5p.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pass 2^$pl thr $thr', round(d['ms_per_step'],1), 'ms', round(d['value']/1e3,2), 'Grays/s', d['frame_identical_to_1gpu'], flush=True)" || { tail -3 gpurun_out/c5p.err; exit 1; }
done; done
