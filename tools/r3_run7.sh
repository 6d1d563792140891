#!/bin/bash
# one frame split into bands over several scenes; the tail threshold on the 10 M-triangle field
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 300 python tools/exp_split_latency.py bunny_1080p 2>&1 | grep -v Warning || exit 1
for t in 8192 0 65536 1024; do
  RT_TAIL=$t timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 2 --workload c5_field_10m_8k_64spp 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 8K tail $t', round(d['ms_per_step'],2), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done
