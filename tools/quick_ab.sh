#!/bin/bash
# parity tests of the traversal / frames, then the batched figure of a few workloads: the check after a kernel change
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
for w in ${WORKLOADS:-bunny_1080p c2_analytic_1080p_4spp c5_field_10m_1080p_4spp}; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --workload $w 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['ms_per_step'],4), 'ms', round(d['value']/1e3,2), 'Grays/s latency', round(d['ms_per_frame_latency'],4), 'sync+D2H', round(d['ms_per_frame'],4), d['frame_identical_to_1gpu'], {k.split('<')[0]: round(v,4) for k,v in d['roofline']['level0_launch_ms'].items()}, flush=True)" || exit 1
done
