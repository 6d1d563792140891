#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_tests3.txt 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r3_tests3.txt
for t in 0 4096 8192 16384; do
  RT_TAIL=$t timeout -k 10 200 python bench.py --no-cpu-baseline --steps 64 --warmup 32 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail $t', 'batched', round(d['ms_per_step'],4), 'latency', round(d['ms_per_frame_latency'],4), 'sync+D2H', round(d['ms_per_frame'],4), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done
for w in c1_scene_yaml_800x600 c3_bunny_1080p_shadows c5_field_10m_1080p_4spp; do for t in 0 8192; do
  RT_TAIL=$t timeout -k 10 200 python bench.py --no-cpu-baseline --steps 64 --warmup 32 --workload $w 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w tail $t', 'batched', round(d['ms_per_step'],4), 'latency', round(d['ms_per_frame_latency'],4), 'sync+D2H', round(d['ms_per_frame'],4), d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style steps 20', 'batched', round(d['ms_per_step'],4), 'latency', round(d['ms_per_frame_latency'],4))"
