#!/bin/bash
# BASELINE config 5 on one GPU: parity of the small field, then the 10 M-triangle frame (7680x4320, 64 spp).
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "field" --timeout 200 --timeout-method thread 2>&1 | tail -5
echo "--- c5 small"
timeout -k 10 200 python -u bench.py --workload c5_field_small_720p --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/c5_small.json 2> gpurun_out/c5_small.err
python - <<'PY'
import json; d=json.loads(open('gpurun_out/c5_small.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['rays_per_frame'], d['config']['scene_build_s'])
PY
echo "--- c5 full"
timeout -k 10 900 python -u bench.py --workload c5_field_10m_8k_64spp --steps 2 --warmup 1 --frames-in-flight 1 --watchdog 850 > gpurun_out/c5_full.json 2> gpurun_out/c5_full.err
python - <<'PY'
import json; d=json.loads(open('gpurun_out/c5_full.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['rays_per_frame'], d['config']['scene_build_s'], d.get('cpu_baseline'))
PY
