#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "--- c5 1080p 4spp"
timeout -k 10 300 python -u bench.py --workload c5_field_10m_1080p_4spp --steps 10 --warmup 2 --no-cpu-baseline --watchdog 280 > gpurun_out/c5_1080.json 2> gpurun_out/c5_1080.err
python - <<'PY'
import json; d=json.loads(open('gpurun_out/c5_1080.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['rays_per_frame'], d['config']['scene_build_s'])
PY
echo "--- traffic"
bash tools/exp_traffic.sh c5_field_10m_1080p_4spp
echo "--- c5 full"
timeout -k 10 900 python -u bench.py --workload c5_field_10m_8k_64spp --steps 2 --warmup 1 --frames-in-flight 1 --watchdog 850 > gpurun_out/c5_full.json 2> gpurun_out/c5_full.err
python - <<'PY'
import json; d=json.loads(open('gpurun_out/c5_full.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['rays_per_frame'], d['config']['scene_build_s'], d.get('cpu_baseline'))
PY
