#!/bin/bash
# GPU parity tests, then the default bench (and optionally other workloads given as arguments).
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -8
timeout -k 10 300 python -u bench.py --watchdog 280 > gpurun_out/bench.json 2> gpurun_out/bench.err
python - <<'PY'
import json; d=json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('one_frame_at_a_time'), d['frame_identical_to_1gpu'], d['roofline']['level0_launch_ms'])
print({k:(round(v['avg_launch_ms'],4), v['launches_per_frame'], round(v['ms_per_frame_serial'],4)) for k,v in d['roofline']['kernels'].items()})
PY
for w in "$@"; do
  timeout -k 10 600 python -u bench.py --workload $w --no-cpu-baseline --steps 32 --warmup 8 --watchdog 580 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err
  python - $w <<'PY'
import json,sys; d=json.loads(open('gpurun_out/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], {k:d[k] for k in ('value','ms_per_step')}, d.get('one_frame_at_a_time'), d['roofline']['level0_launch_ms'])
PY
done
