#!/bin/bash
# k_tail threshold sweep: forms test, then throughput / latency figures of the default bench per RT_TAIL value.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forms_agree" --timeout 200 --timeout-method thread 2>&1 | tail -5
for t in ${TAILS:-0 256 768 2048 8192}; do
  RT_TAIL=$t timeout -k 10 200 python -u bench.py --no-cpu-baseline --watchdog 180 ${BENCH_ARGS} > gpurun_out/bench_tail_$t.json 2> gpurun_out/bench_tail_$t.err
  python - $t <<'PY'
import json,sys; d=json.loads(open('gpurun_out/bench_tail_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('RT_TAIL', sys.argv[1], {k:round(d[k],4) for k in ('value','ms_per_step','ms_per_frame','ms_per_frame_latency','ms_per_frame_host')}, d['frame_identical_to_1gpu'])
PY
done
