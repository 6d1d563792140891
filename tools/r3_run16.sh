#!/bin/bash
# Kernel timeline of the driver's short run (one call of 20 frames).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/bt
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/bt -o p --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/bt.log 2>&1 || { tail -5 gpurun_out/bt.log; exit 1; }
tail -1 gpurun_out/bt.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python3 tools/batch_timeline.py $(find gpurun_out/bt -name '*kernel_trace.csv' | head -1)
