#!/bin/bash
# The round's record: GPU tests, the default bench line, rocprofv3 --kernel-trace --stats of the same command and of
# serially launched frames, one line per workload. Output under gpurun_out/final/ (copy the summaries to profiles/rN/).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $O/gpu_tests.txt 2>&1 || { tail -20 $O/gpu_tests.txt; exit 1; }
tail -2 $O/gpu_tests.txt
timeout -k 10 400 python -u bench.py --watchdog 380 > $O/bench_bunny_1080p.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 bench.py --no-cpu-baseline --watchdog 380 > $O/bench_under_rocprof.json 2> $O/prof_bench.err || { tail -5 $O/prof_bench.err; exit 1; }
cp $(find $O/prof_bench -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv && rm -rf $O/prof_bench
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o serial -- python3 tools/one_frame.py bunny_1080p 20 serial > $O/serial_hip_events.txt 2> $O/prof_serial.err || { tail -5 $O/prof_serial.err; exit 1; }
cp $(find $O/prof_serial -name '*kernel_stats.csv' | head -1) $O/serial_kernel_stats.csv && rm -rf $O/prof_serial
bash tools/run_workloads.sh > $O/workloads.txt 2>&1 || { tail -5 $O/workloads.txt; exit 1; }
cp gpurun_out/w_c5_full.json $O/bench_c5_field_10m_8k_64spp.json; cp gpurun_out/w_c4_bunny_glass_4k_16spp.json $O/bench_c4_bunny_glass_4k_16spp.json
cat $O/workloads.txt
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench_bunny_1080p.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','ms_per_frame','ms_per_frame_latency','ms_per_frame_host','vs_cpu_baseline')})
r=d['roofline']; print({k:r.get(k) for k in ('bound','achieved','peak','frac','traffic','kernel','avg_launch_ms')}); print(d['cpu_baseline'])
PY
