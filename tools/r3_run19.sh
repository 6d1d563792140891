#!/bin/bash
# Per-kernel summary of a workload's bench run (rocprofv3 --kernel-trace --stats).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
w=${WL:-c2_analytic_1080p_4spp}
mkdir -p gpurun_out; rm -rf gpurun_out/ks
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -o p -- python3 bench.py --workload $w --no-cpu-baseline --steps ${STEPS:-64} --warmup 16 > gpurun_out/ks.log 2>&1 || { tail -5 gpurun_out/ks.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/ks/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}% min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
