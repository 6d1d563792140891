#!/bin/bash
# per-kernel times (serial launches, rocprofv3 --kernel-trace --stats) of a workload with and without sorted light loops
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${W:-c5_field_10m_8k_64spp}
for v in ${VALUES:-0 1048576}; do
  rm -rf gpurun_out/ks_$v
  RT_SORT_SHADE=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$v -o p -- python3 tools/one_frame.py $W ${N:-1} serial > gpurun_out/ks_$v.txt 2>&1 || { tail -5 gpurun_out/ks_$v.txt; exit 1; }
  echo "== $W sort_shade=$v"; cat gpurun_out/ks_$v.txt | grep "HIP events"
  python3 - gpurun_out/ks_$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print('%-90s calls %6s total %10.1f us avg %9.1f us' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3))
PY
  rm -rf gpurun_out/ks_$v
done
