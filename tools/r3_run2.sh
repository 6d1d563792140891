#!/bin/bash
# counters of the current kernels, then the bench that reads them
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p profiles/r3 gpurun_out
true
for w in ${WORKLOADS:-bunny_1080p}; do
  bash tools/collect_counters.sh $w ${LIM:-240} > gpurun_out/counters_table_$w.txt 2>&1 || { tail -5 gpurun_out/counters_table_$w.txt; exit 1; }
  cp gpurun_out/counters_$w.json profiles/r3/
  st=128; wu=32
  case $w in c5_field_10m_8k_64spp) st=2; wu=2;; c4_bunny_glass_4k_16spp) st=16; wu=4;; esac
  timeout -k 10 ${LIM:-240}0 python -u bench.py --workload $w --steps $st --warmup $wu --watchdog 900 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err || { tail -5 gpurun_out/bench_$w.err; exit 1; }
  python3 - $w <<'PY'
import json,sys; d=json.loads(open('gpurun_out/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'frac', round(r['frac'],4), 'clock', r['shader_clock_mhz'], 'stale', r['counters_stale'], 'traffic', r['traffic'])
print(' valu_issue', {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.get('valu_issue',{}).items() if k!='note'})
print(' batched', {k:(round(v,4) if isinstance(v,float) else v) for k,v in r.get('batched_launch',{}).items() if k!='note'})
print(' cpu', d.get('cpu_baseline'))
PY
done
