# per-level kernel durations under the RT_DEBUG_SKIP ablations (1: no shadow rays, 4: closest only, 2: neither)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 4 2; do
rm -rf gpurun_out/prof_lv
RT_DEBUG_SKIP=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_lv -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline ${@} > gpurun_out/lv.json 2> gpurun_out/lv.err || exit 1
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open('gpurun_out/prof_lv/b_kernel_trace.csv')))
d = json.load(open('gpurun_out/lv.json'))
tl = [r for r in rows if 'k_trace_level' in r['Kernel_Name'] and 'true' not in r['Kernel_Name']]
n = d['config']['max_depth'] + 1
print('skip $v ms/frame', round(d['ms_per_frame'], 4), 'levels us', [round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1) for r in tl[-n:]])
PY
done
