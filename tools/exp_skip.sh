for v in 0 1 2 4 5; do
  RT_DEBUG_SKIP=$v timeout -k 10 200 python bench.py --no-cpu-baseline --workload ${1:-c2_analytic_1080p_4spp} --steps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('skip $v', round(d['value']), round(d['ms_per_frame'],4), d['config']['rays_per_frame'])"
done
