for g in 1000000 28672 14336 7168 4096 3584; do
  echo "RT_GRID0=$g"
  RT_GRID0=$g timeout -k 10 100 python3 -u bench.py --no-cpu-baseline --steps 64 --warmup 16 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' batched', round(d['ms_per_step'],4), 'latency', round(d['ms_per_frame_latency'],4), d['roofline']['level0_launch_ms'], d['frame_identical_to_1gpu'])"
done
