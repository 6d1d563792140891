# workgroups of 1, 2, 4 waves for the per-lane kernels: is the level-0 launch bound by the dispatcher's workgroup rate?
for b in 64 128 256; do
  echo "RT_WIDE_BLOCK=$b"
  RT_WIDE_BLOCK=$b timeout -k 10 100 python3 -u bench.py --no-cpu-baseline --steps 128 --warmup 32 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' batched', round(d['ms_per_step'],4), 'latency', round(d['ms_per_frame_latency'],4), d['roofline']['level0_launch_ms'], d['frame_identical_to_1gpu'])"
done
