for f in 1 2; do
  for w in c5_field_10m_1080p_4spp c5_field_10m_8k_64spp; do
    st=16; [ $w = c5_field_10m_8k_64spp ] && st=4
    timeout -k 10 280 python3 -u bench.py --workload $w --no-extras --steps $st --warmup 2 --frames-in-flight $f 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w in flight $f:', round(d['ms_per_step'],3), 'ms', round(d['value']/1e3,2), 'Grays/s', d['frame_identical_to_1gpu'], d['config']['frames_per_batch'])"
  done
done
