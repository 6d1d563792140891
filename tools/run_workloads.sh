# one bench line per workload (no CPU baseline), plus the 2-rank gloo rehearsal of the default workload on one GPU
for w in c1_scene_yaml_800x600 c2_analytic_1080p_4spp c3_bunny_1080p_shadows bunny_1080p c4_bunny_glass_4k_16spp c5_field_10m_1080p_4spp; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps ${STEPS:-128} --warmup ${WARMUP:-32} > gpurun_out/w_$w.json 2> gpurun_out/w_$w.err || { echo "FAILED $w"; tail -5 gpurun_out/w_$w.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/w_$w.json')); print('$w', 'batched', round(d['ms_per_step'],4), 'ms/frame', round(d['value']), 'Mrays/s | rt_render sync+D2H', round(d['ms_per_frame'],3), '| one call one frame', round(d['ms_per_frame_latency'],3), '| host pipelined', round(d['ms_per_frame_host'],3), '| B', d['config']['frames_per_batch'], d['config']['rays_per_frame'], 'rays', d['config']['level_rays'][:6], d['frame_identical_to_1gpu'])"
done
# BASELINE config 5 whole on ONE GPU (its 8-GPU form gives each rank an eighth of the tiles): 2 frames
timeout -k 10 600 python bench.py --workload c5_field_10m_8k_64spp --steps 2 --warmup 1 > gpurun_out/w_c5_full.json 2> gpurun_out/w_c5_full.err || { echo "FAILED c5 full"; tail -5 gpurun_out/w_c5_full.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/w_c5_full.json')); print('c5_field_10m_8k_64spp', round(d['ms_per_step'],1), 'ms', round(d['value']), 'Mrays/s', d['config']['rays_per_frame'], 'rays', 'scene build', d['config']['scene_build_s'], 's', 'cpu', d['cpu_baseline'])"
timeout -k 10 300 python bench.py --gpus 2 --steps 8 --warmup 4 --backend gloo --single-device --no-cpu-baseline > gpurun_out/w_n2.json 2> gpurun_out/w_n2.err || { echo "FAILED n2"; tail -5 gpurun_out/w_n2.err; exit 1; }
python3 -c "
import json; d=json.loads(open('gpurun_out/w_n2.json').read().strip().splitlines()[-1]); print('n2 gloo', round(d['ms_per_step'],3), d['n_gpus'], d['rccl_ranks'], d['frame_identical_to_1gpu'])"
