#!/bin/bash
# A/B of library builds that differ in compile-time parameters. Locally (no GPU):
#   tools/exp_ab_builds.sh build name1 "-DRT_WPE_PK=6" name2 "-DRT_LEAF_SERVE_MAX=8" ...   -> tools/bin/librt_amd_<name>.so
# On the GPU box:  LIBS="name1 name2" WORKLOADS="..." bash tools/exp_ab_builds.sh run
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  shift; mkdir -p tools/bin
  while [ $# -ge 2 ]; do
    name=$1; extra=$2; shift 2
    d=$(mktemp -d); cp raytracer_amd/csrc/*.hip raytracer_amd/csrc/*.cpp raytracer_amd/csrc/*.h raytracer_amd/csrc/Makefile $d/
    mkdir -p $d/../host && cp -r raytracer_amd/host $d/../ 2>/dev/null
    sed -i "s#-I../../include#-I$PWD/include#; s#../../include/#$PWD/include/#g; s#../host/#$PWD/raytracer_amd/host/#g" $d/Makefile
    make -C $d -j4 EXTRA="$extra" > /dev/null 2>&1 && cp $d/librt_amd.so tools/bin/librt_amd_$name.so && echo "built $name ($extra)" || echo "FAILED $name"
    rm -rf $d
  done
  exit 0
fi
for w in ${WORKLOADS:-bunny_1080p}; do for rep in 1 2; do for l in $LIBS; do
  st=${STEPS:-128}; wu=${WARMUP:-32}
  case $w in c5_field_10m_8k_64spp) st=2; wu=2;; c4_bunny_glass_4k_16spp) st=16; wu=4;; esac
  lib=$PWD/tools/bin/librt_amd_$l.so; [ "$l" = tree ] && lib=$PWD/raytracer_amd/csrc/librt_amd.so
  RT_AMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w --steps $st --warmup $wu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $l', 'batched', round(d['ms_per_step'],4), 'latency', round(d.get('ms_per_frame_latency',0),4), 'level0', {k.split('<')[0]: round(v,4) for k,v in d['roofline']['level0_launch_ms'].items()}, 'batch', round(d['roofline'].get('batched_launch',{}).get('ms_per_frame',0),4), {k.split('<')[0]: round(v,4) for k,v in d['roofline'].get('batched_call_serial',{}).get('ms_per_frame_by_kernel',{}).items()}, d['frame_identical_to_1gpu'], flush=True)" || exit 1
done; done; done
