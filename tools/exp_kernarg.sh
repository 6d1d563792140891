for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 python bench.py --no-cpu-baseline --workload c2_analytic_1080p_4spp --steps 20 > gpurun_out/ka_$v.json 2> gpurun_out/ka_$v.err
  echo "devkernarg $v rc=$?"; cut -c1-160 gpurun_out/ka_$v.json; tail -2 gpurun_out/ka_$v.err
done
