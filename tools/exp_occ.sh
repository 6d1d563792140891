# EQ0 under capped occupancy (extra LDS per workgroup): per-batch time vs kernel time
for l in 0 12000 36000 76000; do
echo "== RT_DEBUG_LDS=$l"
RT_DEBUG_LDS=$l timeout -k 10 120 python tools/exp_wavelog.py 0 | sed -n '1,5p;12,14p'
done
