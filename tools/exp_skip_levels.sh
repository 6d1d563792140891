# kernel timeline under RT_DEBUG_SKIP ablations: 1 no shadow rays, 2 no closest hit, 4 discard hits, 8 no mesh, 16 no spheres/planes
for v in "$@"; do
echo "== RT_DEBUG_SKIP=$v"
RT_DEBUG_SKIP=$v bash tools/exp_levels.sh | head -8
done
