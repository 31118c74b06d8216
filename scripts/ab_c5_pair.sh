# config-5 stand-in with and without boxes around the spheres that share a leaf (RT_PAIR_BOXES: 0 never, 2 always): bash scripts/ab_c5_pair.sh [spp]
for v in ${VALS:-0 2 0 2}; do
  RT_PAIR_BOXES=$v timeout -k 10 300 python3 scripts/gpu_c5_count.py 16 2>&1 | sed -n 1p | sed "s/^/RT_PAIR_BOXES=$v /"
  RT_PAIR_BOXES=$v timeout -k 10 300 python3 scripts/gpu_c5.py ${1:-32} 2>&1 | sed -n 2p | sed "s/^/RT_PAIR_BOXES=$v /"
done
