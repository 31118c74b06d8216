for m in 7 5 1 4 3 6; do
  RT_OCTANT_AXES=$m timeout -k 10 300 python3 scripts/gpu_c5_count.py 16 2>&1 | sed -n 1p | sed "s/^/axes $m /"
  RT_OCTANT_AXES=$m timeout -k 10 300 python3 scripts/gpu_c5.py 32 2>&1 | sed -n 2p | sed "s/^/axes $m /"
done
