for k in 1 3 6 12 30; do RT_LIST_PARK_COST=$k timeout -k 10 200 python3 scripts/gpu_ab_env.py RT_LIST_CULL final 2>&1 | sed -n 2p | sed "s/^/K=$k /"; done
