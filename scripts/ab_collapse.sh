# leaves of up to n primitives (RT_LEAF_COLLAPSE) on the LDS scenes: bash scripts/ab_collapse.sh "<n values>" <scenes>
for k in $1; do
  RT_LEAF_COLLAPSE=$k timeout -k 10 200 python3 scripts/gpu_ab_env.py RT_NOTHING $2 2>&1 | awk 'NR%4==2' | sed "s/^/collapse $k /"
done
