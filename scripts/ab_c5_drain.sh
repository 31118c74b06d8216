# config-5 stand-in against the hand-over point to the drain kernel (RT_DRAIN_AT = live paths): bash scripts/ab_c5_drain.sh [spp]
for d in 262144 1048576 2097152 4194304 8388608 16777216 33554432; do
  RT_DRAIN_AT=$d timeout -k 10 300 python3 scripts/gpu_c5.py ${1:-32} 2>&1 | sed -n 2p | sed "s/^/drain_at $d /"
done
