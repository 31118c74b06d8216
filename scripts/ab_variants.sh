# usage: bash scripts/ab_variants.sh "<variant names>" <scenes>   (variants built by scripts/sweep.py build; one gpu_ab_env line each, culling on)
for v in $1; do
  lib=$GRAFT_REPO_ROOT/ray-tracer-archive_amd/lib/variants/librt_hip_$v.so
  [ "$v" = default ] && lib=$GRAFT_REPO_ROOT/ray-tracer-archive_amd/lib/librt_hip.so
  RT_HIP_LIB=$lib timeout -k 10 200 python3 scripts/gpu_ab_env.py RT_NOTHING $2 2>&1 | awk 'NR%4==2' | sed "s/^/$v /"
done
