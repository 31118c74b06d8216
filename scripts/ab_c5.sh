# usage: bash scripts/ab_c5.sh "<variant names>" [spp]   (config-5 stand-in under each tuning build; "default" = the product library)
for v in $1; do
  lib=$GRAFT_REPO_ROOT/ray-tracer-archive_amd/lib/variants/librt_hip_$v.so
  [ "$v" = default ] && lib=$GRAFT_REPO_ROOT/ray-tracer-archive_amd/lib/librt_hip.so
  RT_HIP_LIB=$lib timeout -k 10 300 python3 scripts/gpu_c5.py ${2:-32} 2>&1 | sed -n 2p | sed "s/^/$v /"
done
