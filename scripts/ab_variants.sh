# usage (on the GPU box): bash scripts/ab_variants.sh "<variant names>" <scene> <W> <H> <spp> — one render-timing line per tuning build under lib/variants/
for v in $1; do
  export RT_HIP_LIB=$PWD/ray-tracer-archive_amd/lib/variants/librt_hip_$v.so
  echo "== $v"
  python scripts/gpu_render_once.py $2 $3 $4 $5 4 2 2>&1 | grep -v amdgpu.ids | tail -2
done
