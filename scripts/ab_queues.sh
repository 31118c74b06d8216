for v in q8wg q32wg q16wave q32wave q64wave; do
  export RT_HIP_LIB=$PWD/ray-tracer-archive_amd/lib/variants/librt_hip_$v.so
  echo "== $v"
  python scripts/gpu_render_once.py book1 1200 800 500 3 2 2>&1 | grep -v amdgpu.ids | tail -1
  python scripts/gpu_render_once.py cornell 600 600 500 3 2 2>&1 | grep -v amdgpu.ids | tail -1
  python scripts/gpu_render_once.py final 800 800 200 3 2 2>&1 | grep -v amdgpu.ids | tail -1
done
