#!/bin/bash
# Host code of the library (scene compiler, host mirror, C ABI's host-only entry points, multi-GPU plumbing) and the oracle under
# AddressSanitizer + UBSan, on the CPU (GPU sanitizers are not available on the pool). Runs the CPU test files that exercise them.
# usage: bash scripts/asan_host.sh            (from the repo root; ~2 minutes)
set -e
cd "$(dirname "$0")/.."
P=ray-tracer-archive_amd
mkdir -p $P/lib/variants oracle/_build
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
g++ -std=c++17 -fPIC -shared $SAN -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Wno-deprecated-declarations -Wno-unused-result \
    -o $P/lib/variants/librt_hip_asan.so $P/csrc/rt_api.cpp $P/csrc/rt_multi.cpp $P/csrc/scene_compile.cpp $P/csrc/wide_bvh.cpp $P/host/host_capi.cpp scripts/asan_host_stubs.cpp \
    -L/opt/rocm/lib -lamdhip64 -lz -ldl -Wl,-rpath,/opt/rocm/lib
g++ -std=c++17 -fPIC -shared $SAN -pthread -o oracle/_build/liboracle_asan.so oracle/oracle.cpp
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)"
export RT_HIP_LIB=$PWD/$P/lib/variants/librt_hip_asan.so RT_ORACLE_LIB=$PWD/oracle/_build/liboracle_asan.so
python -m pytest tests/test_compile.py tests/test_host.py tests/test_scene_json.py tests/test_dist_gloo.py tests/test_abi.py tests/test_oracle_kat.py -x -q "$@"
