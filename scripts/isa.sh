#!/bin/bash
# usage: scripts/isa.sh <mangled-prefix>   -> /tmp/kern.s with that kernel's ISA
cd /root/repo/ray-tracer-archive_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -S --cuda-device-only kernels.hip -o /tmp/kernels.s 2>&1 | grep -E "error" 
n=$(grep -n "^$1.*:" /tmp/kernels.s | head -1 | cut -d: -f1)
awk -v n=$n 'NR>=n' /tmp/kernels.s | awk '{print} /s_endpgm/{exit}' > /tmp/kern.s
wc -l /tmp/kern.s
