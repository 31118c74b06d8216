// Issue rate of a few VALU instructions on gfx950: 8 resident waves per SIMD, each a stream of independent instructions (8 accumulators).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates scripts/ubench/valu_rates.hip ; run: /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP> __global__ void __launch_bounds__(512) k(unsigned* out, unsigned seed, int iters) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) {
#define X(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##n) : "v"(a0));
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##n) : "v"(a1));
                REP8(X)
#undef X
            } else if (OP == 2) {
#define X(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##n) : "v"(a1));
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(b##n) : "v"(a1), "v"(a2) : "vcc");
                REP8(X)
#undef X
            } else if (OP == 4) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f##n) : "v"(f1), "v"(f2));
                REP8(X)
#undef X
            } else if (OP == 5) {
#define X(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##n) : "v"(a1));
                REP8(X)
#undef X
            } else if (OP == 6) {
#define X(n) asm volatile("v_lshrrev_b64 %0, 27, %0" : "+v"(b##n));
                REP8(X)
#undef X
            } else if (OP == 7) {
#define X(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a##n) : "v"(a1), "v"(a2));
                REP8(X)
#undef X
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7) ^ (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}
template <int OP> static void run(const char* name, unsigned* out) {
    const int blocks = 256 * 4, iters = 2000;      // 4 x 512 threads per CU: 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, out, 1u, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(512), 0, 0, out, 1u, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double n = (double)blocks * 8 /*waves*/ * iters * 16 * 8;           // wave-instructions
    const double per_simd_cycle = n / (1024.0 * ms * 1e-3 * 2.4e9);           // wave-instructions per SIMD per cycle at 2.4 GHz
    printf("%-16s %8.2f ms  %.3f wave-instr per SIMD-cycle  = %.2f cycles per wave-instruction\n", name, ms, per_simd_cycle, 1.0 / per_simd_cycle);
}
int main() {
    unsigned* out; hipMalloc(&out, 256 * 4 * 512 * 4);
    run<0>("v_add_u32", out); run<4>("v_fma_f32", out); run<1>("v_mul_lo_u32", out); run<2>("v_mul_hi_u32", out); run<3>("v_mad_u64_u32", out);
    run<5>("v_mul_u32_u24", out); run<7>("v_mad_u32_u24", out); run<6>("v_lshrrev_b64", out);
    return 0;
}
