// Micro-benchmark: sustained wave64 VALU issue rate on MI355X for the instructions the BF Hamming kernel is made of.
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[CHAINS];
    float f[CHAINS];
    for (int c = 0; c < CHAINS; ++c) {
        a[c] = seed + threadIdx.x * 17 + c * 3;
        f[c] = (float)a[c];
    }
    uint32_t s = seed | 1;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[c]) : "v"(s));
                if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[c]) : "v"(s));
                if (OP == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[c]) : "v"(s));
                if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[c]) : "v"(1.0001f));
                if (OP == 4) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[c]) : "v"(s));
                if (OP == 5) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[c]) : "v"(s));
                if (OP == 6) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[c]) : "s"(s));
            }
        }
    }
    uint32_t r = 0;
    for (int c = 0; c < CHAINS; ++c) r ^= a[c] ^ (uint32_t)f[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char *name, uint32_t *out) {
    const int blocks = 256 * 8, iters = 2000;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 12345u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    double lane_ops = (double)blocks * 256 * iters * 8 * CHAINS;
    double rate = lane_ops / (best * 1e-3);
    double cyc = 1024.0 * 2.4e9 / (rate / 64.0);  // cycles per wave-instruction per SIMD at 2.4 GHz
    printf("%-28s %8.3f ms  %7.2f T lane-ops/s  (%.2f cyc / wave-instr / SIMD at 2.4 GHz)\n", name, best, rate / 1e12, cyc);
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_xor_b32 (vgpr,vgpr)", out);
    run<6>("v_xor_b32 (sgpr,vgpr)", out);
    run<1>("v_bcnt_u32_b32", out);
    run<2>("v_add_u32", out);
    run<4>("v_min_u32", out);
    run<5>("v_lshl_or_b32", out);
    run<3>("v_fma_f32", out);
    return 0;
}
