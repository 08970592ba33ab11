// Issue rate of v_mfma_i32_32x32x32_i8 and of the fp4 form of v_mfma_scale_f32_32x32x64_f8f6f4 on one SIMD (gfx950):
// cycles per instruction from s_memtime around a loop of 4 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void rate_i8(long long *out, int iters) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345) out[1000] = 1;
}
__global__ void rate_fp4(long long *out, int iters) {
    v8i a = {(int)threadIdx.x, 1, 2, 3, 0, 0, 0, 0}, b = {4, 5, 6, (int)threadIdx.x, 0, 0, 0, 0};
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        // cbsz = 4 / blgp = 4: both operands FP4 (E2M1); scales 1.0 (E8M0 127)
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, 127, 0, 127);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) out[1000] = 1;
}
int main() {
    long long *d, h[4];
    hipMalloc(&d, 1001 * 8);
    const int iters = 10000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rate_i8, dim3(1), dim3(64), 0, 0, d, iters);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("i8  32x32x32: %.1f shader-clock ticks per MFMA (one wave)\n", (double)h[0] / (4.0 * iters));
        hipLaunchKernelGGL(rate_fp4, dim3(1), dim3(64), 0, 0, d, iters);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("fp4 32x32x64: %.1f shader-clock ticks per MFMA (one wave)\n", (double)h[0] / (4.0 * iters));
    }
    // wall-clock rate with every SIMD busy: 1024 waves
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(rate_i8, dim3(1024), dim3(64), 0, 0, d, iters);
        else hipLaunchKernelGGL(rate_fp4, dim3(1024), dim3(64), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ops = 1024.0 * iters * 4 * 2.0 * 32 * 32 * (which == 0 ? 32 : 64);
        printf("%s all SIMDs: %.3f ms -> %.2f P(FL)OP/s, %.1f ns per MFMA per SIMD\n", which == 0 ? "i8 " : "fp4", ms, ops / ms / 1e12, ms * 1e6 / (iters * 4));
    }
    return 0;
}
