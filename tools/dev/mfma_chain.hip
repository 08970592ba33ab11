// Issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 (FP4) as a function of the number of independent accumulator chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ void k(float *out, int iters) {
    v8i a = {(int)threadIdx.x, 1, 2, 3, 0, 0, 0, 0}, b = {4, 5, 6, (int)threadIdx.x, 0, 0, 0, 0};
    v16f c[4] = {{0}, {0}, {0}, {0}};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r % CHAINS] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[r % CHAINS], 4, 4, 0, 127, 0, 127);
    }
    if (c[0][0] + c[1][1] + c[2][2] + c[3][3] == 12345.f) out[0] = 1;
}
int main() {
    float *d; (void)hipMalloc(&d, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 10000;
    for (int waves = 1; waves <= 3; waves += 2)
        for (int ch = 1; ch <= 4; ch *= 2) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                const dim3 grid(256), block(256 * waves);      // `waves` waves per SIMD on every CU
                if (ch == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters);
                if (ch == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, iters);
                if (ch == 4) hipLaunchKernelGGL(k<4>, grid, block, 0, 0, d, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%d wave(s) per SIMD, %d independent chain(s): %.1f ns per MFMA per SIMD\n", waves, ch, best * 1e6 / (iters * 4.0 * waves));
        }
    return 0;
}
