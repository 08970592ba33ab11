// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 operands: +1 (0x2) / -1 (0xA) nibbles, scales 1.0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void probe(float *out, float c_init, int mode) {
    const int l = threadIdx.x;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    // mode 0: every element +1 in A and B -> D = 64 everywhere
    // mode 1: A row m has (m + 1) elements +1 then zeros in lanes < 32 only ... B all +1 -> D[m][n] = min(m + 1, 32)
    // mode 2: A all +1, B column n: first (n % 32) + 1 nibbles -1, rest +1 (lanes < 32) ; lanes >= 32 all +1 -> D = 64 - 2 (n + 1)
    for (int i = 0; i < 4; ++i) { a[i] = 0x22222222; b[i] = 0x22222222; }
    if (mode == 1) {
        for (int i = 0; i < 4; ++i) a[i] = 0;
        if (l < 32) for (int k = 0; k <= (l % 32); ++k) a[k / 8] |= 0x2 << (4 * (k % 8));
    }
    if (mode == 2 && l < 32) for (int k = 0; k <= (l % 32); ++k) b[k / 8] ^= 0x8 << (4 * (k % 8));
    v16f acc;
    for (int i = 0; i < 16; ++i) acc[i] = c_init;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 127, 0, 127);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = acc[i] - c_init;
}
int main() {
    float *d, h[1024];
    (void)hipMalloc(&d, sizeof(h));
    const float inits[2] = {0.f, 12582912.f};
    for (int mode = 0; mode < 3; ++mode)
        for (int ci = 0; ci < 2; ++ci) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, inits[ci], mode);
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("mode %d c=%g: lane0:", mode, inits[ci]); for (int i = 0; i < 16; ++i) printf(" %g", h[i]);
            printf(" | lane5:"); for (int i = 0; i < 4; ++i) printf(" %g", h[5 * 16 + i]);
            printf(" | lane37:"); for (int i = 0; i < 4; ++i) printf(" %g", h[37 * 16 + i]);
            printf("\n");
        }
    return 0;
}
