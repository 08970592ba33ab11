// Probe: the panel factorisation of chol.hip (panel16_factor_dpp<PB>) exactly as the fused kernel instantiates it -- from
// and to LDS, one wave -- timed per panel index.  Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics
// tools/dev/panel_insitu.hip -o tools/dev/panel_insitu
#include "../../meatmodeler_amd/csrc/chol.hip"
#include <vector>

template <int PB, int DBG>
__global__ void insitu_kernel(const double *A, unsigned long long *ticks, int reps, double *out) {
    __shared__ double M[NB][NB + 1];
    __shared__ double X[NB][NB + 1];
    __shared__ double R[NB];
    const int lane = threadIdx.x;
    unsigned long long acc = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = lane; e < NB * NB; e += 64) M[e / NB][e % NB] = A[e];
        __syncthreads();
        const unsigned long long t0 = wall_clock64();
        panel16_factor_dpp<PB, DBG>(M, X, R);
        __syncthreads();
        const unsigned long long t1 = wall_clock64();
        if (rep) acc += t1 - t0;
    }
    if (lane == 0) ticks[0] = acc;
    out[lane] = M[lane][16 * PB] + X[16 * PB + 3][16 * PB + (lane & 15)] + R[lane];
}

int main() {
    std::vector<double> A(64 * 64);
    srand(3);
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) A[i * 64 + j] = (i == j ? 70.0 : 0.0) + (rand() / (double)RAND_MAX - 0.5);
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < i; ++j) A[j * 64 + i] = A[i * 64 + j];
    double *dA, *dO;
    unsigned long long *dT, ticks;
    (void)hipMalloc(&dA, A.size() * 8);
    (void)hipMalloc(&dO, 64 * 8);
    (void)hipMalloc(&dT, 8);
    (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const int reps = 201;
    for (int pb = 0; pb < 4; ++pb) {
        if (pb == 0) hipLaunchKernelGGL((insitu_kernel<0, 0>), dim3(1), dim3(64), 0, 0, dA, dT, reps, dO);
        if (pb == 1) hipLaunchKernelGGL((insitu_kernel<2, 0>), dim3(1), dim3(64), 0, 0, dA, dT, reps, dO);
        if (pb == 2) hipLaunchKernelGGL((insitu_kernel<2, 1>), dim3(1), dim3(64), 0, 0, dA, dT, reps, dO);
        if (pb == 3) hipLaunchKernelGGL((insitu_kernel<2, 3>), dim3(1), dim3(64), 0, 0, dA, dT, reps, dO);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        (void)hipMemcpy(&ticks, dT, 8, hipMemcpyDeviceToHost);
        printf("variant %d (0: PB 0 | 1: PB 2 | 2: PB 2 without the X write | 3: PB 2 without X write and identity rows): %.3f us per call\n", pb, ticks * 0.01 / (reps - 1));
    }
    return 0;
}
