// Probe of the accumulator layout of v_mfma_i32_32x32x32_i8 (gfx950): which (row, column) of D does acc[i] of lane l hold?
// A[m][k] = m + 1 for all k (lane l supplies row l % 32), B[k][n] = 1 for k == 0 of each lane's chunk only in column n ... kept
// simple: two runs, one encodes rows, one encodes columns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void probe(int *rows, int *cols) {
    const int l = threadIdx.x;
    // run 1: A row value (l % 32) + 1 in byte 0 of lanes < 32 only (one k), B = 1 in the same k for every column  -> D[m][n] = m + 1
    v4i a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (l < 32) { a[0] = (l % 32) + 1; b[0] = 1; }
    v16i acc = {0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) rows[l * 16 + i] = acc[i] - 1;
    // run 2: A = 1, B column value (l % 32) + 1 -> D[m][n] = n + 1
    a = {0, 0, 0, 0}; b = {0, 0, 0, 0};
    if (l < 32) { a[0] = 1; b[0] = (l % 32) + 1; }
    v16i acc2 = {0};
    acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc2, 0, 0, 0);
    for (int i = 0; i < 16; ++i) cols[l * 16 + i] = acc2[i] - 1;
}
int main() {
    int *r, *c;
    hipMalloc(&r, 64 * 16 * 4); hipMalloc(&c, 64 * 16 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, r, c);
    int hr[1024], hc[1024];
    hipMemcpy(hr, r, sizeof(hr), hipMemcpyDeviceToHost); hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i) {
            const int want_row = 8 * (i / 4) + 4 * (l / 32) + (i % 4), want_col = l % 32;
            if (hr[l * 16 + i] != want_row || hc[l * 16 + i] != want_col) ++bad;
        }
    printf("lane 0 rows:"); for (int i = 0; i < 16; ++i) printf(" %d", hr[i]); printf("\nlane 33 rows:"); for (int i = 0; i < 16; ++i) printf(" %d", hr[33 * 16 + i]);
    printf("\nlane 33 cols:"); for (int i = 0; i < 16; ++i) printf(" %d", hc[33 * 16 + i]);
    printf("\nmismatches against row = 8 (i / 4) + 4 (l / 32) + i %% 4, col = l %% 32: %d\n", bad);
    return bad != 0;
}
