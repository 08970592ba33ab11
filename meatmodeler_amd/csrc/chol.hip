// Dense SPD solve by blocked right-looking Cholesky, f64, trailing updates on v_mfma_f64_16x16x4_f64 (gfx950).
//
// This is the dense linear algebra of the trust-region step: the reduced camera system S (6F x 6F) that the
// reference hands to LSMR implicitly (scipy trf.py:480 via bundleAdjuster.py:180-192) is factored here.
//   step k:  (1) chol_diag   : L_kk = chol(A_kk) in LDS, plus L_kk^-1 (kept in the workspace for the solves)
//            (2) chol_panel  : A_ik <- A_ik L_kk^-T          (64x64x64 MFMA GEMM per row block, B operand = L_kk^-1)
//            (3) chol_update : A_ij <- A_ij - A_ik A_jk^T    (64x64x64 MFMA GEMM per block pair i >= j > k)
// Row-major A, lower triangle referenced / overwritten.  Solves use the stored inverse diagonal blocks so both
// substitutions are GEMVs.  MFMA fragment maps for f64 16x16x4 (guide §3): A lane l -> A[l&15][l>>4],
// B lane l -> B[l>>4][l&15], D reg i of lane l -> D[(l>>4) + 4 i][l&15].
#include "mm_common.h"

namespace {

constexpr int NB = 64;
constexpr int LDT = 66;  // LDS tile leading dimension (doubles): 66 keeps the 32-lane ds_read_b64 groups conflict-free
typedef double double4_t __attribute__((ext_vector_type(4)));

// ---- (1) diagonal block ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_diag_kernel(double *__restrict__ A, int n, int k0, double *__restrict__ Linv,
                                                        int32_t *__restrict__ info) {
    __shared__ double a[NB][NB + 1];
    __shared__ double x[NB][NB + 1];
    __shared__ double dsh;
    const int tid = threadIdx.x;
    const int nb = min(NB, n - k0);
    for (int e = tid; e < NB * NB; e += 256) {
        int r = e / NB, c = e % NB;
        double v = 0.0;
        if (r < nb && c < nb && c <= r) v = A[(size_t)(k0 + r) * n + k0 + c];
        if (r >= nb && r == c) v = 1.0;  // identity padding
        a[r][c] = v;
    }
    __syncthreads();
    for (int j = 0; j < NB; ++j) {
        if (tid == 0) {
            double d = a[j][j];
            if (!(d > 0.0)) {
                if (info[0] == 0) info[0] = k0 + j + 1;
                d = 1.0;
            }
            dsh = sqrt(d);
        }
        __syncthreads();
        const double d = dsh;
        if (tid > j && tid < NB) a[tid][j] /= d;
        if (tid == 0) a[j][j] = d;
        __syncthreads();
        // trailing update of the lower triangle: rows i > j, cols j < c <= i
        const int m = NB - 1 - j;
        for (int e = tid; e < m * m; e += 256) {
            int i = j + 1 + e / m, c = j + 1 + e % m;
            if (c <= i) a[i][c] -= a[i][j] * a[c][j];
        }
        __syncthreads();
    }
    // inverse of the lower-triangular block, one column per thread
    if (tid < NB) {
        const int c = tid;
        for (int i = 0; i < NB; ++i) x[i][c] = 0.0;
        x[c][c] = 1.0 / a[c][c];
        for (int i = c + 1; i < NB; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s += a[i][k] * x[k][c];
            x[i][c] = -s / a[i][i];
        }
    }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) {
        int r = e / NB, c = e % NB;
        if (r < nb && c < nb && c <= r) A[(size_t)(k0 + r) * n + k0 + c] = a[r][c];
        Linv[e] = (c <= r) ? x[r][c] : 0.0;
    }
}

// 64x64 tile product  acc += As (64 x 64, rows) * Bs^T  on f64 MFMA; wave w owns the 32x32 quadrant (w>>1, w&1).
__device__ __forceinline__ void tile_gemm_nt(const double (*As)[LDT], const double (*Bs)[LDT], double4_t (&acc)[2][2]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r0 = (w >> 1) * 32, c0 = (w & 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll 4
    for (int ks = 0; ks < NB / 4; ++ks) {
        const int k = ks * 4 + lk;
        double a0 = As[r0 + lr][k], a1 = As[r0 + 16 + lr][k];
        double b0 = Bs[c0 + lr][k], b1 = Bs[c0 + 16 + lr][k];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}

__device__ __forceinline__ void load_tile(double (*T)[LDT], const double *__restrict__ src, int ld, int rows, int cols) {
    // 64 x 64 doubles, 16 doubles (128 B) per thread-row chunk; zero fill outside (rows, cols)
    for (int e = threadIdx.x; e < NB * (NB / 2); e += 256) {
        int r = e / (NB / 2), c = (e % (NB / 2)) * 2;
        double v0 = 0.0, v1 = 0.0;
        if (r < rows) {
            if (c + 1 < cols) {
                const double2 v = *reinterpret_cast<const double2 *>(src + (size_t)r * ld + c);
                v0 = v.x;
                v1 = v.y;
            } else if (c < cols) {
                v0 = src[(size_t)r * ld + c];
            }
        }
        T[r][c] = v0;
        T[r][c + 1] = v1;
    }
}

// ---- (2) panel: A_ik <- A_ik * Linv^T --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_panel_kernel(double *__restrict__ A, int n, int k0,
                                                         const double *__restrict__ Linv) {
    __shared__ double As[NB][LDT], Bs[NB][LDT];
    const int i0 = k0 + NB + blockIdx.x * NB;
    const int rows = min(NB, n - i0);
    const int kc = min(NB, n - k0);
    double *tile = A + (size_t)i0 * n + k0;
    load_tile(As, tile, n, rows, kc);
    load_tile(Bs, Linv, NB, NB, NB);
    __syncthreads();
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0, 0, 0, 0};
    tile_gemm_nt(As, Bs, acc);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r0 = (w >> 1) * 32, c0 = (w & 1) * 32;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = r0 + a * 16 + (lane >> 4) + 4 * i, c = c0 + b * 16 + (lane & 15);
                if (r < rows && c < kc) tile[(size_t)r * n + c] = acc[a][b][i];
            }
}

// ---- (3) trailing update: A_ij -= A_ik A_jk^T for block pairs i >= j > k ----------------------------------------------
__global__ __launch_bounds__(256) void chol_update_kernel(double *__restrict__ A, int n, int k0) {
    __shared__ double As[NB][LDT], Bs[NB][LDT];
    // decode the lower-triangular pair index
    const int t = blockIdx.x;
    int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    const int i0 = k0 + NB + bi * NB, j0 = k0 + NB + bj * NB;
    const int rows = min(NB, n - i0), cols = min(NB, n - j0);
    const int kc = min(NB, n - k0);
    load_tile(As, A + (size_t)i0 * n + k0, n, rows, kc);
    load_tile(Bs, A + (size_t)j0 * n + k0, n, cols, kc);
    __syncthreads();
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0, 0, 0, 0};
    tile_gemm_nt(As, Bs, acc);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r0 = (w >> 1) * 32, c0 = (w & 1) * 32;
    double *tile = A + (size_t)i0 * n + j0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = r0 + a * 16 + (lane >> 4) + 4 * i, c = c0 + b * 16 + (lane & 15);
                if (r < rows && c < cols && (bi != bj || c <= r)) tile[(size_t)r * n + c] -= acc[a][b][i];
            }
}

// ---- solves -----------------------------------------------------------------------------------------------------------
// y_k = Linv_kk * b_k   (forward)   or   x_k = Linv_kk^T * y_k   (backward); one workgroup, nrhs columns.
__global__ __launch_bounds__(64) void tri_block_kernel(const double *__restrict__ Linv, double *__restrict__ b, int n,
                                                       int nrhs, int k0, int transpose) {
    __shared__ double v[NB];
    const int nb = min(NB, n - k0);
    const int r = threadIdx.x;
    for (int c = 0; c < nrhs; ++c) {
        double *bc = b + (size_t)c * n + k0;
        v[r] = r < nb ? bc[r] : 0.0;
        __syncthreads();
        double s = 0.0;
        if (!transpose) {
            for (int k = 0; k <= r; ++k) s += Linv[r * NB + k] * v[k];
        } else {
            for (int k = r; k < NB; ++k) s += Linv[k * NB + r] * v[k];
        }
        __syncthreads();
        if (r < nb) bc[r] = s;
    }
}

// forward: b_i -= L_ik y_k for the rows below block k; 8 lanes per row, 32 rows per 256-thread workgroup
__global__ __launch_bounds__(256) void fwd_update_kernel(const double *__restrict__ A, double *__restrict__ b, int n,
                                                         int nrhs, int k0) {
    __shared__ double y[NB];
    const int kc = min(NB, n - k0);
    const int row = k0 + NB + blockIdx.x * 32 + (threadIdx.x >> 3);
    const int part = threadIdx.x & 7;
    for (int c = 0; c < nrhs; ++c) {
        __syncthreads();
        if (threadIdx.x < NB) y[threadIdx.x] = threadIdx.x < kc ? b[(size_t)c * n + k0 + threadIdx.x] : 0.0;
        __syncthreads();
        double s = 0.0;
        if (row < n) {
            const double *Lr = A + (size_t)row * n + k0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int k = part * 8 + q;
                if (k < kc) s += Lr[k] * y[k];
            }
        }
        s += __shfl_down(s, 4, 8);
        s += __shfl_down(s, 2, 8);
        s += __shfl_down(s, 1, 8);
        if (row < n && part == 0) b[(size_t)c * n + row] -= s;
    }
}

// backward: y_j -= L_kj^T x_k for the columns left of block k (rows k0..k0+63 of L are contiguous in memory)
__global__ __launch_bounds__(256) void bwd_update_kernel(const double *__restrict__ A, double *__restrict__ b, int n,
                                                         int nrhs, int k0) {
    __shared__ double x[NB];
    const int kr = min(NB, n - k0);
    const int col = blockIdx.x * 256 + threadIdx.x;
    for (int c = 0; c < nrhs; ++c) {
        __syncthreads();
        if (threadIdx.x < NB) x[threadIdx.x] = threadIdx.x < kr ? b[(size_t)c * n + k0 + threadIdx.x] : 0.0;
        __syncthreads();
        if (col < k0) {
            double s = 0.0;
            for (int r = 0; r < kr; ++r) s += A[(size_t)(k0 + r) * n + col] * x[r];
            b[(size_t)c * n + col] -= s;
        }
    }
}

}  // namespace

extern "C" {

size_t mm_chol_workspace_bytes(int n) {
    size_t nblk = (size_t)(n + NB - 1) / NB;
    return mm_align_up(nblk * NB * NB * sizeof(double), 256);
}

int mm_chol_solve(mm_ctx *ctx, double *A, int n, double *b, int nrhs, int32_t *info, void *ws, size_t ws_bytes) {
    if (!ctx) return MM_ERR_ARG;
    if (n == 0) return MM_OK;
    if (!A || !b || !info || n < 0 || nrhs < 0) return mm_fail(ctx, MM_ERR_ARG, "mm_chol_solve: bad argument");
    if (!ws || ws_bytes < mm_chol_workspace_bytes(n)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_chol_solve: workspace too small");
    if (((uintptr_t)A & 15) || (n & 1)) return mm_fail(ctx, MM_ERR_ARG, "mm_chol_solve: A must be 16-byte aligned and n even");
    MM_HIP(ctx, hipMemsetAsync(info, 0, sizeof(int32_t), ctx->stream));
    double *Linv = (double *)ws;
    const int nblk = (n + NB - 1) / NB;
    for (int k = 0; k < nblk; ++k) {
        const int k0 = k * NB;
        double *Lk = Linv + (size_t)k * NB * NB;
        MM_LAUNCH(ctx, "chol_diag_kernel", chol_diag_kernel, dim3(1), dim3(256), 0, A, n, k0, Lk, info);
        const int m = nblk - k - 1;
        if (m > 0) {
            MM_LAUNCH(ctx, "chol_panel_kernel", chol_panel_kernel, dim3(m), dim3(256), 0, A, n, k0, (const double *)Lk);
            MM_LAUNCH(ctx, "chol_update_kernel", chol_update_kernel, dim3(m * (m + 1) / 2), dim3(256), 0, A, n, k0);
        }
    }
    if (nrhs == 0) return MM_OK;
    for (int k = 0; k < nblk; ++k) {  // L y = b
        const int k0 = k * NB;
        MM_LAUNCH(ctx, "tri_block_kernel", tri_block_kernel, dim3(1), dim3(64), 0,
                  (const double *)(Linv + (size_t)k * NB * NB), b, n, nrhs, k0, 0);
        const int below = n - k0 - NB;
        if (below > 0)
            MM_LAUNCH(ctx, "fwd_update_kernel", fwd_update_kernel, dim3((below + 31) / 32), dim3(256), 0,
                      (const double *)A, b, n, nrhs, k0);
    }
    for (int k = nblk - 1; k >= 0; --k) {  // L^T x = y
        const int k0 = k * NB;
        MM_LAUNCH(ctx, "tri_block_kernel", tri_block_kernel, dim3(1), dim3(64), 0,
                  (const double *)(Linv + (size_t)k * NB * NB), b, n, nrhs, k0, 1);
        if (k0 > 0)
            MM_LAUNCH(ctx, "bwd_update_kernel", bwd_update_kernel, dim3((k0 + 255) / 256), dim3(256), 0,
                      (const double *)A, b, n, nrhs, k0);
    }
    return MM_OK;
}

}  // extern "C"
