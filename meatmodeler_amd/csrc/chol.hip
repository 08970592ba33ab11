// SPD solve by blocked right-looking Cholesky, f64, with an optional band (gfx950).
//
// This is the dense linear algebra of the trust-region step: the reduced camera system S (6F x 6F) that the
// reference hands to LSMR implicitly (scipy trf.py:480 via bundleAdjuster.py:180-192) is factored here.
//   step k:  (1) chol_diag   : L_kk = chol(A_kk) and L_kk^-1, ONE wave, the 64x64 block in registers (row per lane),
//                              columns broadcast through LDS; ~2x2016 fully unrolled FMAs per lane
//            (2) chol_panel  : A_ik <- A_ik L_kk^-T          (64x64x64 f64-MFMA GEMM per row block, B operand = L_kk^-1)
//            (3) chol_update : A_ij <- A_ij - A_ik A_jk^T    (64x64x64 f64-MFMA GEMM per block pair i >= j > k)
// Tracks built from consecutive-keyframe matching only connect cameras at most `track length` apart, so S is block
// banded; `half_bandwidth` (A[i][j] == 0 for i - j > half_bandwidth) limits (2) and (3) to the band: n*bw^2 instead
// of n^3/3 flops.  half_bandwidth >= n means dense.
// Solves: one launch per block column and direction; every workgroup recomputes the 64-vector L_kk^-1 b_k (4 kflop)
// instead of waiting for a separate launch.  Row-major A, lower triangle referenced / overwritten.
// MFMA fragment maps for f64 16x16x4 (guide §3): A lane l -> A[l&15][l>>4], B lane l -> B[l>>4][l&15],
// D reg i of lane l -> D[(l>>4) + 4 i][l&15].
#include "mm_common.h"

namespace {

constexpr int NB = 64;
constexpr int LDT = 66;  // LDS tile leading dimension (doubles): 66 keeps the 32-lane ds_read_b64 groups conflict-free
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double readlane_f64(double x, int src_lane /*wave-uniform*/) {
    const unsigned long long u = __double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), src_lane);
    const unsigned hi = __builtin_amdgcn_readlane((int)(u >> 32), src_lane);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// ---- (1) diagonal block ------------------------------------------------------------------------------------------------
// ONE wave; lane = row.  A lone wave is issue-latency bound (~4+ cycles per instruction), so the kernel is written to
// minimise INSTRUCTIONS per FMA: everything is fully unrolled with static register indices, the lane keeps its own row
// of L in registers, other rows come from LDS as same-address (broadcast) reads with immediate offsets, and pivots /
// panel columns travel through v_readlane instead of LDS.  16-column panels: left-looking update of the panel from the
// finished columns (1536 FMAs per lane in total), then the panel is factored in registers (4 x 120 FMAs).  L^-1 by
// forward substitution, lane = column, x in registers (2016 FMAs).  (The rolled LDS-resident version took ~95 us.)
// Panel pb (columns c0 = 16 pb ..): wave w updates rows 16w..16w+15 of the panel from the finished columns with f64
// MFMA (K = c0), then wave 0 factors the 64 x 16 panel in registers (lane = row; pivots and panel columns travel by
// v_readlane).  A lone wave is issue-latency bound, so the point is to minimise instructions per FMA.
template <int PB>
__device__ __forceinline__ void chol_panel16(double (*M)[NB + 1], int k0, int &bad) {
    constexpr int c0 = PB * 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (PB > 0) {
        if (16 * w + 15 >= c0) {  // rows above the panel hold final entries / structural zeros
            double4_t acc = {0, 0, 0, 0};
            const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
            for (int ks = 0; ks < c0 / 4; ++ks) {
                const double av = M[16 * w + lr][4 * ks + lk];
                const double bv = M[c0 + lr][4 * ks + lk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) M[16 * w + (lane >> 4) + 4 * i][c0 + (lane & 15)] -= acc[i];
        }
        __syncthreads();
    }
    if (w == 0) {
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = M[lane][c0 + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            double piv = readlane_f64(a[q], c0 + q);
            const bool ok = piv > 0.0;
            bad = (!ok && bad == 0) ? k0 + c0 + q + 1 : bad;
            piv = ok ? piv : 1.0;
            // 1/sqrt by v_rsq_f64 + two Newton steps (full f64 accuracy): one transcendental and ~8 FMAs per pivot
            // instead of a sqrt and a division (~50 instructions) on the sequential path
            double r = __builtin_amdgcn_rsq(piv);
            r = r * (1.5 - 0.5 * piv * r * r);
            r = r * (1.5 - 0.5 * piv * r * r);
            const double d = piv * r;
            const double lq = lane > c0 + q ? a[q] * r : (lane == c0 + q ? d : 0.0);
            a[q] = lq;
#pragma unroll
            for (int r = q + 1; r < 16; ++r) a[r] -= lq * readlane_f64(lq, c0 + r);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) M[lane][c0 + q] = a[q];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void chol_diag_kernel(double *__restrict__ A, int n, int k0, double *__restrict__ Linv,
                                                        int32_t *__restrict__ info) {
    __shared__ double M[NB][NB + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nb = min(NB, n - k0);
    {   // wave w loads rows 16w..16w+15, lane = column: coalesced rows, 16 loads in flight per lane
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int r = 16 * w + q;
            v[q] = (r < nb && lane < nb && lane <= r) ? A[(size_t)(k0 + r) * n + k0 + lane] : 0.0;
            if (r >= nb && r == lane) v[q] = 1.0;  // identity padding of a partial last block
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) M[16 * w + q][lane] = v[q];
    }
    __syncthreads();
    int bad = 0;
    chol_panel16<0>(M, k0, bad);
    chol_panel16<1>(M, k0, bad);
    chol_panel16<2>(M, k0, bad);
    chol_panel16<3>(M, k0, bad);
    if (bad && threadIdx.x == 0 && info[0] == 0) info[0] = bad;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {  // write L back (posted stores)
        const int r = e / NB, c = e % NB;
        if (r < nb && c < nb && c <= r) A[(size_t)(k0 + r) * n + k0 + c] = M[r][c];
    }
    // ---- L^-1 by 16x16 blocks: the four diagonal blocks are inverted by the four waves in parallel (forward
    // substitution in registers, 120 FMAs), then the off-diagonal blocks level by level on f64 MFMA:
    //   X_ij = -X_ii * sum_{k=j}^{i-1} L_ik X_kj          (i - j = 1, 2, 3)
    // ~5 us instead of ~9.5 us for the 2016-FMA single-wave substitution this replaces.
    __shared__ double X[NB][NB + 1];
    __shared__ double T[4][16][17];
    for (int e = threadIdx.x; e < NB * NB; e += 256) X[e / NB][e % NB] = 0.0;
    __syncthreads();
    if (lane < 16) {
        const int o = 16 * w;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double s = (i == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) s -= M[o + i][o + k] * x[k];
            x[i] = s / M[o + i][o + i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) X[o + i][o + lane] = x[i];   // zero above the diagonal by construction
    }
    __syncthreads();
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int d = 1; d < 4; ++d) {
        const int bi = w + d, bj = w;  // wave w computes block (w + d, w) of this level
        if (bi < 4) {
            double4_t acc = {0, 0, 0, 0};
            for (int bk = bj; bk < bi; ++bk) {
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) {
                    const double av = M[16 * bi + lr][16 * bk + 4 * ss + lk];   // L_ik
                    const double bv = X[16 * bk + 4 * ss + lk][16 * bj + lr];   // X_kj
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) T[w][lk + 4 * i][lr] = acc[i];
            wave_lds_sync();
            double4_t acc2 = {0, 0, 0, 0};
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                const double av = X[16 * bi + lr][16 * bi + 4 * ss + lk];       // X_ii
                const double bv = T[w][4 * ss + lk][lr];
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc2, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) X[16 * bi + lk + 4 * i][16 * bj + lr] = -acc2[i];
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < NB * NB; e += 256) Linv[e] = X[e / NB][e % NB];
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
    // 64 x 64 doubles as 16-byte pieces, 8 per thread; all eight global loads are issued before the first LDS write
    // (a rolled loop with a guarded load per trip serialises eight memory latencies); zero fill outside (rows, cols)
    double2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = threadIdx.x + 256 * q;
        const int r = e / (NB / 2), c = (e % (NB / 2)) * 2;
        double2 t = make_double2(0.0, 0.0);
        if (r < rows) {
            if (c + 1 < cols) {
                t = *reinterpret_cast<const double2 *>(src + (size_t)r * ld + c);
            } else if (c < cols) {
                t.x = src[(size_t)r * ld + c];
            }
        }
        v[q] = t;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = threadIdx.x + 256 * q;
        const int r = e / (NB / 2), c = (e % (NB / 2)) * 2;
        T[r][c] = v[q].x;
        T[r][c + 1] = v[q].y;
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

// ---- (3) trailing update: A_ij -= A_ik A_jk^T for block pairs i >= j > k (inside the band) ------------------------------
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
// z = Linv * v (transpose = 0) or Linv^T * v into LDS vector `out`; 256 threads, 4 per row.
__device__ __forceinline__ void block_gemv64(const double *__restrict__ Linv, const double *vin, double *out,
                                             int transpose) {
    const int r = threadIdx.x >> 2, part = threadIdx.x & 3;
    double s = 0.0;
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        const int k = part * 16 + q;
        s += (transpose ? Linv[k * NB + r] : Linv[r * NB + k]) * vin[k];
    }
    s += __shfl_down(s, 2, 4);
    s += __shfl_down(s, 1, 4);
    if (part == 0) out[r] = s;
}

// forward step k: y_k = Linv_kk b_k (every workgroup, redundantly); workgroup 0 publishes y_k; rows below (inside the
// band) get b_i -= L_ik y_k, 32 rows per workgroup, 8 lanes per row.
__global__ __launch_bounds__(256) void fwd_step_kernel(const double *__restrict__ A, const double *__restrict__ Linv,
                                                       double *__restrict__ b, double *__restrict__ y, int n, int k0,
                                                       int row_end) {
    __shared__ double vin[NB], yk[NB];
    const int kc = min(NB, n - k0);
    if (threadIdx.x < NB) vin[threadIdx.x] = threadIdx.x < kc ? b[k0 + threadIdx.x] : 0.0;
    __syncthreads();
    block_gemv64(Linv, vin, yk, 0);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < kc) y[k0 + threadIdx.x] = yk[threadIdx.x];
    const int row = k0 + NB + blockIdx.x * 32 + (threadIdx.x >> 3);
    const int part = threadIdx.x & 7;
    double s = 0.0;
    if (row < row_end) {
        const double *Lr = A + (size_t)row * n + k0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = part * 8 + q;
            if (k < kc) s += Lr[k] * yk[k];
        }
    }
    s += __shfl_down(s, 4, 8);
    s += __shfl_down(s, 2, 8);
    s += __shfl_down(s, 1, 8);
    if (row < row_end && part == 0) b[row] -= s;
}

// backward step k: x_k = Linv_kk^T y_k (redundantly); workgroup 0 publishes x_k into `x`; columns left of the block
// (inside the band) get y_j -= L_kj^T x_k, one column per thread.
__global__ __launch_bounds__(256) void bwd_step_kernel(const double *__restrict__ A, const double *__restrict__ Linv,
                                                       double *__restrict__ y, double *__restrict__ x, int n, int k0,
                                                       int col_begin) {
    __shared__ double vin[NB], xk[NB];
    const int kr = min(NB, n - k0);
    if (threadIdx.x < NB) vin[threadIdx.x] = threadIdx.x < kr ? y[k0 + threadIdx.x] : 0.0;
    __syncthreads();
    block_gemv64(Linv, vin, xk, 1);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < kr) x[k0 + threadIdx.x] = xk[threadIdx.x];
    // 64 columns per workgroup, 4 row groups of 16: sixteen independent loads per thread instead of a 64-long chain
    __shared__ double part[4][NB];
    const int col = col_begin + blockIdx.x * NB + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    double s = 0.0;
    if (col < k0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int r = rg * 16 + q;
            if (r < kr) s += A[(size_t)(k0 + r) * n + col] * xk[r];
        }
    }
    part[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && col < k0) y[col] -= (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

}  // namespace

extern "C" {

size_t mm_chol_workspace_bytes(int n) {
    size_t nblk = (size_t)(n + NB - 1) / NB;
    return mm_align_up(nblk * NB * NB * sizeof(double), 256) + mm_align_up((size_t)(n + NB) * sizeof(double), 256);
}

int mm_chol_solve(mm_ctx *ctx, double *A, int n, double *b, int nrhs, int half_bandwidth, int32_t *info, void *ws,
                  size_t ws_bytes) {
    if (!ctx) return MM_ERR_ARG;
    if (n == 0) return MM_OK;
    if (!A || !info || n < 0 || nrhs < 0 || (nrhs > 0 && !b) || half_bandwidth < 0)
        return mm_fail(ctx, MM_ERR_ARG, "mm_chol_solve: bad argument");
    if (!ws || ws_bytes < mm_chol_workspace_bytes(n)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_chol_solve: workspace too small");
    if (((uintptr_t)A & 15) || (n & 1)) return mm_fail(ctx, MM_ERR_ARG, "mm_chol_solve: A must be 16-byte aligned and n even");
    MM_HIP(ctx, hipMemsetAsync(info, 0, sizeof(int32_t), ctx->stream));
    const int nblk = (n + NB - 1) / NB;
    double *Linv = (double *)ws;
    double *ytmp = (double *)((char *)ws + mm_align_up((size_t)nblk * NB * NB * sizeof(double), 256));
    // block (bi, bj) can be non-zero iff 64 (bi - bj) - 63 <= half_bandwidth
    long bwb_l = ((long)half_bandwidth + NB - 1) / NB;
    const int bwb = bwb_l > nblk ? nblk : (int)bwb_l;
    for (int k = 0; k < nblk; ++k) {
        const int k0 = k * NB;
        double *Lk = Linv + (size_t)k * NB * NB;
        MM_LAUNCH(ctx, "chol_diag_kernel", chol_diag_kernel, dim3(1), dim3(256), 0, A, n, k0, Lk, info);
        int m = nblk - k - 1;
        if (m > bwb) m = bwb;
        if (m > 0) {
            MM_LAUNCH(ctx, "chol_panel_kernel", chol_panel_kernel, dim3(m), dim3(256), 0, A, n, k0, (const double *)Lk);
            MM_LAUNCH(ctx, "chol_update_kernel", chol_update_kernel, dim3(m * (m + 1) / 2), dim3(256), 0, A, n, k0);
        }
    }
    for (int c = 0; c < nrhs; ++c) {
        double *bc = b + (size_t)c * n;
        for (int k = 0; k < nblk; ++k) {  // L y = b
            const int k0 = k * NB;
            long re = (long)k0 + NB + (long)bwb * NB;
            const int row_end = re > n ? n : (int)re;
            const int below = row_end - (k0 + NB);
            const int grid = below > 0 ? (below + 31) / 32 : 1;
            MM_LAUNCH(ctx, "fwd_step_kernel", fwd_step_kernel, dim3(grid), dim3(256), 0, (const double *)A,
                      (const double *)(Linv + (size_t)k * NB * NB), bc, ytmp, n, k0, row_end);
        }
        for (int k = nblk - 1; k >= 0; --k) {  // L^T x = y
            const int k0 = k * NB;
            long cb = (long)k0 - (long)bwb * NB;
            const int col_begin = cb < 0 ? 0 : (int)cb;
            const int left = k0 - col_begin;
            const int grid = left > 0 ? (left + NB - 1) / NB : 1;
            MM_LAUNCH(ctx, "bwd_step_kernel", bwd_step_kernel, dim3(grid), dim3(256), 0, (const double *)A,
                      (const double *)(Linv + (size_t)k * NB * NB), ytmp, bc, n, k0, col_begin);
        }
    }
    return MM_OK;
}

}  // extern "C"
