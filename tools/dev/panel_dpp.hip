// Probe: 64 x 16 panel factorisation of the banded Cholesky's diagonal block with DPP row broadcasts instead of
// v_readlane + SGPR operands (gfx950).  One wave; lane = row of the 64 x 64 block.  Every 16-lane DPP row also keeps a
// SHADOW copy of the 16 x 16 diagonal block (row t of the block in lane t of each DPP row), so that
//     a[j]  -= bcast_j(slq) * lq        (own row)
//     sh[j] -= bcast_j(slq) * slq       (shadow row)
// are ONE v_fmac_f64_dpp each (row_newbcast:j), against 2 v_readlane_b32 + 1 v_fma_f64 per update today.  Idle lanes
// (rows above the panel) carry identity rows: they come out as X^T = L_dd^-T, i.e. the inverse of the diagonal block for free.
// Build: hipcc -O3 --offload-arch=gfx950 tools/dev/panel_dpp.hip -o tools/dev/panel_dpp ; run: tools/dev/panel_dpp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int J>
__device__ __forceinline__ void fmac_bcast(double &acc, double bsrc, double x) {   // acc -= bcast_J(bsrc) * x
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(x), "n"(J));
}
template <int J>
__device__ __forceinline__ double mov_bcast(double src) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(J));
    return r;
}

template <int Q, int J>
__device__ __forceinline__ void upd(double (&a)[16], double (&sh)[16], double slq, double lq) {
    if constexpr (J < 16) {
        fmac_bcast<J>(sh[J], slq, slq);
        fmac_bcast<J>(a[J], slq, lq);
        upd<Q, J + 1>(a, sh, slq, lq);
    }
}

template <int Q>
__device__ __forceinline__ void column(double (&a)[16], double (&sh)[16], double (&rr)[16]) {
    const double piv = mov_bcast<Q>(sh[Q]);
    double r = __builtin_amdgcn_rsq(piv);
    r = r * (1.5 - 0.5 * piv * r * r);
    r = r * (1.5 - 0.5 * piv * r * r);
    rr[Q] = r;
    const double slq = sh[Q] * r, lq = a[Q] * r;
    sh[Q] = slq;
    a[Q] = lq;
    asm volatile("s_nop 1" : "+v"(sh[Q]), "+v"(a[Q]));   // DPP read of a VGPR written by the previous VALU: 2 wait states
    upd<Q, Q + 1>(a, sh, sh[Q], a[Q]);
}

// LDL^T form of the same panel: the chain per column is  pivot broadcast -> v_rcp_f64 + 2 Newton steps (2 FMAs each) -> one
// multiply -> first update -> next broadcast  (8 dependent instructions instead of 12: no square root on the chain); the
// columns stay unscaled and are multiplied by 1 / sqrt(d_q) once at the end (one v_rsq_f64 for all 16 pivots at once).
template <int Q, int J>
__device__ __forceinline__ void upd_ldl(double (&a)[16], double (&sh)[16], double w) {
    if constexpr (J < 16) {
        fmac_bcast<J>(sh[J], w, sh[Q]);
        fmac_bcast<J>(a[J], w, a[Q]);
        upd_ldl<Q, J + 1>(a, sh, w);
    }
}
template <int Q>
__device__ __forceinline__ void column_ldl(double (&a)[16], double (&sh)[16], double &dvec, int t) {
    const double d = mov_bcast<Q>(sh[Q]);
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    dvec = t == Q ? d : dvec;
    double w = sh[Q] * y;
    asm volatile("s_nop 1" : "+v"(w));   // DPP read of a VGPR written by the previous VALU: 2 wait states
    upd_ldl<Q, Q + 1>(a, sh, w);
}
template <int Q>
__device__ __forceinline__ void scale_col(double (&a)[16], double rvec) {
    a[Q] *= mov_bcast<Q>(rvec);
}

__global__ void panel_ldl_kernel(const double *A, const double *D, int c0, double *Lout, double *Xt, double *R, unsigned long long *ticks,
                                 int reps) {
    const int lane = threadIdx.x, t = lane & 15;
    double a[16], sh[16], dvec = 1.0, rvec = 0.0;
    unsigned long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = lane >= c0 ? (lane >= c0 + q ? A[lane * 16 + q] : 0.0) : (lane < 16 && c0 >= 16 ? (q == lane ? 1.0 : 0.0) : 0.0);
            sh[q] = t >= q ? D[t * 16 + q] : 0.0;
        }
        if (rep == 1) t0 = wall_clock64();
        column_ldl<0>(a, sh, dvec, t);
        column_ldl<1>(a, sh, dvec, t);
        column_ldl<2>(a, sh, dvec, t);
        column_ldl<3>(a, sh, dvec, t);
        column_ldl<4>(a, sh, dvec, t);
        column_ldl<5>(a, sh, dvec, t);
        column_ldl<6>(a, sh, dvec, t);
        column_ldl<7>(a, sh, dvec, t);
        column_ldl<8>(a, sh, dvec, t);
        column_ldl<9>(a, sh, dvec, t);
        column_ldl<10>(a, sh, dvec, t);
        column_ldl<11>(a, sh, dvec, t);
        column_ldl<12>(a, sh, dvec, t);
        column_ldl<13>(a, sh, dvec, t);
        column_ldl<14>(a, sh, dvec, t);
        column_ldl<15>(a, sh, dvec, t);
        rvec = __builtin_amdgcn_rsq(dvec);
        rvec = rvec * (1.5 - 0.5 * dvec * rvec * rvec);
        rvec = rvec * (1.5 - 0.5 * dvec * rvec * rvec);
        scale_col<0>(a, rvec);
        scale_col<1>(a, rvec);
        scale_col<2>(a, rvec);
        scale_col<3>(a, rvec);
        scale_col<4>(a, rvec);
        scale_col<5>(a, rvec);
        scale_col<6>(a, rvec);
        scale_col<7>(a, rvec);
        scale_col<8>(a, rvec);
        scale_col<9>(a, rvec);
        scale_col<10>(a, rvec);
        scale_col<11>(a, rvec);
        scale_col<12>(a, rvec);
        scale_col<13>(a, rvec);
        scale_col<14>(a, rvec);
        scale_col<15>(a, rvec);
        asm volatile("" ::"v"(a[15]), "v"(sh[15]));
    }
    t1 = wall_clock64();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        Lout[lane * 16 + q] = a[q];
        if (lane < 16) Xt[lane * 16 + q] = a[q];
    }
    if (lane < 16) R[lane] = rvec;
    if (lane == 0) ticks[0] = t1 - t0;
}

// A [64][16] panel (row-major, rows >= c0 meaningful), D [16][16] diagonal block; out: L panel [64][16], Xt [16][16] (lane t,
// column q = X[q][t]) from the identity rows in lanes 0..15 (c0 >= 16)
__global__ void panel_kernel(const double *A, const double *D, int c0, double *Lout, double *Xt, double *R, unsigned long long *ticks,
                             int reps) {
    const int lane = threadIdx.x, t = lane & 15;
    double a[16], sh[16], rr[16];
    unsigned long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = lane >= c0 ? (lane >= c0 + q ? A[lane * 16 + q] : 0.0) : (lane < 16 && c0 >= 16 ? (q == lane ? 1.0 : 0.0) : 0.0);
            sh[q] = t >= q ? D[t * 16 + q] : 0.0;
        }
        if (rep == 1) t0 = wall_clock64();
        column<0>(a, sh, rr);
        column<1>(a, sh, rr);
        column<2>(a, sh, rr);
        column<3>(a, sh, rr);
        column<4>(a, sh, rr);
        column<5>(a, sh, rr);
        column<6>(a, sh, rr);
        column<7>(a, sh, rr);
        column<8>(a, sh, rr);
        column<9>(a, sh, rr);
        column<10>(a, sh, rr);
        column<11>(a, sh, rr);
        column<12>(a, sh, rr);
        column<13>(a, sh, rr);
        column<14>(a, sh, rr);
        column<15>(a, sh, rr);
        asm volatile("" ::"v"(a[15]), "v"(sh[15]));
    }
    t1 = wall_clock64();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        Lout[lane * 16 + q] = a[q];
        if (lane < 16) Xt[lane * 16 + q] = a[q];
        if (lane == 0) R[q] = rr[q];
    }
    if (lane == 0) ticks[0] = t1 - t0;
}

int main() {
    const int c0 = 16;
    std::vector<double> A(64 * 16), D(16 * 16), S(48 * 48);
    srand(1);
    // SPD 48 x 48 (rows 16..63 of the block), panel = its first 16 columns
    std::vector<double> G(48 * 48);
    for (auto &v : G) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 48; ++i)
        for (int j = 0; j < 48; ++j) {
            double s = i == j ? 4.0 : 0.0;
            for (int k = 0; k < 48; ++k) s += G[i * 48 + k] * G[j * 48 + k];
            S[i * 48 + j] = s;
        }
    for (int i = 0; i < 64; ++i)
        for (int q = 0; q < 16; ++q) A[i * 16 + q] = i >= c0 ? S[(i - c0) * 48 + q] : 0.0;
    for (int i = 0; i < 16; ++i)
        for (int q = 0; q < 16; ++q) D[i * 16 + q] = S[i * 48 + q];
    // reference: column Cholesky of the panel
    std::vector<double> Lr(A);
    for (int q = 0; q < 16; ++q) {
        const double d = std::sqrt(Lr[(c0 + q) * 16 + q]);
        for (int i = c0 + q; i < 64; ++i) Lr[i * 16 + q] /= d;
        for (int j = q + 1; j < 16; ++j)
            for (int i = c0 + j; i < 64; ++i) Lr[i * 16 + j] -= Lr[i * 16 + q] * Lr[(c0 + j) * 16 + q];
    }
    double *dA, *dD, *dL, *dX, *dR;
    unsigned long long *dT;
    hipMalloc(&dA, A.size() * 8);
    hipMalloc(&dD, D.size() * 8);
    hipMalloc(&dL, 64 * 16 * 8);
    hipMalloc(&dX, 16 * 16 * 8);
    hipMalloc(&dR, 16 * 8);
    hipMalloc(&dT, 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dD, D.data(), D.size() * 8, hipMemcpyHostToDevice);
    const int reps = 101;
    int rc_all = 0;
    for (int variant = 0; variant < 2; ++variant) {
    if (variant == 0)
        hipLaunchKernelGGL(panel_kernel, dim3(1), dim3(64), 0, 0, dA, dD, c0, dL, dX, dR, dT, reps);
    else
        hipLaunchKernelGGL(panel_ldl_kernel, dim3(1), dim3(64), 0, 0, dA, dD, c0, dL, dX, dR, dT, reps);
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("kernel failed\n");
        return 1;
    }
    std::vector<double> L(64 * 16), X(16 * 16);
    unsigned long long ticks;
    hipMemcpy(L.data(), dL, L.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(X.data(), dX, X.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&ticks, dT, 8, hipMemcpyDeviceToHost);
    double err = 0, errx = 0;
    for (int i = c0; i < 64; ++i)
        for (int q = 0; q < 16; ++q)
            if (i >= c0 + q) err = fmax(err, fabs(L[i * 16 + q] - Lr[i * 16 + q]));
    // X^T check: sum_q L[c0+i][q] X[q][t] = delta(i, t), X[q][t] = Xt[t][q]
    for (int i = 0; i < 16; ++i)
        for (int t = 0; t < 16; ++t) {
            double s = 0;
            for (int q = 0; q <= i; ++q) s += Lr[(c0 + i) * 16 + q] * X[t * 16 + q];
            errx = fmax(errx, fabs(s - (i == t ? 1.0 : 0.0)));
        }
    printf("panel_dpp (%s): max |L - ref| = %.3e, max |L X - I| = %.3e, %.3f us per 64x16 panel (100 MHz clock, %d reps)\n",
           variant == 0 ? "L L^T, rsq on the chain" : "L D L^T, rcp on the chain, scaled at the end", err, errx, ticks * 0.01 / (reps - 1), reps - 1);
    if (!(err < 1e-12 && errx < 1e-12)) rc_all = 2;
    }
    return rc_all;
}
