// Bundle-adjustment sweeps (gfx950, f64): residual, analytic Jacobian, block normal equations, J*w,
// Schur complement onto the cameras and back-substitution.
//
// Cost model = reference bundleAdjuster.py:7-52,81-102:
//   theta=|r|, X' = cos X + sin (k x X) + (1-cos)(k.X)k (k=r/theta, 0/0 -> 0), Xc = X'+t, u = K Xc (full 3x3),
//   p = u[:2]/u[2], residual = p - obs, interleaved (x,y).
// The reference differentiates this numerically (scipy 2-point scheme, _numdiff.py:628-705); here the Jacobian is
// analytic.  With a = sin/theta, b = (1-cos)/theta^2, a1 = (theta cos - sin)/theta^3,
// b1 = (theta sin - 2(1-cos))/theta^4 (series below theta = 0.01):
//   X' = cos X + a (r x X) + b (r.X) r
//   dX'/dr_k = r_k (-a X + a1 (r x X) + b1 (r.X) r) + a (e_k x X) + b (X_k r + (r.X) e_k)
//   dX'/dX   = cos I + a [r]x + b r r^T,   dp/dXc = (K[0:2,:] - p K[2,:]) / u2.
//
// Every sweep RECOMPUTES the projection from (cams, pts, obs) instead of materialising J: 48-64 algorithmic bytes
// per observation against ~350 f64 flops — HBM-bound on MI355X (f64 ridge ~12 flop/B).  Reductions are segmented
// (CSR by point / by camera, built once on the host by mm_ba_build_index) and atomic-free, so B, C, g and the cost
// are bitwise reproducible; only the Schur scatter into S uses f64 atomics.
#include "ba_eval.h"

// (Round 4, tried and dropped for the sweeps of this file: the "lean" evaluator of ba_eval.h -- projection and Jacobians
// from per-camera matrices P = K R, q = K t and the right Jacobian J_r, ~64 instead of ~200 f64 instructions per
// observation.  The sweeps are bound by their gathers, not by arithmetic: the lean table is 21 doubles per camera and
// lane against 11 here, and every sweep got SLOWER -- Jacobian product 24 -> 34 us, residual 13 -> 18, back-substitution
// 22 -> 32, point blocks 68 -> 80 us at 1.63 M observations.  The Schur pair kernel, where the two cameras of a chunk
// are wave-uniform scalars, keeps it.)

namespace {

// ---- residual + cost ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ba_residual_body(mm_ba_problem pb, const double *__restrict__ cams,
                                                          const double *__restrict__ pts, const CamCoef *__restrict__ ctab, double *__restrict__ res,
                                                          double *__restrict__ partial, const unsigned bx, const unsigned gx) {
    __shared__ double sm[4];
    __shared__ double Ks[9];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    double acc = 0;
    for (int64_t o = (int64_t)bx * 256 + threadIdx.x; o < pb.O; o += (int64_t)gx * 256) {
        Proj pr;
        const int f = pb.fi[o];
        ba_eval_cc<false, false>(cams + (size_t)f * 6, ctab[f],
                                 pts + (size_t)pb.pi[o] * 3, Ks, pb.obs[2 * o], pb.obs[2 * o + 1], pr);
        if (res) {
            res[2 * o] = pr.r0;
            res[2 * o + 1] = pr.r1;
        }
        acc += pr.r0 * pr.r0 + pr.r1 * pr.r1;
    }
    double t = block_sum<256>(acc, sm);
    if (threadIdx.x == 0) partial[bx] = t;
}
__global__ __launch_bounds__(256) void ba_residual_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                          const double *__restrict__ pts, const CamCoef *__restrict__ ctab, double *__restrict__ res,
                                                          double *__restrict__ partial) { ba_residual_body(pb, cams, pts, ctab, res, partial, blockIdx.x, gridDim.x); }

__global__ __launch_bounds__(256) void sum_partials_kernel(const double *__restrict__ partial, int n,
                                                           double *__restrict__ out) {
    __shared__ double sm[4];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    double t = block_sum<256>(acc, sm);
    if (threadIdx.x == 0) out[0] = t;
}

// sum_partials_kernel (same sum, same order) that also delivers the trust-region driver's scalar board to the host
// mailbox: board[cost_slot] = the sum, then board[0 .. count) and a sequence number (system-scope release) go to pinned
// host memory -- one launch instead of two at the end of every trial step (trf.hip).
struct MMHostBoard {
    double v[16];
    unsigned long long seq;
};
__device__ __forceinline__ void sum_partials_publish_body(const double *__restrict__ partial, int n,
                                                                   double *__restrict__ board, int cost_slot, int count,
                                                                   MMHostBoard *hb, unsigned long long seq, const unsigned bx, const unsigned gx) {
    __shared__ double sm[4];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    const double t = block_sum<256>(acc, sm);
    if (threadIdx.x == 0) board[cost_slot] = t;
    if (threadIdx.x < 64) {      // wave 0 (which holds t in its lane 0)
        const double t0 = __shfl(t, 0, 64);      // (by the whole wave: a shuffle under divergence reads inactive lanes)
        if ((int)threadIdx.x < count) hb->v[threadIdx.x] = (int)threadIdx.x == cost_slot ? t0 : board[threadIdx.x];
        __threadfence_system();
        __builtin_amdgcn_wave_barrier();
        if (threadIdx.x == 0) __hip_atomic_store(&hb->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(256) void sum_partials_publish_kernel(const double *__restrict__ partial, int n,
                                                                   double *__restrict__ board, int cost_slot, int count,
                                                                   MMHostBoard *hb, unsigned long long seq) { sum_partials_publish_body(partial, n, board, cost_slot, count, hb, seq, blockIdx.x, gridDim.x); }

// ---- analytic Jacobian blocks (parity surface) --------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_jacobian_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                          const double *__restrict__ pts, const CamCoef *__restrict__ ctab, double *__restrict__ Jc,
                                                          double *__restrict__ Jp) {
    __shared__ double Ks[9];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= pb.O) return;
    Proj pr;
    const int f = pb.fi[o];
    ba_eval_cc<true, true>(cams + (size_t)f * 6, ctab[f],
                           pts + (size_t)pb.pi[o] * 3, Ks, pb.obs[2 * o], pb.obs[2 * o + 1], pr);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        Jc[o * 12 + k] = pr.Jc[0][k];
        Jc[o * 12 + 6 + k] = pr.Jc[1][k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Jp[o * 6 + k] = pr.Jp[0][k];
        Jp[o * 6 + 3 + k] = pr.Jp[1][k];
    }
}

// ---- point blocks: C[P,6] (upper triangle) and gp[P,3]; FOUR lanes per point over its (contiguous) observations --------
// (One thread per point waits for the waves that own the longest tracks -- 4 observations on average, up to the band width
// (88 at C3): 68 us against 24 us for the Jacobian product over the same observations.  Four lanes stride over a point's
// observations and add up in a fixed butterfly: deterministic, tracks of <= 4 observations in one trip.)
#ifndef MM_PB_LANES
#define MM_PB_LANES 4
#endif
constexpr int PB_LANES = MM_PB_LANES;      // lanes per point in the point-block sweeps (normal equations 84.3 / 81.6 / 98.0 us with 2 / 4 / 8: round 4)
__device__ __forceinline__ void ba_point_blocks_body(mm_ba_problem pb, const double *__restrict__ cams,
                                                              const double *__restrict__ pts, const CamCoef *__restrict__ ctab, double *__restrict__ C,
                                                              double *__restrict__ gp, const unsigned bx, const unsigned gx) {
    __shared__ double Ks[9];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    const int sub = threadIdx.x & (PB_LANES - 1);
    const int p = bx * (256 / PB_LANES) + (threadIdx.x / PB_LANES);
    const bool live = p < pb.P;      // (no early return: the shuffles below are executed by whole waves)
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // c00 c01 c02 c11 c12 c22 g0 g1 g2
    if (live) {
        const double *Xp = pts + (size_t)p * 3;
        const int e_end = pb.pt_ptr[p + 1];
        for (int e = pb.pt_ptr[p] + sub; e < e_end; e += PB_LANES) {
            const int o = pb.pt_obs[e];
            Proj pr;
            const int f = pb.fi[o];
            ba_eval_cc<false, true>(cams + (size_t)f * 6, ctab[f], Xp, Ks, pb.obs[2 * (size_t)o], pb.obs[2 * (size_t)o + 1], pr);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const double j0 = pr.Jp[m][0], j1 = pr.Jp[m][1], j2 = pr.Jp[m][2];
                const double r = m == 0 ? pr.r0 : pr.r1;
                c[0] += j0 * j0; c[1] += j0 * j1; c[2] += j0 * j2;
                c[3] += j1 * j1; c[4] += j1 * j2; c[5] += j2 * j2;
                c[6] += j0 * r; c[7] += j1 * r; c[8] += j2 * r;
            }
        }
    }
#pragma unroll
    for (int off = 1; off < PB_LANES; off <<= 1) {
        double t[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) t[q] = __shfl_xor(c[q], off, 64);
#pragma unroll
        for (int q = 0; q < 9; ++q) c[q] += t[q];      // (a + b and b + a are the same double: all four lanes hold the same sums)
    }
    if (live && sub == 0) {
        double *Cp = C + (size_t)p * 6;
#pragma unroll
        for (int q = 0; q < 6; ++q) Cp[q] = c[q];
        gp[(size_t)p * 3] = c[6]; gp[(size_t)p * 3 + 1] = c[7]; gp[(size_t)p * 3 + 2] = c[8];
    }
}
__global__ __launch_bounds__(256) void ba_point_blocks_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                              const double *__restrict__ pts, const CamCoef *__restrict__ ctab, double *__restrict__ C,
                                                              double *__restrict__ gp) { ba_point_blocks_body(pb, cams, pts, ctab, C, gp, blockIdx.x, gridDim.x); }

// ---- camera blocks: B[F,6,6] and gc[F,6]; one workgroup per camera over its (gathered) observations -----------------
__device__ __forceinline__ void ba_camera_blocks_body(mm_ba_problem pb, const double *__restrict__ cams,
                                                               const double *__restrict__ pts, double *__restrict__ B,
                                                               double *__restrict__ gc, const unsigned bx, const unsigned gx) {
    __shared__ double smn[4 * 27];
    __shared__ double Ks[9];
    __shared__ double cs[6];
    const int f = bx;
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    if (threadIdx.x < 6) cs[threadIdx.x] = cams[(size_t)f * 6 + threadIdx.x];
    __syncthreads();
    const CamCoef ccf = cam_coef_of(cs);
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0;
    for (int e = pb.cam_ptr[f] + threadIdx.x; e < pb.cam_ptr[f + 1]; e += 256) {
        int o = pb.cam_obs[e];
        Proj pr;
        ba_eval_cc<true, false>(cs, ccf, pts + (size_t)pb.pi[o] * 3, Ks, pb.obs[2 * (size_t)o], pb.obs[2 * (size_t)o + 1],
                                pr);
        int t = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = i; j < 6; ++j) {
                acc[t] += pr.Jc[0][i] * pr.Jc[0][j] + pr.Jc[1][i] * pr.Jc[1][j];
                ++t;
            }
            acc[21 + i] += pr.Jc[0][i] * pr.r0 + pr.Jc[1][i] * pr.r1;
        }
    }
    block_sum_n<27, 256>(acc, smn);
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = i; j < 6; ++j) {
                B[(size_t)f * 36 + i * 6 + j] = acc[t];
                B[(size_t)f * 36 + j * 6 + i] = acc[t];
                ++t;
            }
            gc[(size_t)f * 6 + i] = acc[21 + i];
        }
    }
}
__global__ __launch_bounds__(256) void ba_camera_blocks_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                               const double *__restrict__ pts, double *__restrict__ B,
                                                               double *__restrict__ gc) { ba_camera_blocks_body(pb, cams, pts, B, gc, blockIdx.x, gridDim.x); }

// ---- both block sweeps in ONE launch (round 4) ----------------------------------------------------------------------------
// The two are independent and neither fills the chip (one workgroup per camera walking ~3 k observations; four lanes per
// point): one after the other 44 + 43 us at C3, side by side about the longer of the two.  Workgroups [0, F) take the
// cameras (the long-running ones first), the rest the points; bodies unchanged.
__global__ __launch_bounds__(256) void ba_normal_eq_kernel(mm_ba_problem pb, const double *__restrict__ cams, const double *__restrict__ pts,
                                                           const CamCoef *__restrict__ ctab, double *__restrict__ B, double *__restrict__ gc,
                                                           double *__restrict__ C, double *__restrict__ gp) {
    if ((int)blockIdx.x < pb.F) ba_camera_blocks_body(pb, cams, pts, B, gc, blockIdx.x, (unsigned)pb.F);
    else ba_point_blocks_body(pb, cams, pts, ctab, C, gp, blockIdx.x - (unsigned)pb.F, gridDim.x - (unsigned)pb.F);
}

// ---- out = Jc wc[fi] + Jp wp[pi] ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_jvp_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                     const double *__restrict__ pts, const CamCoef *__restrict__ ctab, const double *__restrict__ wc,
                                                     const double *__restrict__ wp, double *__restrict__ out) {
    __shared__ double Ks[9];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= pb.O) return;
    const int f = pb.fi[o], p = pb.pi[o];
    Proj pr;
    ba_eval_cc<true, true>(cams + (size_t)f * 6, ctab[f],
                           pts + (size_t)p * 3, Ks, pb.obs[2 * o], pb.obs[2 * o + 1], pr);
    double y0 = 0, y1 = 0;
    if (wc) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            double w = wc[(size_t)f * 6 + k];
            y0 += pr.Jc[0][k] * w;
            y1 += pr.Jc[1][k] * w;
        }
    }
    if (wp) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double w = wp[(size_t)p * 3 + k];
            y0 += pr.Jp[0][k] * w;
            y1 += pr.Jp[1][k] * w;
        }
    }
    out[2 * o] = y0;
    out[2 * o + 1] = y1;
}

// The same product with its inner products fused in: rows[0] = <out, other> (other == NULL: <out, out>), rows[1] =
// <out, out>, each as {0, total, total} like mm_multi_dot's rows for a residual-space vector.  The trust-region driver
// needs exactly these sums right after each of its two Jacobian products (|J d g_h|^2; <J s2, J d g_h>, |J s2|^2):
// two vector passes less per iteration.  Deterministic: fixed tree per workgroup, then one small launch adds the
// per-workgroup partials in index order.
__device__ __forceinline__ void ba_jvp_dots_body(mm_ba_problem pb, const double *__restrict__ cams,
                                                          const double *__restrict__ pts, const CamCoef *__restrict__ ctab, const double *__restrict__ wc,
                                                          const double *__restrict__ wp, double *__restrict__ out,
                                                          const double *__restrict__ other, double *__restrict__ partial, const unsigned bx, const unsigned gx) {
    __shared__ double Ks[9];
    __shared__ double sm[(256 / 64) * 2];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    // (grid-stride: at most JVP_MAX_WG workgroups, so the follow-up launch adds a few hundred partials, not thousands)
    double acc[2] = {0.0, 0.0};
    for (int64_t o = (int64_t)bx * 256 + threadIdx.x; o < pb.O; o += (int64_t)gx * 256) {
        const int f = pb.fi[o], p = pb.pi[o];
        Proj pr;
        ba_eval_cc<true, true>(cams + (size_t)f * 6, ctab[f],
                               pts + (size_t)p * 3, Ks, pb.obs[2 * o], pb.obs[2 * o + 1], pr);
        double y0 = 0, y1 = 0;
        if (wc) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double w = wc[(size_t)f * 6 + k];
                y0 += pr.Jc[0][k] * w;
                y1 += pr.Jc[1][k] * w;
            }
        }
        if (wp) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double w = wp[(size_t)p * 3 + k];
                y0 += pr.Jp[0][k] * w;
                y1 += pr.Jp[1][k] * w;
            }
        }
        out[2 * o] = y0;
        out[2 * o + 1] = y1;
        const double sq = y0 * y0 + y1 * y1;
        acc[1] += sq;
        acc[0] += other ? y0 * other[2 * o] + y1 * other[2 * o + 1] : sq;
    }
    block_sum_n<2, 256>(acc, sm);
    if (threadIdx.x == 0) {
        partial[2 * (size_t)bx] = acc[0];
        partial[2 * (size_t)bx + 1] = acc[1];
    }
}
__global__ __launch_bounds__(256) void ba_jvp_dots_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                          const double *__restrict__ pts, const CamCoef *__restrict__ ctab, const double *__restrict__ wc,
                                                          const double *__restrict__ wp, double *__restrict__ out,
                                                          const double *__restrict__ other, double *__restrict__ partial) { ba_jvp_dots_body(pb, cams, pts, ctab, wc, wp, out, other, partial, blockIdx.x, gridDim.x); }

__global__ __launch_bounds__(256) void jvp_rows_kernel(const double *__restrict__ partial, unsigned n_wg, double *__restrict__ rows) { (void)jvp_rows_body(partial, n_wg, rows, blockIdx.x, gridDim.x); }

// ---- back-substitution: dp = Cinv (gp - sum_o Jp_o^T (Jc_o dc[f_o])) ------------------------------------------------
__global__ __launch_bounds__(256) void ba_backsub_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                         const double *__restrict__ pts, const CamCoef *__restrict__ ctab,
                                                         const double *__restrict__ Cinv, const double *__restrict__ gp,
                                                         const double *__restrict__ dc, double *__restrict__ dp) {
    __shared__ double Ks[9];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= pb.P) return;
    const double *Xp = pts + (size_t)p * 3;
    double t0 = gp[(size_t)p * 3], t1 = gp[(size_t)p * 3 + 1], t2 = gp[(size_t)p * 3 + 2];
    for (int e = pb.pt_ptr[p]; e < pb.pt_ptr[p + 1]; ++e) {
        int o = pb.pt_obs[e];
        int f = pb.fi[o];
        Proj pr;
        ba_eval_cc<true, true>(cams + (size_t)f * 6, ctab[f], Xp, Ks,
                               pb.obs[2 * (size_t)o], pb.obs[2 * (size_t)o + 1], pr);
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            double d = dc[(size_t)f * 6 + k];
            s0 += pr.Jc[0][k] * d;
            s1 += pr.Jc[1][k] * d;
        }
        t0 -= pr.Jp[0][0] * s0 + pr.Jp[1][0] * s1;
        t1 -= pr.Jp[0][1] * s0 + pr.Jp[1][1] * s1;
        t2 -= pr.Jp[0][2] * s0 + pr.Jp[1][2] * s1;
    }
    const double *c = Cinv + (size_t)p * 6;
    dp[(size_t)p * 3] = c[0] * t0 + c[1] * t1 + c[2] * t2;
    dp[(size_t)p * 3 + 1] = c[1] * t0 + c[3] * t1 + c[4] * t2;
    dp[(size_t)p * 3 + 2] = c[2] * t0 + c[4] * t1 + c[5] * t2;
}

// The same back-substitution with one thread per OBSERVATION: tracks are 4 observations long on average but up to the
// band width (88 at C3), and with one thread per point the whole launch waits for the waves that own the longest ones
// (85 us at C3 against 25 us for the Jacobian product over the same observations).  Pass 1 writes every observation's
// 3-vector Jp^T (Jc dc) into a scratch array in CSR-by-point order, pass 2 adds each point's rows in that order (a fixed
// order: deterministic) and applies Cinv.
__device__ __forceinline__ void ba_backsub_obs_body(mm_ba_problem pb, const double *__restrict__ cams,
                                                             const double *__restrict__ pts, const CamCoef *__restrict__ ctab,
                                                             const double *__restrict__ dc, double *__restrict__ T, const unsigned bx, const unsigned gx) {
    __shared__ double Ks[9];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    const int64_t e = (int64_t)bx * 256 + threadIdx.x;
    if (e >= pb.O) return;
    const int o = pb.pt_obs[e];
    const int f = pb.fi[o], p = pb.pi[o];
    Proj pr;
    ba_eval_cc<true, true>(cams + (size_t)f * 6, ctab[f], pts + (size_t)p * 3, Ks, pb.obs[2 * (size_t)o],
                           pb.obs[2 * (size_t)o + 1], pr);
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double d = dc[(size_t)f * 6 + k];
        s0 += pr.Jc[0][k] * d;
        s1 += pr.Jc[1][k] * d;
    }
    T[3 * e] = pr.Jp[0][0] * s0 + pr.Jp[1][0] * s1;
    T[3 * e + 1] = pr.Jp[0][1] * s0 + pr.Jp[1][1] * s1;
    T[3 * e + 2] = pr.Jp[0][2] * s0 + pr.Jp[1][2] * s1;
}
__global__ __launch_bounds__(256) void ba_backsub_obs_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                             const double *__restrict__ pts, const CamCoef *__restrict__ ctab,
                                                             const double *__restrict__ dc, double *__restrict__ T) { ba_backsub_obs_body(pb, cams, pts, ctab, dc, T, blockIdx.x, gridDim.x); }

// (four lanes per point like the point blocks: rows of T that belong to one point are read by neighbouring lanes, and the
// launch no longer waits for the threads that own the longest tracks; fixed butterfly: deterministic)
__device__ __forceinline__ void ba_backsub_points_body(mm_ba_problem pb, const double *__restrict__ T,
                                                                const double *__restrict__ Cinv, const double *__restrict__ gp,
                                                                double *__restrict__ dp, const unsigned bx, const unsigned gx) {
    const int sub = threadIdx.x & (PB_LANES - 1);
    const int p = bx * (256 / PB_LANES) + (threadIdx.x / PB_LANES);
    const bool live = p < pb.P;      // (no early return: whole waves execute the shuffles)
    double t[3] = {0.0, 0.0, 0.0};
    if (live) {
        const int e_end = pb.pt_ptr[p + 1];
        for (int e = pb.pt_ptr[p] + sub; e < e_end; e += PB_LANES) {
            t[0] += T[3 * (size_t)e];
            t[1] += T[3 * (size_t)e + 1];
            t[2] += T[3 * (size_t)e + 2];
        }
    }
#pragma unroll
    for (int off = 1; off < PB_LANES; off <<= 1) {
        double u[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) u[q] = __shfl_xor(t[q], off, 64);
#pragma unroll
        for (int q = 0; q < 3; ++q) t[q] += u[q];
    }
    if (live && sub == 0) {
        const double t0 = gp[(size_t)p * 3] - t[0], t1 = gp[(size_t)p * 3 + 1] - t[1], t2 = gp[(size_t)p * 3 + 2] - t[2];
        const double *c = Cinv + (size_t)p * 6;
        dp[(size_t)p * 3] = c[0] * t0 + c[1] * t1 + c[2] * t2;
        dp[(size_t)p * 3 + 1] = c[1] * t0 + c[3] * t1 + c[4] * t2;
        dp[(size_t)p * 3 + 2] = c[2] * t0 + c[4] * t1 + c[5] * t2;
    }
}
__global__ __launch_bounds__(256) void ba_backsub_points_kernel(mm_ba_problem pb, const double *__restrict__ T,
                                                                const double *__restrict__ Cinv, const double *__restrict__ gp,
                                                                double *__restrict__ dp) { ba_backsub_points_body(pb, T, Cinv, gp, dp, blockIdx.x, gridDim.x); }

// ---- rotation coefficients of every camera, once per parameter vector ------------------------------------------------
// (Every workgroup of every sweep used to fill its own LDS copy: two sincos per thread -- more arithmetic than the
// observations it then processed -- and 40 KB of LDS.  Now: one small launch per NEW camera vector, gathers from L2.)
__device__ __forceinline__ void cam_coef_body(const double *__restrict__ cams, int F, CamCoef *__restrict__ tab, const unsigned bx, const unsigned gx) {
    const int f = bx * 256 + threadIdx.x;
    if (f < F) tab[f] = cam_coef_of(cams + (size_t)f * 6);
}
__global__ __launch_bounds__(256) void cam_coef_kernel(const double *__restrict__ cams, int F, CamCoef *__restrict__ tab) { cam_coef_body(cams, F, tab, blockIdx.x, gridDim.x); }

int check_pb(mm_ctx *ctx, const mm_ba_problem *pb, const char *who) {
    if (!ctx) return MM_ERR_ARG;
    if (!pb || pb->F < 0 || pb->P < 0 || pb->O < 0 || !pb->K) return mm_fail(ctx, MM_ERR_ARG, "%s: bad problem", who);
    if (pb->O > 0 && (!pb->fi || !pb->pi || !pb->obs)) return mm_fail(ctx, MM_ERR_ARG, "%s: null observation arrays", who);
    return MM_OK;
}

constexpr int RES_BLOCKS = 2048;
constexpr int JVP_MAX_WG = 2048;      // workgroups of the Jacobian product with fused inner products (grid-stride beyond)
// (the caps actually used: MM_RES_WG / MM_JVP_WG, at most the compile-time capacities above -- for sweeps)
inline int res_wg_cap() {
    static const int cap = [] {
        const char *e = getenv("MM_RES_WG");
        const int v = e ? atoi(e) : RES_BLOCKS;
        return v < 1 ? 1 : (v > RES_BLOCKS ? RES_BLOCKS : v);
    }();
    return cap;
}
inline int jvp_wg_cap() {
    static const int cap = [] {
        const char *e = getenv("MM_JVP_WG");
        const int v = e ? atoi(e) : 1024;      // (25.0 us per product at the bench shape; 2048: 26.5, 512: 31.9 -- round 4)
        return v < 1 ? 1 : (v > JVP_MAX_WG ? JVP_MAX_WG : v);
    }();
    return cap;
}

// Regulariser of the 2-D subspace trust-region step, on the device so that the host does not have to wait for the
// three scalars before the reduced system can be built (SciPy trf.py:473-477 via least_squares(method='trf',
// tr_solver='lsmr'), reference bundleAdjuster.py:180-192): with a = 0.5 |J_h g_h|^2, b = -|g_h|^2 minimise
// t (a t + b) over [0, Delta / |g_h|]; reg = -min / Delta^2.  out = {reg, max(reg, min_damping)}.
__device__ __forceinline__ void trf_damping_body(const double *__restrict__ gh2, const double *__restrict__ d11, double Delta,
                                                 double min_damping, double *__restrict__ out) {
    if (threadIdx.x != 0) return;
    const double reg = trf_damping_value(gh2[0], d11[0], Delta);
    out[0] = reg;
    out[1] = fmax(reg, min_damping);
}
__global__ void trf_damping_kernel(const double *__restrict__ gh2, const double *__restrict__ d11, double Delta,
                                   double min_damping, double *__restrict__ out) {
    if (blockIdx.x != 0) return;
    trf_damping_body(gh2, d11, Delta, min_damping, out);
}

// ---- batched wrappers (mm_ba_trf_batched): blockIdx.y picks the problem, bodies as above with the problem's own grid ------
#define MM_BATCH_PROB(gfield)                                   \
    const int pid = list[blockIdx.y];                           \
    const mm_batch_prob &bp = tab[pid];                         \
    if (blockIdx.x >= bp.gfield) return
__global__ __launch_bounds__(256) void cam_coef_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list, int which) {
    MM_BATCH_PROB(g_coef);
    cam_coef_body(which ? bp.x_new : bp.x, bp.pb.F, (CamCoef *)(which ? bp.ctab_new : bp.ctab_x), blockIdx.x, bp.g_coef);
}
__global__ __launch_bounds__(256) void ba_residual_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    MM_BATCH_PROB(g_res);
    ba_residual_body(bp.pb, bp.x_new, bp.x_new + bp.nc, (const CamCoef *)bp.ctab_new, nullptr, bp.res_partial, blockIdx.x, bp.g_res);
}
__global__ __launch_bounds__(256) void sum_partials_publish_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list,
                                                                         const mm_batch_dyn *__restrict__ dyn) {
    const int pid = list[blockIdx.x];
    const mm_batch_prob &bp = tab[pid];
    sum_partials_publish_body(bp.res_partial, (int)bp.g_res, bp.board, 14, 16, (MMHostBoard *)bp.mailbox, dyn[pid].seq, 0, 1);
}
// rows (<= 16 doubles at `src_off` doubles behind r0 ... here: r0 itself) of the listed problems into their mailboxes
__global__ __launch_bounds__(64) void publish_r0_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list,
                                                              const mm_batch_dyn *__restrict__ dyn) {
    const int pid = list[blockIdx.x];
    const mm_batch_prob &bp = tab[pid];
    MMHostBoard *hb = (MMHostBoard *)bp.mailbox;
    if (threadIdx.x < 6) hb->v[threadIdx.x] = bp.r0[threadIdx.x];
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (threadIdx.x == 0) __hip_atomic_store(&hb->seq, dyn[pid].seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(256) void ba_jvp_dots_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list, int second) {
    MM_BATCH_PROB(g_jvp);
    // first product of an iteration: u1 = J (d g_h); second: Jq2 = J s2 with <Jq2, u1>
    const double *w = second ? bp.s2 : bp.ghs;
    ba_jvp_dots_body(bp.pb, bp.x, bp.x + bp.nc, (const CamCoef *)bp.ctab_x, w, w + bp.nc, second ? bp.Jq2 : bp.u1,
                     second ? bp.u1 : nullptr, bp.jvp_partial, blockIdx.x, bp.g_jvp);
}
__global__ __launch_bounds__(256) void jvp_rows_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list, int second) {
    const mm_batch_prob &bp = tab[list[blockIdx.x]];
    jvp_rows_body(bp.jvp_partial, bp.g_jvp, second ? bp.bs : bp.d11, 0, 1);
}
__global__ void trf_damping_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list,
                                         const mm_batch_dyn *__restrict__ dyn) {
    const int pid = list[blockIdx.x];
    const mm_batch_prob &bp = tab[pid];
    trf_damping_body(bp.r0 + 2, bp.d11 + 2, dyn[pid].Delta, dyn[pid].min_damping, bp.damp);
}
// a retry of the reduced solve with raised damping: damp[1] <- the host's value (mm_ba_trf writes it with a copy)
__global__ void set_reg_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list,
                                     const mm_batch_dyn *__restrict__ dyn) {
    const int pid = list[blockIdx.x];
    if (threadIdx.x == 0) tab[pid].damp[1] = dyn[pid].reg;
}
__global__ __launch_bounds__(256) void ba_backsub_obs_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    MM_BATCH_PROB(g_obs);
    ba_backsub_obs_body(bp.pb, bp.x, bp.x + bp.nc, (const CamCoef *)bp.ctab_x, bp.v, bp.backsub_T, blockIdx.x, bp.g_obs);
}
__global__ __launch_bounds__(256) void ba_backsub_points_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    MM_BATCH_PROB(g_pts);
    ba_backsub_points_body(bp.pb, bp.backsub_T, bp.Cinv, bp.g + bp.nc, bp.dp, blockIdx.x, bp.g_pts);
}
__global__ __launch_bounds__(256) void ba_normal_eq_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    const mm_batch_prob &bp = tab[list[blockIdx.y]];
    const unsigned F = (unsigned)bp.pb.F;
    if (blockIdx.x < F) ba_camera_blocks_body(bp.pb, bp.x, bp.x + bp.nc, bp.B, bp.g, blockIdx.x, F);
    else if (blockIdx.x - F < bp.g_pblk)
        ba_point_blocks_body(bp.pb, bp.x, bp.x + bp.nc, (const CamCoef *)bp.ctab_x, bp.C, bp.g + bp.nc, blockIdx.x - F, bp.g_pblk);
}

}  // namespace

// The table lives in the context.  `cam_tab_for` remembers which camera vector it was computed from; a caller that
// knows the vector has not changed since (the library's own trust-region loop, trf.hip) keeps it valid with
// mm_cam_table_hold() and the sweeps then skip the launch.  Everybody else gets a fresh table per call.
int mm_cam_coef_table(mm_ctx *ctx, const double *cams, int F, const void **tab_out) {
    if (F > ctx->cam_tab_cap) {
        if (ctx->cam_tab) (void)hipFree(ctx->cam_tab);
        ctx->cam_tab = nullptr;
        ctx->cam_tab_cap = 0;
        const int cap = F < 1024 ? 1024 : F + F / 2;
        MM_HIP(ctx, hipMalloc(&ctx->cam_tab, (size_t)cap * sizeof(CamCoef)));
        ctx->cam_tab_cap = cap;
        ctx->cam_tab_for = nullptr;
    }
    if (!(ctx->cam_tab_hold && ctx->cam_tab_for == cams && ctx->cam_tab_F == F)) {
        if (F > 0)
            MM_LAUNCH(ctx, "cam_coef_kernel", cam_coef_kernel, dim3((F + 255) / 256), dim3(256), 0, cams, F, (CamCoef *)ctx->cam_tab);
        ctx->cam_tab_for = cams;
        ctx->cam_tab_F = F;
    }
    *tab_out = ctx->cam_tab;
    return MM_OK;
}
int mm_cam_table_adopt(mm_ctx *ctx, const double *cams, int F, void **tab_out) {
    if (F > ctx->cam_tab_cap) {
        if (ctx->cam_tab) (void)hipFree(ctx->cam_tab);
        ctx->cam_tab = nullptr;
        ctx->cam_tab_cap = 0;
        const int cap = F < 1024 ? 1024 : F + F / 2;
        MM_HIP(ctx, hipMalloc(&ctx->cam_tab, (size_t)cap * sizeof(CamCoef)));
        ctx->cam_tab_cap = cap;
    }
    ctx->cam_tab_for = cams;
    ctx->cam_tab_F = F;
    *tab_out = ctx->cam_tab;
    return MM_OK;
}
void mm_cam_table_hold(mm_ctx *ctx, bool on) {
    ctx->cam_tab_hold = on;
    ctx->cam_tab_for = nullptr;      // (also when switching ON: an earlier, un-held call may have left the address of a
}                                    // buffer that has been reused for other camera values since)
void mm_cam_table_invalidate(mm_ctx *ctx) { ctx->cam_tab_for = nullptr; }

// mm_ba_residual whose final sum lands in board[cost_slot] AND in the host mailbox (see sum_partials_publish_kernel)
int mm_ba_residual_publish(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, void *ws,
                           size_t ws_bytes, double *board, int cost_slot, int count, void *host_board, unsigned long long seq);

#define MM_CAM_TABLE(ctx, pb, cams)                                               \
    const CamCoef *ctab = nullptr;                                                \
    do {                                                                          \
        const void *t_ = nullptr;                                                 \
        int rc_ = mm_cam_coef_table(ctx, cams, (pb)->F, &t_);                     \
        if (rc_) return rc_;                                                      \
        ctab = (const CamCoef *)t_;                                               \
    } while (0)

extern "C" {

int mm_ba_residual(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, double *res,
                   double *cost2, void *ws, size_t ws_bytes) {
    int rc = check_pb(ctx, pb, "mm_ba_residual");
    if (rc) return rc;
    if (!cams || !pts || !cost2) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_residual: null pointer");
    if (!ws || ws_bytes < RES_BLOCKS * sizeof(double)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_ba_residual: workspace < 16 KiB");
    int64_t nb = (pb->O + 255) / 256;
    int blocks = (int)(nb < 1 ? 1 : (nb > res_wg_cap() ? res_wg_cap() : nb));
    MM_CAM_TABLE(ctx, pb, cams);
    MM_LAUNCH(ctx, "ba_residual_kernel", ba_residual_kernel, dim3(blocks), dim3(256), 0, *pb, cams, pts, ctab, res, (double *)ws);
    MM_LAUNCH(ctx, "sum_partials_kernel", sum_partials_kernel, dim3(1), dim3(256), 0, (const double *)ws, blocks, cost2);
    return MM_OK;
}

int mm_ba_jacobian(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, double *Jc, double *Jp) {
    int rc = check_pb(ctx, pb, "mm_ba_jacobian");
    if (rc) return rc;
    if (!cams || !pts || !Jc || !Jp) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_jacobian: null pointer");
    if (pb->O == 0) return MM_OK;
    MM_CAM_TABLE(ctx, pb, cams);
    MM_LAUNCH(ctx, "ba_jacobian_kernel", ba_jacobian_kernel, dim3((unsigned)((pb->O + 255) / 256)), dim3(256), 0, *pb, cams, pts, ctab, Jc, Jp);
    return MM_OK;
}

int mm_ba_normal_eq(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, double *B, double *gc,
                    double *C, double *gp) {
    int rc = check_pb(ctx, pb, "mm_ba_normal_eq");
    if (rc) return rc;
    if (!cams || !pts) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_normal_eq: null pointer");
    if ((B == nullptr) != (gc == nullptr) || (C == nullptr) != (gp == nullptr))
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_normal_eq: B/gc and C/gp come in pairs");
    if (B && C && pb->P > 0 && pb->F > 0) {      // both sweeps side by side in one launch
        if (!pb->pt_ptr || !pb->pt_obs || !pb->cam_ptr || !pb->cam_obs) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_normal_eq: CSR missing");
        MM_CAM_TABLE(ctx, pb, cams);
        const unsigned g_pblk = (unsigned)((pb->P + 256 / PB_LANES - 1) / (256 / PB_LANES));
        MM_LAUNCH(ctx, "ba_normal_eq_kernel", ba_normal_eq_kernel, dim3((unsigned)pb->F + g_pblk), dim3(256), 0, *pb, cams, pts, ctab, B, gc, C, gp);
        return MM_OK;
    }
    if (C) {
        if (!pb->pt_ptr || !pb->pt_obs) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_normal_eq: point CSR missing");
        if (pb->P > 0) {
            MM_CAM_TABLE(ctx, pb, cams);
            MM_LAUNCH(ctx, "ba_point_blocks_kernel", ba_point_blocks_kernel, dim3((pb->P + 256 / PB_LANES - 1) / (256 / PB_LANES)), dim3(256), 0,
                      *pb, cams, pts, ctab, C, gp);
        }
    }
    if (B) {
        if (!pb->cam_ptr || !pb->cam_obs) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_normal_eq: camera CSR missing");
        if (pb->F > 0) {
            MM_LAUNCH(ctx, "ba_camera_blocks_kernel", ba_camera_blocks_kernel, dim3(pb->F), dim3(256), 0, *pb, cams, pts, B, gc);
        }
    }
    return MM_OK;
}

int mm_ba_jvp(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const double *wc,
              const double *wp, double *out) {
    int rc = check_pb(ctx, pb, "mm_ba_jvp");
    if (rc) return rc;
    if (!cams || !pts || !out) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_jvp: null pointer");
    if (pb->O == 0) return MM_OK;
    MM_CAM_TABLE(ctx, pb, cams);
    MM_LAUNCH(ctx, "ba_jvp_kernel", ba_jvp_kernel, dim3((unsigned)((pb->O + 255) / 256)), dim3(256), 0, *pb, cams, pts, ctab, wc, wp, out);
    return MM_OK;
}

size_t mm_ba_jvp_dots_workspace_bytes(const mm_ba_problem *pb) {
    if (!pb || pb->O <= 0) return 256;
    return 256 + mm_align_up((size_t)((pb->O + 255) / 256) * 2 * sizeof(double), 256);
}

int mm_ba_jvp_dots(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const double *wc,
                   const double *wp, double *out, const double *other, double *rows, void *ws, size_t ws_bytes) {
    int rc = check_pb(ctx, pb, "mm_ba_jvp_dots");
    if (rc) return rc;
    if (!cams || !pts || !out || !rows) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_jvp_dots: null pointer");
    if (!ws || ws_bytes < mm_ba_jvp_dots_workspace_bytes(pb) || ((uintptr_t)ws & 255))
        return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_ba_jvp_dots: workspace too small or misaligned");
    if (pb->O == 0) {
        MM_HIP(ctx, hipMemsetAsync(rows, 0, 6 * sizeof(double), ctx->stream));
        return MM_OK;
    }
    const int64_t n_all = (pb->O + 255) / 256;
    const unsigned n_wg = (unsigned)(n_all < jvp_wg_cap() ? n_all : jvp_wg_cap());
    MM_CAM_TABLE(ctx, pb, cams);
    MM_LAUNCH(ctx, "ba_jvp_kernel", ba_jvp_dots_kernel, dim3(n_wg), dim3(256), 0, *pb, cams, pts, ctab, wc, wp, out, other,
              (double *)((char *)ws + 256));
    if (ctx->jvp_rows_deferred) {      // (mm_ba_trf: the consumer's own kernel adds the partials -- mm_trf_rows_step2d)
        ctx->jvp_partial = (const double *)((char *)ws + 256);
        ctx->jvp_n_wg = n_wg;
        return MM_OK;
    }
    MM_LAUNCH(ctx, "jvp_rows_kernel", jvp_rows_kernel, dim3(1), dim3(256), 0, (const double *)((char *)ws + 256), n_wg, rows);
    return MM_OK;
}

size_t mm_ba_backsub_workspace_bytes(const mm_ba_problem *pb) {
    if (!pb || pb->O <= 0) return 256;
    return mm_align_up((size_t)pb->O * 3 * sizeof(double), 256);
}

int mm_ba_backsub(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const double *Cinv,
                  const double *gp, const double *dc, double *dp, void *ws, size_t ws_bytes) {
    int rc = check_pb(ctx, pb, "mm_ba_backsub");
    if (rc) return rc;
    if (!cams || !pts || !Cinv || !gp || !dc || !dp || !pb->pt_ptr || !pb->pt_obs)
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_backsub: null pointer");
    if (ws && ws_bytes < mm_ba_backsub_workspace_bytes(pb)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_ba_backsub: workspace too small");
    if (pb->P == 0) return MM_OK;
    MM_CAM_TABLE(ctx, pb, cams);
    if (ws && pb->O > 0) {
        MM_LAUNCH(ctx, "ba_backsub_kernel", ba_backsub_obs_kernel, dim3((unsigned)((pb->O + 255) / 256)), dim3(256), 0, *pb, cams, pts,
                  ctab, dc, (double *)ws);
        MM_LAUNCH(ctx, "ba_backsub_points_kernel", ba_backsub_points_kernel, dim3((pb->P + 256 / PB_LANES - 1) / (256 / PB_LANES)), dim3(256), 0,
                  *pb, (const double *)ws, Cinv, gp, dp);
        return MM_OK;
    }
    MM_LAUNCH(ctx, "ba_backsub_kernel", ba_backsub_kernel, dim3((pb->P + 255) / 256), dim3(256), 0, *pb, cams, pts, ctab, Cinv, gp, dc, dp);
    return MM_OK;
}

int mm_trf_damping(mm_ctx *ctx, const double *gh2, const double *d11, double Delta, double min_damping, double *out) {
    if (!ctx) return MM_ERR_ARG;
    if (!gh2 || !d11 || !out || !(Delta > 0.0)) return mm_fail(ctx, MM_ERR_ARG, "mm_trf_damping: bad argument");
    MM_LAUNCH(ctx, "trf_damping_kernel", trf_damping_kernel, dim3(1), dim3(64), 0, gh2, d11, Delta, min_damping, out);
    return MM_OK;
}

}  // extern "C"

int mm_ba_residual_publish(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, void *ws,
                           size_t ws_bytes, double *board, int cost_slot, int count, void *host_board, unsigned long long seq) {
    int rc = check_pb(ctx, pb, "mm_ba_residual");
    if (rc) return rc;
    if (!cams || !pts || !board || !host_board || count > 16 || cost_slot >= count) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_residual_publish: bad argument");
    if (!ws || ws_bytes < RES_BLOCKS * sizeof(double)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_ba_residual: workspace < 16 KiB");
    int64_t nb = (pb->O + 255) / 256;
    int blocks = (int)(nb < 1 ? 1 : (nb > res_wg_cap() ? res_wg_cap() : nb));
    MM_CAM_TABLE(ctx, pb, cams);
    MM_LAUNCH(ctx, "ba_residual_kernel", ba_residual_kernel, dim3(blocks), dim3(256), 0, *pb, cams, pts, ctab, (double *)nullptr, (double *)ws);
    MM_LAUNCH(ctx, "sum_partials_kernel", sum_partials_publish_kernel, dim3(1), dim3(256), 0, (const double *)ws, blocks, board,
              cost_slot, count, (MMHostBoard *)host_board, seq);
    return MM_OK;
}


// ---- batched launches of the sweeps (mm_ba_trf_batched, trf.hip) -------------------------------------------------------------
void mm_batch_ba_setup(mm_batch_prob *bp) {
    const mm_ba_problem &pb = bp->pb;
    const int64_t nb = (pb.O + 255) / 256;
    bp->g_res = (uint32_t)(nb < 1 ? 1 : (nb > res_wg_cap() ? res_wg_cap() : nb));
    bp->g_jvp = (uint32_t)(nb < jvp_wg_cap() ? (nb < 1 ? 1 : nb) : jvp_wg_cap());
    bp->g_pblk = (uint32_t)((pb.P + 256 / PB_LANES - 1) / (256 / PB_LANES));
    bp->g_obs = (uint32_t)(nb < 1 ? 1 : nb);
    bp->g_pts = (uint32_t)((pb.P + 256 / PB_LANES - 1) / (256 / PB_LANES));
    bp->g_coef = (uint32_t)((pb.F + 255) / 256);
}
int mm_batch_cam_coef(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g, int which) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "cam_coef_kernel", cam_coef_batch_kernel, dim3(max_g, (unsigned)n_list), dim3(256), 0, tab, list, which);
    return MM_OK;
}
int mm_batch_jvp_dots(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g, int second) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "ba_jvp_kernel", ba_jvp_dots_batch_kernel, dim3(max_g, (unsigned)n_list), dim3(256), 0, tab, list, second);
    MM_LAUNCH(ctx, "jvp_rows_kernel", jvp_rows_batch_kernel, dim3((unsigned)n_list), dim3(256), 0, tab, list, second);
    return MM_OK;
}
int mm_batch_damping(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, const mm_batch_dyn *dyn) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "trf_damping_kernel", trf_damping_batch_kernel, dim3((unsigned)n_list), dim3(64), 0, tab, list, dyn);
    return MM_OK;
}
int mm_batch_set_reg(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, const mm_batch_dyn *dyn) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "set_reg_kernel", set_reg_batch_kernel, dim3((unsigned)n_list), dim3(64), 0, tab, list, dyn);
    return MM_OK;
}
int mm_batch_backsub(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g_obs, unsigned max_g_pts) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "ba_backsub_kernel", ba_backsub_obs_batch_kernel, dim3(max_g_obs, (unsigned)n_list), dim3(256), 0, tab, list);
    MM_LAUNCH(ctx, "ba_backsub_points_kernel", ba_backsub_points_batch_kernel, dim3(max_g_pts, (unsigned)n_list), dim3(256), 0, tab, list);
    return MM_OK;
}
int mm_batch_residual_publish(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g,
                              const mm_batch_dyn *dyn) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "ba_residual_kernel", ba_residual_batch_kernel, dim3(max_g, (unsigned)n_list), dim3(256), 0, tab, list);
    MM_LAUNCH(ctx, "sum_partials_kernel", sum_partials_publish_batch_kernel, dim3((unsigned)n_list), dim3(256), 0, tab, list, dyn);
    return MM_OK;
}
int mm_batch_normal_eq(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g_pblk, unsigned max_F) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "ba_normal_eq_kernel", ba_normal_eq_batch_kernel, dim3(max_F + max_g_pblk, (unsigned)n_list), dim3(256), 0, tab, list);
    return MM_OK;
}
int mm_batch_publish_rows(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, const mm_batch_dyn *dyn) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "board_publish_kernel", publish_r0_batch_kernel, dim3((unsigned)n_list), dim3(64), 0, tab, list, dyn);
    return MM_OK;
}
