// Shared device code of the BA sweeps: the projection / analytic-Jacobian evaluator and block reductions.
// (see ba.hip for the derivation; reference cost model bundleAdjuster.py:7-52,81-102)
#pragma once
#include "mm_common.h"

namespace {

struct Proj {
    double r0, r1;          // residual
    double Jc[2][6];        // d r / d (rvec, tvec)
    double Jp[2][3];        // d r / d X
};

// Rotation coefficients of one camera: they depend on the camera only, so the sweeps take them from a table computed
// once per camera vector (ba.hip: cam_coef_kernel) instead of paying a sincos, a sqrt and four divisions per observation.
struct CamCoef {
    double c, a, b, a1, b1;  // cos, sin/th, (1-cos)/th^2, (th cos - sin)/th^3, (th sin - 2(1-cos))/th^4
};

__device__ __forceinline__ CamCoef cam_coef_of(const double *__restrict__ cam) {
    const double rx = cam[0], ry = cam[1], rz = cam[2];
    const double th2 = rx * rx + ry * ry + rz * rz;
    CamCoef k;
    if (th2 < 1e-4) {
        k.c = cos(sqrt(th2));
        k.a = 1.0 + th2 * (-1.0 / 6 + th2 * (1.0 / 120 - th2 * (1.0 / 5040)));
        k.b = 0.5 + th2 * (-1.0 / 24 + th2 * (1.0 / 720 - th2 * (1.0 / 40320)));
        k.a1 = -1.0 / 3 + th2 * (1.0 / 30 + th2 * (-1.0 / 840 + th2 * (1.0 / 45360)));
        k.b1 = -1.0 / 12 + th2 * (1.0 / 180 + th2 * (-1.0 / 6720 + th2 * (1.0 / 453600)));
    } else {
        const double th = sqrt(th2);
        double s;
        sincos(th, &s, &k.c);
        const double sh = sin(0.5 * th);
        const double omc = 2.0 * sh * sh;  // 1 - cos, without cancellation
        k.a = s / th;
        k.b = omc / th2;
        k.a1 = (th * k.c - s) / (th2 * th);
        k.b1 = (th * s - 2.0 * omc) / (th2 * th2);
    }
    return k;
}

// Camera parameters + rotation coefficients held as VALUES (11 doubles).  A wave that works on one camera makes them
// wave-uniform with cam_vals_uniform(): they then live in scalar registers and reach the f64 FMAs as SGPR operands
// instead of occupying 22 vector registers per camera.
struct CamVals {
    double rx, ry, rz, tx, ty, tz;
    CamCoef k;
};

__device__ __forceinline__ double uniform_f64(double x) {
    const unsigned long long u = __double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((int)(u >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ CamVals cam_vals_uniform(const double *__restrict__ cam) {
    const CamCoef k = cam_coef_of(cam);
    CamVals v;
    v.rx = uniform_f64(cam[0]);
    v.ry = uniform_f64(cam[1]);
    v.rz = uniform_f64(cam[2]);
    v.tx = uniform_f64(cam[3]);
    v.ty = uniform_f64(cam[4]);
    v.tz = uniform_f64(cam[5]);
    v.k.c = uniform_f64(k.c);
    v.k.a = uniform_f64(k.a);
    v.k.b = uniform_f64(k.b);
    v.k.a1 = uniform_f64(k.a1);
    v.k.b1 = uniform_f64(k.b1);
    return v;
}

template <bool WANT_JC, bool WANT_JP>
__device__ __forceinline__ void ba_eval_vals(const CamVals &cv, const double *__restrict__ Xp, const double *__restrict__ K,
                                             double ox, double oy, Proj &o);

template <bool WANT_JC, bool WANT_JP>
__device__ __forceinline__ void ba_eval_cc(const double *__restrict__ cam, const CamCoef &cc,
                                           const double *__restrict__ Xp, const double *__restrict__ K, double ox,
                                           double oy, Proj &o) {
    CamVals cv;
    cv.rx = cam[0];
    cv.ry = cam[1];
    cv.rz = cam[2];
    cv.tx = cam[3];
    cv.ty = cam[4];
    cv.tz = cam[5];
    cv.k = cc;
    ba_eval_vals<WANT_JC, WANT_JP>(cv, Xp, K, ox, oy, o);
}

template <bool WANT_JC, bool WANT_JP>
__device__ __forceinline__ void ba_eval_vals(const CamVals &cv, const double *__restrict__ Xp, const double *__restrict__ K,
                                             double ox, double oy, Proj &o) {
    const double rx = cv.rx, ry = cv.ry, rz = cv.rz;
    const CamCoef &cc = cv.k;
    const double X0 = Xp[0], X1 = Xp[1], X2 = Xp[2];
    const double c = cc.c, a = cc.a, b = cc.b, a1 = cc.a1, b1 = cc.b1;
    // r x X and r.X
    const double cx0 = ry * X2 - rz * X1, cx1 = rz * X0 - rx * X2, cx2 = rx * X1 - ry * X0;
    const double rdx = rx * X0 + ry * X1 + rz * X2;
    const double Xr0 = c * X0 + a * cx0 + b * rdx * rx;
    const double Xr1 = c * X1 + a * cx1 + b * rdx * ry;
    const double Xr2 = c * X2 + a * cx2 + b * rdx * rz;
    const double Y0 = Xr0 + cv.tx, Y1 = Xr1 + cv.ty, Y2 = Xr2 + cv.tz;
    const double u0 = K[0] * Y0 + K[1] * Y1 + K[2] * Y2;
    const double u1 = K[3] * Y0 + K[4] * Y1 + K[5] * Y2;
    const double u2 = K[6] * Y0 + K[7] * Y1 + K[8] * Y2;
    const double iz = 1.0 / u2;
    const double p0 = u0 * iz, p1 = u1 * iz;
    o.r0 = p0 - ox;
    o.r1 = p1 - oy;
    if (!WANT_JC && !WANT_JP) return;
    double M[2][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        M[0][j] = (K[j] - p0 * K[6 + j]) * iz;
        M[1][j] = (K[3 + j] - p1 * K[6 + j]) * iz;
    }
    if (WANT_JC) {
        // common vector  w = -a X + a1 (r x X) + b1 (r.X) r
        const double w0 = -a * X0 + a1 * cx0 + b1 * rdx * rx;
        const double w1 = -a * X1 + a1 * cx1 + b1 * rdx * ry;
        const double w2 = -a * X2 + a1 * cx2 + b1 * rdx * rz;
        const double rr[3] = {rx, ry, rz};
        const double XX[3] = {X0, X1, X2};
        // e_k x X
        const double ex[3][3] = {{0.0, -X2, X1}, {X2, 0.0, -X0}, {-X1, X0, 0.0}};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double d0 = rr[k] * w0 + a * ex[k][0] + b * XX[k] * rx;
            double d1 = rr[k] * w1 + a * ex[k][1] + b * XX[k] * ry;
            double d2 = rr[k] * w2 + a * ex[k][2] + b * XX[k] * rz;
            if (k == 0) d0 += b * rdx;
            if (k == 1) d1 += b * rdx;
            if (k == 2) d2 += b * rdx;
            o.Jc[0][k] = M[0][0] * d0 + M[0][1] * d1 + M[0][2] * d2;
            o.Jc[1][k] = M[1][0] * d0 + M[1][1] * d1 + M[1][2] * d2;
            o.Jc[0][3 + k] = M[0][k];
            o.Jc[1][3 + k] = M[1][k];
        }
    }
    if (WANT_JP) {
        // R = c I + a [r]x + b r r^T
        const double R[3][3] = {{c + b * rx * rx, -a * rz + b * rx * ry, a * ry + b * rx * rz},
                                {a * rz + b * ry * rx, c + b * ry * ry, -a * rx + b * ry * rz},
                                {-a * ry + b * rz * rx, a * rx + b * rz * ry, c + b * rz * rz}};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o.Jp[0][k] = M[0][0] * R[0][k] + M[0][1] * R[1][k] + M[0][2] * R[2][k];
            o.Jp[1][k] = M[1][0] * R[0][k] + M[1][1] * R[1][k] + M[1][2] * R[2][k];
        }
    }
}

// ---- the lean evaluator's camera table (round 4; used by the Schur pair kernel, schur.hip) ---------------------------------
// Everything a sweep needs of an observation follows from per-CAMERA matrices, tabulated once per camera vector:
//   u = P X + q           P = K R (3 x 3), q = K t                       (9 FMA)
//   p = u[:2] / u[2], residual = p - obs
//   Jp[m] = (P[m] - p_m P[2]) / u_2      d proj / d X                    (12)
//   Jt[m] = (K[m] - p_m K[2]) / u_2      d proj / d t                    (12)
//   Jr[m] = (X x Jp[m])^T J_r            d proj / d rvec                 (6 + 9 per row)
// with J_r = a I - b [r]x + e r r^T the right Jacobian of SO(3): d (R X) / d r = -R [X]x J_r, hence
// d proj / d r = M R (-[X]x) J_r = -Jp [X]x J_r (a = sin th / th, b = (1 - cos th) / th^2, e = (1 - a) / th^2).
// ~64 f64 instructions per observation instead of the ~200 of ba_eval_vals (which differentiates the Rodrigues formula
// term by term); the same function up to rounding.  It pays where the camera is wave-uniform (the pair kernel: P, q in
// scalar registers); the per-observation sweeps of ba.hip gather their camera per lane and were slower with the 21-double
// row than with the 11 doubles ba_eval_cc reads (measured, see ba.hip).
constexpr int CAMTAB2 = 24;      // doubles per camera: P[9] q[3] Jr[9] + 3 padding (16-byte rows)
__device__ __forceinline__ void cam_table2_row(const double *__restrict__ c, const double *__restrict__ K, double *__restrict__ t) {
    const double rx = c[0], ry = c[1], rz = c[2];
    const double th2 = rx * rx + ry * ry + rz * rz;
    const CamCoef k = cam_coef_of(c);
    double e;
    if (th2 < 1e-2)      // (1 - sin th / th) / th^2 without cancellation
        e = 1.0 / 6 + th2 * (-1.0 / 120 + th2 * (1.0 / 5040 + th2 * (-1.0 / 362880 + th2 * (1.0 / 39916800 - th2 * (1.0 / 6227020800.0)))));
    else
        e = (1.0 - k.a) / th2;
    const double R[3][3] = {{k.c + k.b * rx * rx, -k.a * rz + k.b * rx * ry, k.a * ry + k.b * rx * rz},
                            {k.a * rz + k.b * ry * rx, k.c + k.b * ry * ry, -k.a * rx + k.b * ry * rz},
                            {-k.a * ry + k.b * rz * rx, k.a * rx + k.b * rz * ry, k.c + k.b * rz * rz}};
    for (int m = 0; m < 3; ++m) {
        for (int j = 0; j < 3; ++j) t[3 * m + j] = K[3 * m] * R[0][j] + K[3 * m + 1] * R[1][j] + K[3 * m + 2] * R[2][j];
        t[9 + m] = K[3 * m] * c[3] + K[3 * m + 1] * c[4] + K[3 * m + 2] * c[5];
    }
    const double rr[3] = {rx, ry, rz};
    const double hat[3][3] = {{0.0, -rz, ry}, {rz, 0.0, -rx}, {-ry, rx, 0.0}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[12 + 3 * i + j] = (i == j ? k.a : 0.0) - k.b * hat[i][j] + e * rr[i] * rr[j];
    t[21] = t[22] = t[23] = 0.0;
}

template <bool WANT_JC, bool WANT_JP>
__device__ __forceinline__ void ba_eval(const double *__restrict__ cam, const double *__restrict__ Xp,
                                        const double *__restrict__ K, double ox, double oy, Proj &o) {
    ba_eval_cc<WANT_JC, WANT_JP>(cam, cam_coef_of(cam), Xp, K, ox, oy, o);
}

// Regulariser of the 2-D subspace trust-region step (SciPy trf.py:473-477): with a = 0.5 |J_h g_h|^2, b = -|g_h|^2 minimise
// t (a t + b) over [0, Delta / |g_h|]; reg = -min / Delta^2.  Shared by trf_damping_kernel (ba.hip) and the damping sweep
// that computes it itself (vec.hip): the same expression, the same bits.
__device__ __forceinline__ double trf_damping_value(double gh2, double d11, double Delta) {
    const double a = 0.5 * d11, b = -gh2;
    const double to_tr = Delta / sqrt(gh2);
    double best = fmin(0.0, to_tr * (a * to_tr + b));
    if (a != 0.0) {
        const double ext = -0.5 * b / a;
        if (ext > 0.0 && ext < to_tr) best = fmin(best, ext * (a * ext + b));
    }
    return -best / (Delta * Delta);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// N independent wave sums at once: all N shuffles of a tree level are issued back to back, so their ~100-cycle
// latencies overlap (N separate wave_sum calls serialise 6 N dependent LDS round trips: 10 us for N = 42).
template <int N>
__device__ __forceinline__ void wave_sum_n(double (&v)[N]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double t[N];
#pragma unroll
        for (int q = 0; q < N; ++q) t[q] = __shfl_down(v[q], off, 64);
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] += t[q];
    }
}

// N workgroup sums at once (fixed tree: lanes by shuffles, then waves in index order); results valid in thread 0.
// sm must hold (THREADS / 64) * N doubles.
template <int N, int THREADS>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double *sm) {
    wave_sum_n<N>(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) sm[w * N + q] = v[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < N; ++q) {
            double t = 0;
#pragma unroll
            for (int i = 0; i < THREADS / 64; ++i) t += sm[i * N + q];
            v[q] = t;
        }
    }
}

// Deterministic workgroup sum (fixed tree); result valid in thread 0.
template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double *sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < THREADS / 64; ++i) t += sm[i];
    }
    return t;
}

// adds the per-workgroup partials of ba_jvp_dots_kernel in index order (one workgroup; a shared arrival counter costs
// ~11 ns per workgroup on 6 k workgroups -- more than this launch.  Round 4 re-measured it with the 2048-workgroup cap:
// finishing the sums in the kernel's last-arriving workgroup made the product 45 us instead of 27 + 5 for this launch.)
// (returns the two sums, valid in thread 0)
__device__ __forceinline__ double2 jvp_rows_body(const double *__restrict__ partial, unsigned n_wg, double *__restrict__ rows, const unsigned bx, const unsigned gx) {
    __shared__ double sm[(256 / 64) * 2];
    double acc[2] = {0.0, 0.0};
    for (unsigned g = threadIdx.x; g < n_wg; g += 256) {
        acc[0] += partial[2 * (size_t)g];
        acc[1] += partial[2 * (size_t)g + 1];
    }
    block_sum_n<2, 256>(acc, sm);
    if (threadIdx.x == 0) {
        rows[0] = 0.0;
        rows[1] = acc[0];
        rows[2] = acc[0];
        rows[3] = 0.0;
        rows[4] = acc[1];
        rows[5] = acc[1];
    }
    return make_double2(acc[0], acc[1]);
}

}  // namespace
