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

template <bool WANT_JC, bool WANT_JP>
__device__ __forceinline__ void ba_eval(const double *__restrict__ cam, const double *__restrict__ Xp,
                                        const double *__restrict__ K, double ox, double oy, Proj &o) {
    const double rx = cam[0], ry = cam[1], rz = cam[2];
    const double X0 = Xp[0], X1 = Xp[1], X2 = Xp[2];
    const double th2 = rx * rx + ry * ry + rz * rz;
    double c, a, b, a1 = 0, b1 = 0;
    if (th2 < 1e-4) {
        c = cos(sqrt(th2));
        a = 1.0 + th2 * (-1.0 / 6 + th2 * (1.0 / 120 - th2 * (1.0 / 5040)));
        b = 0.5 + th2 * (-1.0 / 24 + th2 * (1.0 / 720 - th2 * (1.0 / 40320)));
        if (WANT_JC) {
            a1 = -1.0 / 3 + th2 * (1.0 / 30 + th2 * (-1.0 / 840 + th2 * (1.0 / 45360)));
            b1 = -1.0 / 12 + th2 * (1.0 / 180 + th2 * (-1.0 / 6720 + th2 * (1.0 / 453600)));
        }
    } else {
        const double th = sqrt(th2);
        double s;
        sincos(th, &s, &c);
        const double sh = sin(0.5 * th);
        const double omc = 2.0 * sh * sh;  // 1 - cos, without cancellation
        a = s / th;
        b = omc / th2;
        if (WANT_JC) {
            a1 = (th * c - s) / (th2 * th);
            b1 = (th * s - 2.0 * omc) / (th2 * th2);
        }
    }
    // r x X and r.X
    const double cx0 = ry * X2 - rz * X1, cx1 = rz * X0 - rx * X2, cx2 = rx * X1 - ry * X0;
    const double rdx = rx * X0 + ry * X1 + rz * X2;
    const double Xr0 = c * X0 + a * cx0 + b * rdx * rx;
    const double Xr1 = c * X1 + a * cx1 + b * rdx * ry;
    const double Xr2 = c * X2 + a * cx2 + b * rdx * rz;
    const double Y0 = Xr0 + cam[3], Y1 = Xr1 + cam[4], Y2 = Xr2 + cam[5];
    const double u0 = K[0] * Y0 + K[1] * Y1 + K[2] * Y2;
    const double u1 = K[3] * Y0 + K[4] * Y1 + K[5] * Y2;
    const double u2 = K[6] * Y0 + K[7] * Y1 + K[8] * Y2;
    const double p0 = u0 / u2, p1 = u1 / u2;
    o.r0 = p0 - ox;
    o.r1 = p1 - oy;
    if (!WANT_JC && !WANT_JP) return;
    const double iz = 1.0 / u2;
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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
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

}  // namespace
