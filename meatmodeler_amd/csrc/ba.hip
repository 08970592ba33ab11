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

// ---- residual + cost ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_residual_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                          const double *__restrict__ pts, double *__restrict__ res,
                                                          double *__restrict__ partial) {
    __shared__ double sm[4];
    __shared__ double Ks[9];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    double acc = 0;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < pb.O; o += (int64_t)gridDim.x * 256) {
        Proj pr;
        ba_eval<false, false>(cams + (size_t)pb.fi[o] * 6, pts + (size_t)pb.pi[o] * 3, Ks, pb.obs[2 * o],
                              pb.obs[2 * o + 1], pr);
        if (res) {
            res[2 * o] = pr.r0;
            res[2 * o + 1] = pr.r1;
        }
        acc += pr.r0 * pr.r0 + pr.r1 * pr.r1;
    }
    double t = block_sum<256>(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double *__restrict__ partial, int n,
                                                           double *__restrict__ out) {
    __shared__ double sm[4];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    double t = block_sum<256>(acc, sm);
    if (threadIdx.x == 0) out[0] = t;
}

// ---- analytic Jacobian blocks (parity surface) --------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_jacobian_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                          const double *__restrict__ pts, double *__restrict__ Jc,
                                                          double *__restrict__ Jp) {
    __shared__ double Ks[9];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= pb.O) return;
    Proj pr;
    ba_eval<true, true>(cams + (size_t)pb.fi[o] * 6, pts + (size_t)pb.pi[o] * 3, Ks, pb.obs[2 * o], pb.obs[2 * o + 1],
                        pr);
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

// ---- point blocks: C[P,6] (upper triangle) and gp[P,3]; one thread per point, its observations are contiguous ------
__global__ __launch_bounds__(256) void ba_point_blocks_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                              const double *__restrict__ pts, double *__restrict__ C,
                                                              double *__restrict__ gp) {
    __shared__ double Ks[9];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= pb.P) return;
    double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0, g0 = 0, g1 = 0, g2 = 0;
    const double *Xp = pts + (size_t)p * 3;
    for (int e = pb.pt_ptr[p]; e < pb.pt_ptr[p + 1]; ++e) {
        int o = pb.pt_obs[e];
        Proj pr;
        ba_eval<false, true>(cams + (size_t)pb.fi[o] * 6, Xp, Ks, pb.obs[2 * (size_t)o], pb.obs[2 * (size_t)o + 1], pr);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const double j0 = pr.Jp[m][0], j1 = pr.Jp[m][1], j2 = pr.Jp[m][2];
            const double r = m == 0 ? pr.r0 : pr.r1;
            c00 += j0 * j0; c01 += j0 * j1; c02 += j0 * j2;
            c11 += j1 * j1; c12 += j1 * j2; c22 += j2 * j2;
            g0 += j0 * r; g1 += j1 * r; g2 += j2 * r;
        }
    }
    double *Cp = C + (size_t)p * 6;
    Cp[0] = c00; Cp[1] = c01; Cp[2] = c02; Cp[3] = c11; Cp[4] = c12; Cp[5] = c22;
    gp[(size_t)p * 3] = g0; gp[(size_t)p * 3 + 1] = g1; gp[(size_t)p * 3 + 2] = g2;
}

// ---- camera blocks: B[F,6,6] and gc[F,6]; one workgroup per camera over its (gathered) observations -----------------
__global__ __launch_bounds__(256) void ba_camera_blocks_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                               const double *__restrict__ pts, double *__restrict__ B,
                                                               double *__restrict__ gc) {
    __shared__ double sm[4];
    __shared__ double Ks[9];
    __shared__ double cs[6];
    const int f = blockIdx.x;
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    if (threadIdx.x < 6) cs[threadIdx.x] = cams[(size_t)f * 6 + threadIdx.x];
    __syncthreads();
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0;
    for (int e = pb.cam_ptr[f] + threadIdx.x; e < pb.cam_ptr[f + 1]; e += 256) {
        int o = pb.cam_obs[e];
        Proj pr;
        ba_eval<true, false>(cs, pts + (size_t)pb.pi[o] * 3, Ks, pb.obs[2 * (size_t)o], pb.obs[2 * (size_t)o + 1], pr);
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
    int t = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) {
            double s = block_sum<256>(acc[t], sm);
            if (threadIdx.x == 0) {
                B[(size_t)f * 36 + i * 6 + j] = s;
                B[(size_t)f * 36 + j * 6 + i] = s;
            }
            ++t;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = block_sum<256>(acc[21 + i], sm);
        if (threadIdx.x == 0) gc[(size_t)f * 6 + i] = s;
    }
}

// ---- out = Jc wc[fi] + Jp wp[pi] ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_jvp_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                     const double *__restrict__ pts, const double *__restrict__ wc,
                                                     const double *__restrict__ wp, double *__restrict__ out) {
    __shared__ double Ks[9];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= pb.O) return;
    const int f = pb.fi[o], p = pb.pi[o];
    Proj pr;
    ba_eval<true, true>(cams + (size_t)f * 6, pts + (size_t)p * 3, Ks, pb.obs[2 * o], pb.obs[2 * o + 1], pr);
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

// ---- Schur complement ------------------------------------------------------------------------------------------------
// S <- blockdiag(Bd), v <- gc
__global__ __launch_bounds__(256) void schur_init_kernel(int F, const double *__restrict__ Bd,
                                                         const double *__restrict__ gc, double *__restrict__ S,
                                                         double *__restrict__ v) {
    const size_t n = (size_t)F * 6;
    const size_t total = n * n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t r = i / n, c = i % n;
        size_t fr = r / 6, fc = c / 6;
        S[i] = (fr == fc) ? Bd[fr * 36 + (r % 6) * 6 + (c % 6)] : 0.0;
        if (i < n) v[i] = gc[i];
    }
}

// Cinv = Cd^-1 (3x3 symmetric, upper triangle storage xx,xy,xz,yy,yz,zz)
__global__ __launch_bounds__(256) void point_inverse_kernel(int P, const double *__restrict__ Cd,
                                                            double *__restrict__ Cinv) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const double *c = Cd + (size_t)p * 6;
    const double a = c[0], b = c[1], d = c[2], e = c[3], f = c[4], g = c[5];
    const double m00 = e * g - f * f, m01 = d * f - b * g, m02 = b * f - d * e;
    const double det = a * m00 + b * m01 + d * m02;
    const double id = 1.0 / det;
    double *o = Cinv + (size_t)p * 6;
    o[0] = m00 * id;
    o[1] = m01 * id;
    o[2] = m02 * id;
    o[3] = (a * g - d * d) * id;
    o[4] = (b * d - a * f) * id;
    o[5] = (a * e - b * b) * id;
}

// One wave per point.  Lanes first build E_i = Jc_i^T Jp_i and Y_i = E_i Cinv for a tile of the point's observations
// in LDS, then the wave spreads the (i, j) camera-pair 6x6 blocks  -Y_i E_j^T  over its lanes and adds them into the
// LOWER block triangle of S with f64 atomics; v gets  -Y_i gp.
constexpr int SCH_TILE = 16;
constexpr int SCH_WAVES = 4;

__global__ __launch_bounds__(64 * SCH_WAVES) void schur_accum_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                                     const double *__restrict__ pts,
                                                                     const double *__restrict__ Cinv,
                                                                     const double *__restrict__ gp,
                                                                     double *__restrict__ S, double *__restrict__ v) {
    __shared__ double Ks[9];
    __shared__ double Ei[SCH_WAVES][SCH_TILE][18], Yi[SCH_WAVES][SCH_TILE][18], Ej[SCH_WAVES][SCH_TILE][18];
    __shared__ int fI[SCH_WAVES][SCH_TILE], fJ[SCH_WAVES][SCH_TILE];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t n = (size_t)pb.F * 6;
    for (int p = blockIdx.x * SCH_WAVES + w; p < pb.P; p += gridDim.x * SCH_WAVES) {
        const int e0 = pb.pt_ptr[p], e1 = pb.pt_ptr[p + 1];
        const double *Xp = pts + (size_t)p * 3;
        const double *ci = Cinv + (size_t)p * 6;
        const double q00 = ci[0], q01 = ci[1], q02 = ci[2], q11 = ci[3], q12 = ci[4], q22 = ci[5];
        const double g0 = gp[(size_t)p * 3], g1 = gp[(size_t)p * 3 + 1], g2 = gp[(size_t)p * 3 + 2];
        for (int it = e0; it < e1; it += SCH_TILE) {
            const int ni = min(SCH_TILE, e1 - it);
            // --- tile i: E, Y
            if (lane < ni) {
                int o = pb.pt_obs[it + lane];
                int f = pb.fi[o];
                Proj pr;
                ba_eval<true, true>(cams + (size_t)f * 6, Xp, Ks, pb.obs[2 * (size_t)o], pb.obs[2 * (size_t)o + 1], pr);
                fI[w][lane] = f;
                double yg[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    double ea0 = pr.Jc[0][a] * pr.Jp[0][0] + pr.Jc[1][a] * pr.Jp[1][0];
                    double ea1 = pr.Jc[0][a] * pr.Jp[0][1] + pr.Jc[1][a] * pr.Jp[1][1];
                    double ea2 = pr.Jc[0][a] * pr.Jp[0][2] + pr.Jc[1][a] * pr.Jp[1][2];
                    Ei[w][lane][a * 3 + 0] = ea0;
                    Ei[w][lane][a * 3 + 1] = ea1;
                    Ei[w][lane][a * 3 + 2] = ea2;
                    double y0 = ea0 * q00 + ea1 * q01 + ea2 * q02;
                    double y1 = ea0 * q01 + ea1 * q11 + ea2 * q12;
                    double y2 = ea0 * q02 + ea1 * q12 + ea2 * q22;
                    Yi[w][lane][a * 3 + 0] = y0;
                    Yi[w][lane][a * 3 + 1] = y1;
                    Yi[w][lane][a * 3 + 2] = y2;
                    yg[a] = y0 * g0 + y1 * g1 + y2 * g2;
                }
#pragma unroll
                for (int a = 0; a < 6; ++a) unsafeAtomicAdd(&v[(size_t)f * 6 + a], -yg[a]);
            }
            for (int jt = it; jt < e1; jt += SCH_TILE) {
                const int nj = min(SCH_TILE, e1 - jt);
                // --- tile j: E  (the diagonal tile reuses tile i)
                if (jt != it) {
                    if (lane < nj) {
                        int o = pb.pt_obs[jt + lane];
                        int f = pb.fi[o];
                        Proj pr;
                        ba_eval<true, true>(cams + (size_t)f * 6, Xp, Ks, pb.obs[2 * (size_t)o],
                                            pb.obs[2 * (size_t)o + 1], pr);
                        fJ[w][lane] = f;
#pragma unroll
                        for (int a = 0; a < 6; ++a) {
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                Ej[w][lane][a * 3 + k] = pr.Jc[0][a] * pr.Jp[0][k] + pr.Jc[1][a] * pr.Jp[1][k];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const bool diag = (jt == it);
                const int nent = ni * nj * 36;
                for (int x = lane; x < nent; x += 64) {
                    const int ab = x % 36, pairi = x / 36;
                    const int i = pairi / nj, j = pairi % nj;
                    if (diag && j < i) continue;  // unordered pairs once
                    const int a = ab / 6, b2 = ab % 6;
                    const double *Y = Yi[w][i];
                    const double *E = diag ? Ei[w][j] : Ej[w][j];
                    const int fi_ = fI[w][i], fj_ = diag ? fI[w][j] : fJ[w][j];
                    const double val = -(Y[a * 3] * E[b2 * 3] + Y[a * 3 + 1] * E[b2 * 3 + 1] + Y[a * 3 + 2] * E[b2 * 3 + 2]);
                    // block (fi, fj) entry (a, b);  keep the lower block triangle
                    const bool same_obs = diag && (i == j);
                    if (fi_ > fj_ || same_obs) {
                        unsafeAtomicAdd(&S[((size_t)fi_ * 6 + a) * n + (size_t)fj_ * 6 + b2], val);
                    } else if (fi_ < fj_) {
                        unsafeAtomicAdd(&S[((size_t)fj_ * 6 + b2) * n + (size_t)fi_ * 6 + a], val);
                    } else {  // two different observations of the point in the same camera: M + M^T on the diagonal block
                        unsafeAtomicAdd(&S[((size_t)fi_ * 6 + a) * n + (size_t)fj_ * 6 + b2], val);
                        unsafeAtomicAdd(&S[((size_t)fj_ * 6 + b2) * n + (size_t)fi_ * 6 + a], val);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// ---- back-substitution: dp = Cinv (gp - sum_o Jp_o^T (Jc_o dc[f_o])) ------------------------------------------------
__global__ __launch_bounds__(256) void ba_backsub_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                         const double *__restrict__ pts,
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
        ba_eval<true, true>(cams + (size_t)f * 6, Xp, Ks, pb.obs[2 * (size_t)o], pb.obs[2 * (size_t)o + 1], pr);
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

int check_pb(mm_ctx *ctx, const mm_ba_problem *pb, const char *who) {
    if (!ctx) return MM_ERR_ARG;
    if (!pb || pb->F < 0 || pb->P < 0 || pb->O < 0 || !pb->K) return mm_fail(ctx, MM_ERR_ARG, "%s: bad problem", who);
    if (pb->O > 0 && (!pb->fi || !pb->pi || !pb->obs)) return mm_fail(ctx, MM_ERR_ARG, "%s: null observation arrays", who);
    return MM_OK;
}

constexpr int RES_BLOCKS = 2048;

}  // namespace

extern "C" {

int mm_ba_residual(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, double *res,
                   double *cost2, void *ws, size_t ws_bytes) {
    int rc = check_pb(ctx, pb, "mm_ba_residual");
    if (rc) return rc;
    if (!cams || !pts || !cost2) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_residual: null pointer");
    if (!ws || ws_bytes < RES_BLOCKS * sizeof(double)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_ba_residual: workspace < 16 KiB");
    int64_t nb = (pb->O + 255) / 256;
    int blocks = (int)(nb < 1 ? 1 : (nb > RES_BLOCKS ? RES_BLOCKS : nb));
    MM_LAUNCH(ctx, "ba_residual_kernel", ba_residual_kernel, dim3(blocks), dim3(256), 0, *pb, cams, pts, res, (double *)ws);
    MM_LAUNCH(ctx, "sum_partials_kernel", sum_partials_kernel, dim3(1), dim3(256), 0, (const double *)ws, blocks, cost2);
    return MM_OK;
}

int mm_ba_jacobian(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, double *Jc, double *Jp) {
    int rc = check_pb(ctx, pb, "mm_ba_jacobian");
    if (rc) return rc;
    if (!cams || !pts || !Jc || !Jp) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_jacobian: null pointer");
    if (pb->O == 0) return MM_OK;
    MM_LAUNCH(ctx, "ba_jacobian_kernel", ba_jacobian_kernel, dim3((unsigned)((pb->O + 255) / 256)), dim3(256), 0, *pb, cams, pts, Jc, Jp);
    return MM_OK;
}

int mm_ba_normal_eq(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, double *B, double *gc,
                    double *C, double *gp) {
    int rc = check_pb(ctx, pb, "mm_ba_normal_eq");
    if (rc) return rc;
    if (!cams || !pts) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_normal_eq: null pointer");
    if ((B == nullptr) != (gc == nullptr) || (C == nullptr) != (gp == nullptr))
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_normal_eq: B/gc and C/gp come in pairs");
    if (C) {
        if (!pb->pt_ptr || !pb->pt_obs) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_normal_eq: point CSR missing");
        if (pb->P > 0) {
            MM_LAUNCH(ctx, "ba_point_blocks_kernel", ba_point_blocks_kernel, dim3((pb->P + 255) / 256), dim3(256), 0, *pb, cams, pts, C, gp);
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
    MM_LAUNCH(ctx, "ba_jvp_kernel", ba_jvp_kernel, dim3((unsigned)((pb->O + 255) / 256)), dim3(256), 0, *pb, cams, pts, wc, wp, out);
    return MM_OK;
}

int mm_ba_schur(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const double *Bd,
                const double *Cd, const double *gc, const double *gp, double *S, double *v, double *Cinv) {
    int rc = check_pb(ctx, pb, "mm_ba_schur");
    if (rc) return rc;
    if (!cams || !pts || !Bd || !Cd || !gc || !gp || !S || !v || !Cinv || !pb->pt_ptr || !pb->pt_obs)
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_schur: null pointer");
    if (pb->F == 0) return MM_OK;
    size_t total = (size_t)pb->F * 6 * pb->F * 6;
    int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    MM_LAUNCH(ctx, "schur_init_kernel", schur_init_kernel, dim3(blocks), dim3(256), 0, pb->F, Bd, gc, S, v);
    if (pb->P == 0) return MM_OK;
    MM_LAUNCH(ctx, "point_inverse_kernel", point_inverse_kernel, dim3((pb->P + 255) / 256), dim3(256), 0, pb->P, Cd, Cinv);
    int wgs = (pb->P + SCH_WAVES - 1) / SCH_WAVES;
    if (wgs > 8192) wgs = 8192;
    MM_LAUNCH(ctx, "schur_accum_kernel", schur_accum_kernel, dim3(wgs), dim3(64 * SCH_WAVES), 0, *pb, cams, pts, Cinv, gp, S, v);
    return MM_OK;
}

int mm_ba_backsub(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const double *Cinv,
                  const double *gp, const double *dc, double *dp) {
    int rc = check_pb(ctx, pb, "mm_ba_backsub");
    if (rc) return rc;
    if (!cams || !pts || !Cinv || !gp || !dc || !dp || !pb->pt_ptr || !pb->pt_obs)
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_backsub: null pointer");
    if (pb->P == 0) return MM_OK;
    MM_LAUNCH(ctx, "ba_backsub_kernel", ba_backsub_kernel, dim3((pb->P + 255) / 256), dim3(256), 0, *pb, cams, pts, Cinv, gp, dc, dp);
    return MM_OK;
}

}  // extern "C"
