// Two-view homogeneous DLT triangulation, one thread per track (gfx950, f64).
//
// Replaces the per-track loop of processor.triangulatePoints (reference processor.py:254-261):
//   cv2.triangulatePoints(P_first, P_last, x_first, x_last) -> X[:3] / X[3].
// OpenCV builds the 4x4 system  x*P[2]-P[0], y*P[2]-P[1]  for both views and takes the right singular vector of
// the smallest singular value (calib3d triangulate.cpp, one-sided Jacobi SVD).  The same one-sided (Hestenes)
// Jacobi runs here entirely in registers: 16 + 16 doubles per thread, fully unrolled so every index is static.
// The work is tiny (64 B in, 24 B out, ~2 kflop per track): the kernel is latency-bound by construction.
#include "mm_common.h"

namespace {

__global__ __launch_bounds__(256) void dlt_kernel(const double *__restrict__ proj, const int32_t *__restrict__ f0,
                                                  const int32_t *__restrict__ f1, const double *__restrict__ x0,
                                                  const double *__restrict__ x1, int64_t n, double *__restrict__ X) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *P0 = proj + (size_t)f0[i] * 12;
    const double *P1 = proj + (size_t)f1[i] * 12;
    const double ax = x0[2 * i], ay = x0[2 * i + 1], bx = x1[2 * i], by = x1[2 * i + 1];
    double A[4][4], V[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        A[0][c] = ax * P0[8 + c] - P0[c];
        A[1][c] = ay * P0[8 + c] - P0[4 + c];
        A[2][c] = bx * P1[8 + c] - P1[c];
        A[3][c] = by * P1[8 + c] - P1[4 + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) V[r][c] = (r == c) ? 1.0 : 0.0;
    }
    const double eps = 2.220446049250313e-16;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double a = 0, b = 0, g = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a += A[r][p] * A[r][p];
                    b += A[r][q] * A[r][q];
                    g += A[r][p] * A[r][q];
                }
                if (fabs(g) > eps * sqrt(a * b) && g != 0.0) {
                    rotated = true;
                    double zeta = (b - a) / (2.0 * g);
                    double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    double c = 1.0 / sqrt(1.0 + t * t);
                    double s = c * t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double ap = A[r][p], aq = A[r][q];
                        A[r][p] = c * ap - s * aq;
                        A[r][q] = s * ap + c * aq;
                        double vp = V[r][p], vq = V[r][q];
                        V[r][p] = c * vp - s * vq;
                        V[r][q] = s * vp + c * vq;
                    }
                }
            }
        }
        if (!rotated) break;
    }
    // column with the smallest norm = smallest singular value
    double best = 1e300;
    double v0 = 0, v1 = 0, v2 = 0, v3 = 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double nn = A[0][c] * A[0][c] + A[1][c] * A[1][c] + A[2][c] * A[2][c] + A[3][c] * A[3][c];
        if (nn < best) {
            best = nn;
            v0 = V[0][c];
            v1 = V[1][c];
            v2 = V[2][c];
            v3 = V[3][c];
        }
    }
    X[3 * i] = v0 / v3;
    X[3 * i + 1] = v1 / v3;
    X[3 * i + 2] = v2 / v3;
}

}  // namespace

extern "C" int mm_triangulate_dlt(mm_ctx *ctx, const double *proj, const int32_t *f0, const int32_t *f1,
                                  const double *x0, const double *x1, int64_t n, double *X) {
    if (!ctx) return MM_ERR_ARG;
    if (n == 0) return MM_OK;
    if (!proj || !f0 || !f1 || !x0 || !x1 || !X || n < 0) return mm_fail(ctx, MM_ERR_ARG, "mm_triangulate_dlt: bad argument");
    int64_t blocks = (n + 255) / 256;
    MM_LAUNCH(ctx, "dlt_kernel", dlt_kernel, dim3((unsigned)blocks), dim3(256), 0, proj, f0, f1, x0, x1, n, X);
    return MM_OK;
}
