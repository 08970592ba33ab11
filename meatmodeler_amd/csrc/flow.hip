// Keyframe gating front end (gfx950): pyramidal Lucas-Kanade tracking and Shi-Tomasi corners.
//
// Replaces the two OpenCV calls of processor.keyframeTracking (reference processor.py:61-110):
//   cv2.calcOpticalFlowPyrLK(prev, cur, pts, None, **lk_params)          :79
//   cv2.goodFeaturesToTrack(grey, mask=None, **feature_params)           :104
// OpenCV is absent offline, so the arithmetic is DEFINED by oracle/frame_oracle.c (published algorithms with OpenCV's
// defaults, made integer exact: every window sum is a 64-bit integer, the few floating-point operations after them have
// a fixed order; this file is compiled with -ffp-contract=off) and the kernels below reproduce it bit for bit.
//
//   pyr_down          5-tap [1 4 6 4 1]/16 separable, reflect-101, (sum + 128) >> 8; one output pixel per thread from an
//                     LDS tile (streaming, HBM bound: 1.25 B per input pixel)
//   lk_track          ONE WAVE per point, all pyramid levels in one launch: the (w+3) x (h+3) patch of the previous
//                     image and the (w+1) x (h+1) patch of the current one are staged in LDS, lanes stride over the
//                     window, the five integer sums are all-reduced across the wave with xor butterflies, so every lane
//                     carries the same scalars through the Newton iteration (no divergence, no broadcast)
//   min_eig           Sobel structure tensor, block x block box sums from an LDS tile of derivatives, f64 eigenvalue
//   corner_candidates threshold + 3x3 non-maximum suppression, compaction by one atomic per wave
// The greedy minimum-distance selection walks the candidates in order of strength and stops after maxCorners: a short,
// inherently sequential loop over a grid of buckets -- host code (host_index.cpp: mm_gftt_select), as in OpenCV.
#include "mm_common.h"

namespace {

__device__ __forceinline__ int refl(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// ---- pyramid -----------------------------------------------------------------------------------------------------------
constexpr int PD_TW = 64, PD_TH = 16;  // output tile
__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t *__restrict__ src, int w, int h, int ps,
                                                       uint8_t *__restrict__ dst, int wd, int hd, int pd) {
    __shared__ uint8_t T[2 * PD_TH + 4][2 * PD_TW + 4];
    const int ox = blockIdx.x * PD_TW, oy = blockIdx.y * PD_TH;
    for (int e = threadIdx.x; e < (2 * PD_TH + 4) * (2 * PD_TW + 4); e += 256) {
        const int r = e / (2 * PD_TW + 4), c = e % (2 * PD_TW + 4);
        T[r][c] = src[(size_t)refl(2 * oy - 2 + r, h) * ps + refl(2 * ox - 2 + c, w)];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PD_TW * PD_TH; e += 256) {
        const int ty = e / PD_TW, tx = e % PD_TW;
        const int x = ox + tx, y = oy + ty;
        if (x >= wd || y >= hd) continue;
        int s = 0;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
            const int ky = dy == 0 || dy == 4 ? 1 : (dy == 2 ? 6 : 4);
            const uint8_t *row = &T[2 * ty + dy][2 * tx];
            s += ky * (row[0] + 4 * row[1] + 6 * row[2] + 4 * row[3] + row[4]);
        }
        dst[(size_t)y * pd + x] = (uint8_t)((s + 128) >> 8);
    }
}

// ---- Lucas-Kanade ------------------------------------------------------------------------------------------------------
constexpr int LK_MAX_LEVELS = 8;
constexpr int LK_MAX_WIN = 41;
constexpr int W_BITS = 14;
struct LkLevels {
    const uint8_t *prev[LK_MAX_LEVELS];
    const uint8_t *next[LK_MAX_LEVELS];
    int w[LK_MAX_LEVELS], h[LK_MAX_LEVELS], p[LK_MAX_LEVELS];
    int levels;
};

__device__ __forceinline__ long long wave_allsum(long long v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ long long descale(long long x, int n) { return (x + (1ll << (n - 1))) >> n; }

struct Weights {
    int w00, w01, w10, w11;
};
__device__ __forceinline__ Weights bilinear_weights(float a, float b) {
    Weights q;
    q.w00 = __float2int_rn((1.0f - a) * (1.0f - b) * (float)(1 << W_BITS));
    q.w01 = __float2int_rn(a * (1.0f - b) * (float)(1 << W_BITS));
    q.w10 = __float2int_rn((1.0f - a) * b * (float)(1 << W_BITS));
    q.w11 = (1 << W_BITS) - q.w00 - q.w01 - q.w10;
    return q;
}

// stage the (cw x ch) patch with top-left corner (x0, y0) of a level image (reflect-101) into LDS
__device__ __forceinline__ void stage_patch(uint8_t *dst, const uint8_t *__restrict__ img, int W, int H, int P, int x0, int y0,
                                            int cw, int ch, int lane) {
    for (int e = lane; e < cw * ch; e += 64) {
        const int r = e / cw, c = e % cw;
        dst[e] = img[(size_t)refl(y0 + r, H) * P + refl(x0 + c, W)];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64) void lk_track_kernel(LkLevels L, const float *__restrict__ pts, int n, int ww, int wh,
                                                      int max_count, double eps, float *__restrict__ out,
                                                      uint8_t *__restrict__ status, float *__restrict__ err) {
    extern __shared__ int lk_smem[];
    const int npx = ww * wh;
    int *ip = lk_smem, *ix = ip + npx, *iy = ix + npx;
    uint8_t *Ipatch = reinterpret_cast<uint8_t *>(iy + npx);          // (ww + 3) x (wh + 3)
    uint8_t *Jpatch = Ipatch + ((ww + 3) * (wh + 3) + 3) / 4 * 4;    // (ww + 1) x (wh + 1)
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const float hwx = (ww - 1) * 0.5f, hwy = (wh - 1) * 0.5f;
    const float p0x = pts[2 * i], p0y = pts[2 * i + 1];
    float nx = 0.0f, ny = 0.0f, er = 0.0f;
    int st = 1;
    for (int l = L.levels - 1; l >= 0; --l) {
        const float sc = 1.0f / (float)(1 << l);
        float px = p0x * sc, py = p0y * sc;
        if (l == L.levels - 1) {
            nx = px;
            ny = py;
        } else {
            nx = nx * 2.0f;
            ny = ny * 2.0f;
        }
        const uint8_t *I = L.prev[l], *J = L.next[l];
        const int W = L.w[l], H = L.h[l], P = L.p[l];
        px -= hwx;
        py -= hwy;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -ww || ipx >= W || ipy < -wh || ipy >= H) {
            if (l == 0) {
                st = 0;
                er = 0.0f;
            }
            continue;
        }
        Weights q = bilinear_weights(px - (float)ipx, py - (float)ipy);
        // previous image: pixels (ipx - 1 .. ipx + ww + 1) x (ipy - 1 .. ipy + wh + 1)
        const int cw = ww + 3;
        stage_patch(Ipatch, I, W, H, P, ipx - 1, ipy - 1, cw, wh + 3, lane);
        long long sA11 = 0, sA12 = 0, sA22 = 0;
        for (int e = lane; e < npx; e += 64) {
            const int y = e / ww, x = e % ww;
            auto at = [&](int xx, int yy) { return (int)Ipatch[(yy + 1) * cw + xx + 1]; };   // (xx, yy) relative to (ipx, ipy)
            // (derivative samples outside the image are zero, as in OpenCV's zero-padded derivative image; inside, the taps
            // reflect at the edges like the staged intensities)
            auto scharr = [&](int xx, int yy, int &dx, int &dy) {
                if (ipx + xx < 0 || ipx + xx >= W || ipy + yy < 0 || ipy + yy >= H) {
                    dx = dy = 0;
                    return;
                }
                const int a = at(xx - 1, yy - 1), b = at(xx, yy - 1), c = at(xx + 1, yy - 1);
                const int d = at(xx - 1, yy), f = at(xx + 1, yy);
                const int g = at(xx - 1, yy + 1), hh = at(xx, yy + 1), ii = at(xx + 1, yy + 1);
                dx = 3 * (c - a) + 10 * (f - d) + 3 * (ii - g);
                dy = 3 * (g - a) + 10 * (hh - b) + 3 * (ii - c);
            };
            int dx00, dy00, dx01, dy01, dx10, dy10, dx11, dy11;
            scharr(x, y, dx00, dy00);
            scharr(x + 1, y, dx01, dy01);
            scharr(x, y + 1, dx10, dy10);
            scharr(x + 1, y + 1, dx11, dy11);
            const int iv = (int)descale((long long)at(x, y) * q.w00 + (long long)at(x + 1, y) * q.w01 +
                                            (long long)at(x, y + 1) * q.w10 + (long long)at(x + 1, y + 1) * q.w11,
                                        W_BITS - 5);
            const int xv = (int)descale((long long)dx00 * q.w00 + (long long)dx01 * q.w01 + (long long)dx10 * q.w10 + (long long)dx11 * q.w11, W_BITS);
            const int yv = (int)descale((long long)dy00 * q.w00 + (long long)dy01 * q.w01 + (long long)dy10 * q.w10 + (long long)dy11 * q.w11, W_BITS);
            ip[e] = iv;
            ix[e] = xv;
            iy[e] = yv;
            sA11 += (long long)xv * xv;
            sA12 += (long long)xv * yv;
            sA22 += (long long)yv * yv;
        }
        const long long iA11 = wave_allsum(sA11), iA12 = wave_allsum(sA12), iA22 = wave_allsum(sA22);
        const double S = 1.0 / 1048576.0;
        const double A11 = (double)iA11 * S, A12 = (double)iA12 * S, A22 = (double)iA22 * S;
        double D = A11 * A22 - A12 * A12;
        const double dd = A11 - A22;
        const double min_eig = (A22 + A11 - sqrt(dd * dd + 4.0 * (A12 * A12))) / (double)(2 * ww * wh);
        if (min_eig < 1e-4 || D < 1.1920928955078125e-07) {
            if (l == 0) st = 0;
            continue;
        }
        D = 1.0 / D;
        nx -= hwx;
        ny -= hwy;
        float pdx = 0.0f, pdy = 0.0f;
        const int jw = ww + 1;
        // patch difference sums at the current position of the window in the next image
        auto sums_at = [&](float fx, float fy, int inx, int iny, long long &b1, long long &b2, long long &ae) {
            const Weights r = bilinear_weights(fx - (float)inx, fy - (float)iny);
            stage_patch(Jpatch, J, W, H, P, inx, iny, jw, wh + 1, lane);
            long long s1 = 0, s2 = 0, s3 = 0;
            for (int e = lane; e < npx; e += 64) {
                const int y = e / ww, x = e % ww;
                const uint8_t *r0 = Jpatch + y * jw + x;
                const int jv = (int)descale((long long)r0[0] * r.w00 + (long long)r0[1] * r.w01 + (long long)r0[jw] * r.w10 +
                                                (long long)r0[jw + 1] * r.w11,
                                            W_BITS - 5);
                const int df = jv - ip[e];
                s1 += (long long)df * ix[e];
                s2 += (long long)df * iy[e];
                s3 += df < 0 ? -df : df;
            }
            b1 = wave_allsum(s1);
            b2 = wave_allsum(s2);
            ae = wave_allsum(s3);
        };
        for (int j = 0; j < max_count; ++j) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -ww || inx >= W || iny < -wh || iny >= H) {
                if (l == 0) st = 0;
                break;
            }
            long long ib1, ib2, iae;
            sums_at(nx, ny, inx, iny, ib1, ib2, iae);
            const double b1 = (double)ib1 * S, b2 = (double)ib2 * S;
            const float ddx = (float)((A12 * b2 - A22 * b1) * D), ddy = (float)((A12 * b1 - A11 * b2) * D);
            nx += ddx;
            ny += ddy;
            if ((double)ddx * (double)ddx + (double)ddy * (double)ddy <= eps) break;
            if (j > 0 && fabsf(ddx + pdx) < 0.01f && fabsf(ddy + pdy) < 0.01f) {
                nx -= ddx * 0.5f;
                ny -= ddy * 0.5f;
                break;
            }
            pdx = ddx;
            pdy = ddy;
        }
        if (l == 0 && st) {  // L1 patch error at the final position
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -ww || inx >= W || iny < -wh || iny >= H) {
                st = 0;
            } else {
                long long ib1, ib2, iae;
                sums_at(nx, ny, inx, iny, ib1, ib2, iae);
                er = (float)((double)iae / (double)(32 * ww * wh));
            }
        }
        nx += hwx;
        ny += hwy;
    }
    if (lane == 0) {
        out[2 * i] = nx;
        out[2 * i + 1] = ny;
        status[i] = (uint8_t)st;
        err[i] = st ? er : 0.0f;
    }
}

// ---- Shi-Tomasi ---------------------------------------------------------------------------------------------------------
constexpr int ME_T = 32;        // output tile
constexpr int ME_MAX_BS = 15;
__global__ __launch_bounds__(256) void min_eig_kernel(const uint8_t *__restrict__ img, int w, int h, int p, int bs,
                                                      double *__restrict__ eig) {
    __shared__ short Dx[(ME_T + ME_MAX_BS - 1) * (ME_T + ME_MAX_BS - 1)], Dy[(ME_T + ME_MAX_BS - 1) * (ME_T + ME_MAX_BS - 1)];
    const int lo = -(bs / 2);
    const int tw = ME_T + bs - 1;
    const int ox = blockIdx.x * ME_T, oy = blockIdx.y * ME_T;
    // derivative at the REFLECTED position of every tile entry (the definition: window positions are reflected first)
    for (int e = threadIdx.x; e < tw * tw; e += 256) {
        const int r = e / tw, c = e % tw;
        const int x = refl(ox + lo + c, w), y = refl(oy + lo + r, h);
        auto px = [&](int xx, int yy) { return (int)img[(size_t)refl(yy, h) * p + refl(xx, w)]; };
        const int a = px(x - 1, y - 1), b = px(x, y - 1), cc = px(x + 1, y - 1);
        const int d = px(x - 1, y), f = px(x + 1, y);
        const int g = px(x - 1, y + 1), hh = px(x, y + 1), ii = px(x + 1, y + 1);
        Dx[e] = (short)((cc - a) + 2 * (f - d) + (ii - g));
        Dy[e] = (short)((g - a) + 2 * (hh - b) + (ii - cc));
    }
    __syncthreads();
    const double sc = 1.0 / (4.0 * (double)bs * 255.0), s2 = sc * sc;
    for (int e = threadIdx.x; e < ME_T * ME_T; e += 256) {
        const int ty = e / ME_T, tx = e % ME_T;
        const int x = ox + tx, y = oy + ty;
        if (x >= w || y >= h) continue;
        long long a = 0, b = 0, c = 0;
        for (int v = 0; v < bs; ++v)
            for (int u = 0; u < bs; ++u) {
                const int dx = Dx[(ty + v) * tw + tx + u], dy = Dy[(ty + v) * tw + tx + u];
                a += dx * dx;
                b += dx * dy;
                c += dy * dy;
            }
        const double A = 0.5 * (double)a, C = 0.5 * (double)c, B = (double)b;
        const double df = A - C;
        eig[(size_t)y * w + x] = ((A + C) - sqrt(df * df + B * B)) * s2;
    }
}

// max of a non-negative f64 map: bit patterns of non-negative doubles order like unsigned integers
__global__ __launch_bounds__(256) void max_f64_kernel(const double *__restrict__ v, size_t n, unsigned long long *__restrict__ out) {
    unsigned long long m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double x = v[i];
        if (x > 0.0) m = max(m, (unsigned long long)__double_as_longlong(x));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned long long)__shfl_down((long long)m, off, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// candidates: v > quality * max, v != 0, no thresholded 3x3 neighbour larger; interior pixels only
__global__ __launch_bounds__(256) void corner_candidates_kernel(const double *__restrict__ eig, int w, int h, double quality,
                                                                const unsigned long long *__restrict__ max_bits,
                                                                long long *__restrict__ val_bits, int32_t *__restrict__ pos,
                                                                int cap, int32_t *__restrict__ count) {
    const double thr = __longlong_as_double((long long)max_bits[0]) * quality;
    const int x = 1 + blockIdx.x * 64 + (threadIdx.x & 63), y = 1 + blockIdx.y * 4 + (threadIdx.x >> 6);
    bool keep = false;
    double v = 0.0;
    if (x < w - 1 && y < h - 1) {
        v = eig[(size_t)y * w + x];
        if (v > thr && v != 0.0) {
            keep = true;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    double q = eig[(size_t)(y + dy) * w + x + dx];
                    if (!(q > thr)) q = 0.0;
                    if (q > v) keep = false;
                }
        }
    }
    const unsigned long long m = __ballot(keep);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, 0, 64);
    if (keep) {
        const int slot = base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1));
        if (slot < cap) {
            val_bits[slot] = __double_as_longlong(v);
            pos[slot] = y * w + x;
        }
    }
}

}  // namespace

extern "C" int mm_pyr_down(mm_ctx *ctx, const uint8_t *src, int w, int h, int src_pitch, uint8_t *dst, int dst_pitch) {
    if (!ctx) return MM_ERR_ARG;
    if (!src || !dst || w < 1 || h < 1 || src_pitch < w || dst_pitch < (w + 1) / 2) return mm_fail(ctx, MM_ERR_ARG, "mm_pyr_down: bad argument");
    const int wd = (w + 1) / 2, hd = (h + 1) / 2;
    MM_LAUNCH(ctx, "pyr_down_kernel", pyr_down_kernel, dim3((wd + PD_TW - 1) / PD_TW, (hd + PD_TH - 1) / PD_TH), dim3(256), 0, src, w,
              h, src_pitch, dst, wd, hd, dst_pitch);
    return MM_OK;
}

extern "C" int mm_lk_track(mm_ctx *ctx, const uint8_t *const *prev_levels, const uint8_t *const *next_levels, const int *w,
                           const int *h, const int *pitch, int levels, const float *pts, int n, int win_w, int win_h,
                           int max_count, double epsilon_sq, float *next_pts, uint8_t *status, float *err) {
    if (!ctx) return MM_ERR_ARG;
    if (n == 0) return MM_OK;
    if (!prev_levels || !next_levels || !w || !h || !pitch || levels < 1 || levels > LK_MAX_LEVELS || !pts || n < 0 || win_w < 3 ||
        win_h < 3 || win_w > LK_MAX_WIN || win_h > LK_MAX_WIN || max_count < 0 || !next_pts || !status || !err)
        return mm_fail(ctx, MM_ERR_ARG, "mm_lk_track: bad argument (1..8 levels, window 3..41)");
    LkLevels L = {};
    L.levels = levels;
    for (int l = 0; l < levels; ++l) {
        if (!prev_levels[l] || !next_levels[l] || w[l] < 1 || h[l] < 1 || pitch[l] < w[l]) return mm_fail(ctx, MM_ERR_ARG, "mm_lk_track: bad level %d", l);
        L.prev[l] = prev_levels[l];
        L.next[l] = next_levels[l];
        L.w[l] = w[l];
        L.h[l] = h[l];
        L.p[l] = pitch[l];
    }
    const size_t lds = (size_t)3 * win_w * win_h * sizeof(int) + ((size_t)(win_w + 3) * (win_h + 3) + 3) / 4 * 4 + (size_t)(win_w + 1) * (win_h + 1) + 16;
    MM_LAUNCH(ctx, "lk_track_kernel", lk_track_kernel, dim3(n), dim3(64), lds, L, pts, n, win_w, win_h, max_count, epsilon_sq, next_pts,
              status, err);
    return MM_OK;
}

extern "C" int mm_min_eig(mm_ctx *ctx, const uint8_t *img, int w, int h, int pitch, int block_size, double *eig) {
    if (!ctx) return MM_ERR_ARG;
    if (!img || !eig || w < 1 || h < 1 || pitch < w || block_size < 1 || block_size > ME_MAX_BS)
        return mm_fail(ctx, MM_ERR_ARG, "mm_min_eig: bad argument (block size 1..15)");
    MM_LAUNCH(ctx, "min_eig_kernel", min_eig_kernel, dim3((w + ME_T - 1) / ME_T, (h + ME_T - 1) / ME_T), dim3(256), 0, img, w, h, pitch,
              block_size, eig);
    return MM_OK;
}

// candidates of goodFeaturesToTrack: value bit patterns (positive doubles) + flat positions, unordered; count [1] is the
// number found (may exceed cap: only the first cap slots are written).  max_bits [1]: workspace, receives the map maximum.
extern "C" int mm_corner_candidates(mm_ctx *ctx, const double *eig, int w, int h, double quality, unsigned long long *max_bits,
                                    long long *val_bits, int32_t *pos, int cap, int32_t *count) {
    if (!ctx) return MM_ERR_ARG;
    if (!eig || !max_bits || !val_bits || !pos || !count || w < 3 || h < 3 || cap < 0) return mm_fail(ctx, MM_ERR_ARG, "mm_corner_candidates: bad argument");
    MM_HIP(ctx, hipMemsetAsync(max_bits, 0, sizeof(unsigned long long), ctx->stream));
    MM_HIP(ctx, hipMemsetAsync(count, 0, sizeof(int32_t), ctx->stream));
    MM_LAUNCH(ctx, "max_f64_kernel", max_f64_kernel, dim3(256), dim3(256), 0, eig, (size_t)w * h, max_bits);
    MM_LAUNCH(ctx, "corner_candidates_kernel", corner_candidates_kernel, dim3((w - 2 + 63) / 64, (h - 2 + 3) / 4), dim3(256), 0, eig, w, h,
              quality, (const unsigned long long *)max_bits, val_bits, pos, cap, count);
    return MM_OK;
}
