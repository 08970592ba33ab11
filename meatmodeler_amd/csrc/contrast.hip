// increaseContrast on the device (gfx950): BGR -> L*a*b* -> CLAHE on L -> BGR, and the grey conversion that follows it.
//
// Replaces processor.increaseContrast (reference processor.py:12-26: cv2.cvtColor(BGR2LAB), cv2.createCLAHE(3.5, (8, 8))
// .apply(L), cv2.cvtColor(Lab2BGR)) and cv2.cvtColor(frame, COLOR_BGR2GRAY) at processor.py:357.  OpenCV is absent
// offline; the arithmetic is DEFINED by oracle/frame_oracle.c (table-driven fixed-point LAB with the tables of
// meatmodeler_amd/frame_tables.py, OpenCV's published CLAHE: clip, redistribute, cumulative LUT, bilinear blending of
// the four neighbouring tiles' LUTs) and reproduced here bit for bit (integer per-pixel arithmetic; the float LUT
// blending has a fixed operation order and this file is compiled with -ffp-contract=off).
//
// Three streaming passes per frame, all HBM bound (3 B in + 3 B out per pixel, L / a / b planes 3 B written + read):
//   lab_forward   BGR -> L, a, b planes
//   clahe_lut     one workgroup per tile: 256-bin LDS histogram (integer atomics: exact), clip / redistribute / scan
//   clahe_apply   L' = blend of 4 LUTs, L'a'b -> BGR (+ grey of the result, optional)
#include "mm_common.h"

namespace {

__device__ __forceinline__ int refl(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int clamp4095(long long v) { return v < 0 ? 0 : (v > 4095 ? 4095 : (int)v); }
__device__ __forceinline__ int div_round(long long num, long long den) {
    return (int)(num >= 0 ? (num + den / 2) / den : -((-num + den / 2) / den));
}
__device__ __forceinline__ int lab_finv(int f) {
    long long t;
    if (f > 6779)
        t = ((long long)f * f * f * 4095 + (1ll << 44)) >> 45;
    else
        t = ((long long)(f - 4520) * 269254 + (1ll << 23)) >> 24;
    return clamp4095(t);
}

struct LabTables {
    const uint16_t *gamma, *cbrt_tab;
    const uint8_t *gamma_inv;
};

__device__ __forceinline__ void lab_forward_px(const LabTables &T, int B, int G, int R, int &L, int &A, int &Bb) {
    const int r = T.gamma[R], g = T.gamma[G], b = T.gamma[B];
    int X = (1777 * r + 1541 * g + 778 * b + 2048) >> 12;
    int Y = (871 * r + 2929 * g + 296 * b + 2048) >> 12;
    int Z = (73 * r + 448 * g + 3575 * b + 2048) >> 12;
    X = X > 4095 ? 4095 : X;
    Y = Y > 4095 ? 4095 : Y;
    Z = Z > 4095 ? 4095 : Z;
    const int fx = T.cbrt_tab[X], fy = T.cbrt_tab[Y], fz = T.cbrt_tab[Z];
    L = clamp255((int)(((long long)(116 * fy - 16 * 32768) * 255 + 50 * 32768) / (100 * 32768)));
    A = clamp255((int)((500 * (long long)(fx - fy) + 128 * 32768 + 16384) >> 15));
    Bb = clamp255((int)((200 * (long long)(fy - fz) + 128 * 32768 + 16384) >> 15));
}

// four pixels per thread: 12 bytes in as three dwords, one dword out per plane (n4 = number of pixel quads; the tail of
// a pixel count that is not a multiple of four takes the scalar path in the same kernel)
__global__ __launch_bounds__(256) void lab_forward_kernel(LabTables T, const uint8_t *__restrict__ bgr, size_t n,
                                                          uint8_t *__restrict__ Lp, uint8_t *__restrict__ Ap,
                                                          uint8_t *__restrict__ Bp) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t i0 = q * 4;
    if (i0 >= n) return;
    if (i0 + 4 <= n && (((uintptr_t)bgr | (uintptr_t)Lp | (uintptr_t)Ap | (uintptr_t)Bp) & 3) == 0) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(bgr + 3 * i0);
        const uint32_t wv[3] = {src[0], src[1], src[2]};
        auto byte_at = [&](int idx) { return (int)((wv[idx >> 2] >> (8 * (idx & 3))) & 255u); };
        int Lq[4], Aq[4], Bq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) lab_forward_px(T, byte_at(3 * k), byte_at(3 * k + 1), byte_at(3 * k + 2), Lq[k], Aq[k], Bq[k]);
        *reinterpret_cast<uchar4 *>(Lp + i0) = make_uchar4((unsigned char)Lq[0], (unsigned char)Lq[1], (unsigned char)Lq[2], (unsigned char)Lq[3]);
        *reinterpret_cast<uchar4 *>(Ap + i0) = make_uchar4((unsigned char)Aq[0], (unsigned char)Aq[1], (unsigned char)Aq[2], (unsigned char)Aq[3]);
        *reinterpret_cast<uchar4 *>(Bp + i0) = make_uchar4((unsigned char)Bq[0], (unsigned char)Bq[1], (unsigned char)Bq[2], (unsigned char)Bq[3]);
        return;
    }
    for (size_t i = i0; i < n && i < i0 + 4; ++i) {
        int L, A, Bb;
        lab_forward_px(T, bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2], L, A, Bb);
        Lp[i] = (uint8_t)L;
        Ap[i] = (uint8_t)A;
        Bp[i] = (uint8_t)Bb;
    }
}

// one workgroup per (tile, frame): histogram of the tile (the image is padded by reflection to a multiple of the grid),
// clip, redistribute, cumulative LUT
__global__ __launch_bounds__(256) void clahe_lut_kernel(const uint8_t *__restrict__ Lp, int w, int h, int tx, int ty, int tw,
                                                        int th, int climit, float lut_scale, uint8_t *__restrict__ lut) {
    __shared__ int hist[4][256];   // one per wave: a quarter of the same-bin collisions of a smooth tile
    __shared__ int scan[256];
    __shared__ int s_clipped;
    const int i = blockIdx.x % tx, j = blockIdx.x / tx;
    const uint8_t *img = Lp + (size_t)blockIdx.y * w * h;
#pragma unroll
    for (int k = 0; k < 4; ++k) hist[k][threadIdx.x] = 0;
    if (threadIdx.x == 0) s_clipped = 0;
    __syncthreads();
    int *myh = hist[threadIdx.x >> 6];
    // threads walk the tile's rows (column = thread, several column passes for tiles wider than 256), sixteen rows per
    // trip so that sixteen loads are in flight before the first LDS atomic needs its value (the pass is latency bound)
    for (int xb = threadIdx.x; xb < tw; xb += 256) {
        const int xs = refl(i * tw + xb, w);
        int y = 0;
        for (; y + 16 <= th; y += 16) {
            int v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = img[(size_t)refl(j * th + y + k, h) * w + xs];
            const unsigned long long active = __ballot(1);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                // LDS atomics on ONE address serialise (~16 cycles each: a flat region cost 64 of them per row and wave);
                // a row segment of one value is counted by a single add instead
                const int v0 = __builtin_amdgcn_readfirstlane(v[k]);
                if (__ballot(v[k] == v0) == active) {
                    if ((int)(threadIdx.x & 63) == __builtin_ctzll(active)) atomicAdd(&myh[v0], __popcll(active));
                } else {
                    atomicAdd(&myh[v[k]], 1);
                }
            }
        }
        for (; y < th; ++y) atomicAdd(&myh[img[(size_t)refl(j * th + y, h) * w + xs]], 1);
    }
    __syncthreads();
    int v = (hist[0][threadIdx.x] + hist[1][threadIdx.x]) + (hist[2][threadIdx.x] + hist[3][threadIdx.x]);
    if (v > climit) {
        atomicAdd(&s_clipped, v - climit);
        v = climit;
    }
    __syncthreads();
    const int clipped = s_clipped;
    const int batch = clipped / 256;
    const int residual = clipped - batch * 256;
    v += batch;
    if (residual != 0) {
        int step = 256 / residual;
        if (step < 1) step = 1;
        // bins 0, step, 2 step, ... get one more each, `residual` of them (while the bin index stays below 256)
        if ((int)threadIdx.x % step == 0 && (int)threadIdx.x / step < residual) v += 1;
    }
    scan[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // inclusive scan
        const int t = threadIdx.x >= (unsigned)off ? scan[threadIdx.x - off] : 0;
        __syncthreads();
        scan[threadIdx.x] += t;
        __syncthreads();
    }
    const int q = __float2int_rn((float)scan[threadIdx.x] * lut_scale);
    lut[(((size_t)blockIdx.y * ty + j) * tx + i) * 256 + threadIdx.x] = (uint8_t)clamp255(q);
}

__device__ __forceinline__ void clahe_apply_px(const LabTables &T, const uint8_t *__restrict__ lt, int tx, int x, int v, int A,
                                               int Bb, int y1, int y2, float ya, float ya1, float inv_tw, int &B, int &G, int &R) {
    const float txf = (float)x * inv_tw - 0.5f;
    int x1 = (int)floorf(txf), x2 = x1 + 1;
    const float xa = txf - (float)x1, xa1 = 1.0f - xa;
    x1 = x1 < 0 ? 0 : x1;
    x2 = x2 > tx - 1 ? tx - 1 : x2;
    const float r = ((float)lt[((size_t)y1 * tx + x1) * 256 + v] * xa1 + (float)lt[((size_t)y1 * tx + x2) * 256 + v] * xa) * ya1 +
                    ((float)lt[((size_t)y2 * tx + x1) * 256 + v] * xa1 + (float)lt[((size_t)y2 * tx + x2) * 256 + v] * xa) * ya;
    const int L = clamp255(__float2int_rn(r));
    const int fy = div_round(((long long)L * 100 + 16 * 255) * 32768, 116 * 255);
    const int fx = fy + div_round((long long)(A - 128) * 65536, 1000);
    const int fz = fy - div_round((long long)(Bb - 128) * 16384, 100);
    const int X = lab_finv(fx), Y = lab_finv(fy), Z = lab_finv(fz);
    const int rr = clamp4095((12621 * X - 6300 * Y - 2225 * Z + 2048) >> 12);
    const int gg = clamp4095((-3775 * X + 7686 * Y + 185 * Z + 2048) >> 12);
    const int bb = clamp4095((215 * X - 834 * Y + 4715 * Z + 2048) >> 12);
    R = T.gamma_inv[rr];
    G = T.gamma_inv[gg];
    B = T.gamma_inv[bb];
}

// four pixels of one row per thread (dword loads of the planes, three dwords of BGR + one of grey out); rows whose
// width is not a multiple of four finish with single pixels
__global__ __launch_bounds__(256) void clahe_apply_kernel(LabTables T, const uint8_t *__restrict__ Lp,
                                                          const uint8_t *__restrict__ Ap, const uint8_t *__restrict__ Bp,
                                                          int w, int h, int tx, int ty, float inv_tw, float inv_th,
                                                          const uint8_t *__restrict__ lut, uint8_t *__restrict__ out,
                                                          uint8_t *__restrict__ grey) {
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= w) return;
    const size_t fo = (size_t)blockIdx.z * w * h, i0 = fo + (size_t)y * w + x0;
    const uint8_t *lt = lut + (size_t)blockIdx.z * tx * ty * 256;
    const float tyf = (float)y * inv_th - 0.5f;
    int y1 = (int)floorf(tyf), y2 = y1 + 1;
    const float ya = tyf - (float)y1, ya1 = 1.0f - ya;
    y1 = y1 < 0 ? 0 : y1;
    y2 = y2 > ty - 1 ? ty - 1 : y2;
    const bool quad = x0 + 4 <= w && (i0 & 3) == 0 &&
                      (((uintptr_t)Lp | (uintptr_t)Ap | (uintptr_t)Bp | (uintptr_t)out | (uintptr_t)grey) & 3) == 0;
    if (quad) {
        const uint32_t l4 = *reinterpret_cast<const uint32_t *>(Lp + i0), a4 = *reinterpret_cast<const uint32_t *>(Ap + i0),
                       b4 = *reinterpret_cast<const uint32_t *>(Bp + i0);
        unsigned char o[12], gq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int B, G, R;
            clahe_apply_px(T, lt, tx, x0 + k, (l4 >> (8 * k)) & 255, (a4 >> (8 * k)) & 255, (b4 >> (8 * k)) & 255, y1, y2, ya, ya1,
                           inv_tw, B, G, R);
            o[3 * k] = (unsigned char)B;
            o[3 * k + 1] = (unsigned char)G;
            o[3 * k + 2] = (unsigned char)R;
            gq[k] = (unsigned char)((B * 1868 + G * 9617 + R * 4899 + 8192) >> 14);
        }
        uchar4 *dst = reinterpret_cast<uchar4 *>(out + 3 * i0);
        dst[0] = make_uchar4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uchar4(o[4], o[5], o[6], o[7]);
        dst[2] = make_uchar4(o[8], o[9], o[10], o[11]);
        if (grey) *reinterpret_cast<uchar4 *>(grey + i0) = make_uchar4(gq[0], gq[1], gq[2], gq[3]);
        return;
    }
    for (int k = 0; k < 4 && x0 + k < w; ++k) {
        const size_t i = i0 + k;
        int B, G, R;
        clahe_apply_px(T, lt, tx, x0 + k, Lp[i], Ap[i], Bp[i], y1, y2, ya, ya1, inv_tw, B, G, R);
        out[3 * i] = (uint8_t)B;
        out[3 * i + 1] = (uint8_t)G;
        out[3 * i + 2] = (uint8_t)R;
        if (grey) grey[i] = (uint8_t)((B * 1868 + G * 9617 + R * 4899 + 8192) >> 14);
    }
}

__global__ __launch_bounds__(256) void bgr_to_grey_kernel(const uint8_t *__restrict__ bgr, size_t n, uint8_t *__restrict__ grey) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    grey[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + 8192) >> 14);
}

}  // namespace

extern "C" size_t mm_contrast_workspace_bytes(int batch, int w, int h, int tiles_x, int tiles_y) {
    if (batch < 1 || w < 1 || h < 1 || tiles_x < 1 || tiles_y < 1) return 0;
    return mm_align_up((size_t)batch * w * h, 256) * 3 + mm_align_up((size_t)batch * tiles_x * tiles_y * 256, 256);
}

// bgr [batch,h,w,3] u8 -> out [batch,h,w,3] u8 (and grey [batch,h,w] of the result when grey != NULL).
// tables: gamma [256] u16, cbrt_tab [4096] u16, gamma_inv [4096] u8 (device).
extern "C" int mm_increase_contrast(mm_ctx *ctx, const uint8_t *bgr, int batch, int w, int h, const uint16_t *gamma,
                                    const uint16_t *cbrt_tab, const uint8_t *gamma_inv, double clip_limit, int tiles_x,
                                    int tiles_y, uint8_t *out, uint8_t *grey, void *ws, size_t ws_bytes) {
    if (!ctx) return MM_ERR_ARG;
    if (batch == 0) return MM_OK;
    if (!bgr || !out || !gamma || !cbrt_tab || !gamma_inv || batch < 0 || w < 1 || h < 1 || tiles_x < 1 || tiles_y < 1 ||
        tiles_x > w || tiles_y > h || !(clip_limit > 0.0))
        return mm_fail(ctx, MM_ERR_ARG, "mm_increase_contrast: bad argument");
    if (!ws || ws_bytes < mm_contrast_workspace_bytes(batch, w, h, tiles_x, tiles_y))
        return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_increase_contrast: workspace too small");
    const size_t plane = mm_align_up((size_t)batch * w * h, 256), n = (size_t)batch * w * h;
    uint8_t *Lp = (uint8_t *)ws, *Ap = Lp + plane, *Bp = Ap + plane, *lut = Bp + plane;
    LabTables T = {gamma, cbrt_tab, gamma_inv};
    MM_LAUNCH(ctx, "lab_forward_kernel", lab_forward_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, T, bgr, n, Lp, Ap, Bp);
    const int tw = (w + tiles_x - 1) / tiles_x, th = (h + tiles_y - 1) / tiles_y;
    int climit = (int)(clip_limit * (double)(tw * th) / 256.0);
    if (climit < 1) climit = 1;
    const float lut_scale = 255.0f / (float)(tw * th);
    MM_LAUNCH(ctx, "clahe_lut_kernel", clahe_lut_kernel, dim3(tiles_x * tiles_y, batch), dim3(256), 0, (const uint8_t *)Lp, w, h,
              tiles_x, tiles_y, tw, th, climit, lut_scale, lut);
    MM_LAUNCH(ctx, "clahe_apply_kernel", clahe_apply_kernel, dim3((w + 1023) / 1024, h, batch), dim3(256), 0, T, (const uint8_t *)Lp,
              (const uint8_t *)Ap, (const uint8_t *)Bp, w, h, tiles_x, tiles_y, 1.0f / (float)tw, 1.0f / (float)th,
              (const uint8_t *)lut, out, grey);
    return MM_OK;
}

extern "C" int mm_bgr_to_grey(mm_ctx *ctx, const uint8_t *bgr, size_t n_pixels, uint8_t *grey) {
    if (!ctx) return MM_ERR_ARG;
    if (n_pixels == 0) return MM_OK;
    if (!bgr || !grey) return mm_fail(ctx, MM_ERR_ARG, "mm_bgr_to_grey: null pointer");
    MM_LAUNCH(ctx, "bgr_to_grey_kernel", bgr_to_grey_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, bgr, n_pixels, grey);
    return MM_OK;
}
