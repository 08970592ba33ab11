/* CPU restatement of the per-frame front end that surrounds the hot path:
 *   - pyramidal Lucas-Kanade tracking + Shi-Tomasi corners, the two OpenCV calls of `keyframeTracking`
 *     (reference processor.py:61-110: cv2.calcOpticalFlowPyrLK :79, cv2.goodFeaturesToTrack :104);
 *   - CLAHE on the L channel of LAB and the grey conversion (reference processor.py:12-26 `increaseContrast`, :357).
 *
 * TEST INFRASTRUCTURE ONLY: this file is the checker for the HIP kernels in meatmodeler_amd/csrc/flow.hip and
 * contrast.hip.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product never
 * does.
 *
 * PARITY UNPINNED: OpenCV (opencv-python~=4.5.2.54, reference requirements.txt:4) is not vendored, cannot be installed
 * offline, and the reference holds no fixture for these calls.  The algorithms restated here are the published ones
 * with OpenCV's documented defaults -- Bouguet's pyramidal LK (5-tap [1 4 6 4 1]/16 pyramid, Scharr derivatives,
 * 14-bit bilinear patch weights, minimum-eigenvalue rejection at 1e-4, L1 patch error / (32 w h)), Shi-Tomasi minimum
 * eigenvalue of the Sobel structure tensor with quality threshold, 3x3 non-maximum suppression and greedy minimum
 * distance, CLAHE with clip-and-redistribute histograms and bilinear LUT blending -- made INTEGER EXACT (all window sums
 * are 64-bit integers, the few floating-point operations that follow have a fixed order and no contraction) so that the
 * GPU result can be compared bit for bit.  Implementation-defined points of OpenCV fixed here: border handling is
 * reflect-101 everywhere, ties between equally strong corners go to the smaller (y, x), the LAB conversion is a
 * table-driven fixed-point one whose tables come from meatmodeler_amd/frame_tables.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int refl(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}
static inline int64_t descale(int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; }

/* ---- pyramid ------------------------------------------------------------------------------------------------------- */
void frc_pyr_down(const uint8_t *src, int w, int h, int ps, uint8_t *dst, int pd) {
    static const int k[5] = {1, 4, 6, 4, 1};
    const int wd = (w + 1) / 2, hd = (h + 1) / 2;
    for (int y = 0; y < hd; ++y)
        for (int x = 0; x < wd; ++x) {
            int s = 0;
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx)
                    s += k[dy + 2] * k[dx + 2] * src[(size_t)refl(2 * y + dy, h) * ps + refl(2 * x + dx, w)];
            dst[(size_t)y * pd + x] = (uint8_t)((s + 128) >> 8);
        }
}

static inline int pix(const uint8_t *img, int w, int h, int p, int x, int y) { return img[(size_t)refl(y, h) * p + refl(x, w)]; }
/* Scharr derivatives at integer position (x, y).  Inside the image the taps reflect (101) at the edges; OUTSIDE it the
 * derivative is zero: OpenCV pads the derivative image of calcOpticalFlowPyrLK with BORDER_CONSTANT zeros, only the
 * intensity pyramid continues by reflection (lkpyramid.cpp: calcSharrDeriv + copyMakeBorder(..., BORDER_CONSTANT)). */
static inline void scharr(const uint8_t *img, int w, int h, int p, int x, int y, int *dx, int *dy) {
    if (x < 0 || x >= w || y < 0 || y >= h) {
        *dx = *dy = 0;
        return;
    }
    const int a = pix(img, w, h, p, x - 1, y - 1), b = pix(img, w, h, p, x, y - 1), c = pix(img, w, h, p, x + 1, y - 1);
    const int d = pix(img, w, h, p, x - 1, y), f = pix(img, w, h, p, x + 1, y);
    const int g = pix(img, w, h, p, x - 1, y + 1), hh = pix(img, w, h, p, x, y + 1), i = pix(img, w, h, p, x + 1, y + 1);
    *dx = 3 * (c - a) + 10 * (f - d) + 3 * (i - g);
    *dy = 3 * (g - a) + 10 * (hh - b) + 3 * (i - c);
}

/* ---- pyramidal Lucas-Kanade ----------------------------------------------------------------------------------------
 * prev / next pyramids: level l image at img[l] with size (w[l], h[l]) and pitch p[l].  pts [n,2] f32 in level-0 pixels.
 * out: next [n,2] f32, status [n] u8, err [n] f32.  win (ww, wh), levels = maxLevel + 1, max_count, eps = epsilon^2. */
#define W_BITS 14
void frc_lk_track(const uint8_t *const *prev, const uint8_t *const *next, const int *w, const int *h, const int *p,
                  int levels, const float *pts, int n, int ww, int wh, int max_count, double eps, float *out,
                  uint8_t *status, float *err) {
    const float hwx = (ww - 1) * 0.5f, hwy = (wh - 1) * 0.5f;
    int *ip = (int *)malloc(sizeof(int) * 3 * ww * wh), *ix = ip + ww * wh, *iy = ix + ww * wh;
    for (int i = 0; i < n; ++i) {
        float nx = 0, ny = 0;
        int st = 1;
        float er = 0.0f;
        for (int l = levels - 1; l >= 0; --l) {
            const float sc = 1.0f / (float)(1 << l);
            float px = pts[2 * i] * sc, py = pts[2 * i + 1] * sc;
            if (l == levels - 1) {
                nx = px;
                ny = py;
            } else {
                nx = nx * 2.0f;
                ny = ny * 2.0f;
            }
            const uint8_t *I = prev[l], *J = next[l];
            const int W = w[l], H = h[l], P = p[l];
            px -= hwx;
            py -= hwy;
            int ipx = (int)floorf(px), ipy = (int)floorf(py);
            if (ipx < -ww || ipx >= W || ipy < -wh || ipy >= H) {
                if (l == 0) {
                    st = 0;
                    er = 0.0f;
                }
                continue;
            }
            float a = px - (float)ipx, b = py - (float)ipy;
            int iw00 = (int)lrintf((1.0f - a) * (1.0f - b) * (float)(1 << W_BITS));
            int iw01 = (int)lrintf(a * (1.0f - b) * (float)(1 << W_BITS));
            int iw10 = (int)lrintf((1.0f - a) * b * (float)(1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t iA11 = 0, iA12 = 0, iA22 = 0;
            for (int y = 0; y < wh; ++y)
                for (int x = 0; x < ww; ++x) {
                    const int X = ipx + x, Y = ipy + y;
                    int dx00, dy00, dx01, dy01, dx10, dy10, dx11, dy11;
                    scharr(I, W, H, P, X, Y, &dx00, &dy00);
                    scharr(I, W, H, P, X + 1, Y, &dx01, &dy01);
                    scharr(I, W, H, P, X, Y + 1, &dx10, &dy10);
                    scharr(I, W, H, P, X + 1, Y + 1, &dx11, &dy11);
                    const int iv = (int)descale((int64_t)pix(I, W, H, P, X, Y) * iw00 + (int64_t)pix(I, W, H, P, X + 1, Y) * iw01 +
                                                    (int64_t)pix(I, W, H, P, X, Y + 1) * iw10 + (int64_t)pix(I, W, H, P, X + 1, Y + 1) * iw11,
                                                W_BITS - 5);
                    const int xv = (int)descale((int64_t)dx00 * iw00 + (int64_t)dx01 * iw01 + (int64_t)dx10 * iw10 + (int64_t)dx11 * iw11, W_BITS);
                    const int yv = (int)descale((int64_t)dy00 * iw00 + (int64_t)dy01 * iw01 + (int64_t)dy10 * iw10 + (int64_t)dy11 * iw11, W_BITS);
                    ip[y * ww + x] = iv;
                    ix[y * ww + x] = xv;
                    iy[y * ww + x] = yv;
                    iA11 += (int64_t)xv * xv;
                    iA12 += (int64_t)xv * yv;
                    iA22 += (int64_t)yv * yv;
                }
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
            float pdx = 0, pdy = 0;
            for (int j = 0; j < max_count; ++j) {
                const int inx = (int)floorf(nx), iny = (int)floorf(ny);
                if (inx < -ww || inx >= W || iny < -wh || iny >= H) {
                    if (l == 0) st = 0;
                    break;
                }
                a = nx - (float)inx;
                b = ny - (float)iny;
                iw00 = (int)lrintf((1.0f - a) * (1.0f - b) * (float)(1 << W_BITS));
                iw01 = (int)lrintf(a * (1.0f - b) * (float)(1 << W_BITS));
                iw10 = (int)lrintf((1.0f - a) * b * (float)(1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t ib1 = 0, ib2 = 0;
                for (int y = 0; y < wh; ++y)
                    for (int x = 0; x < ww; ++x) {
                        const int X = inx + x, Y = iny + y;
                        const int jv = (int)descale((int64_t)pix(J, W, H, P, X, Y) * iw00 + (int64_t)pix(J, W, H, P, X + 1, Y) * iw01 +
                                                        (int64_t)pix(J, W, H, P, X, Y + 1) * iw10 + (int64_t)pix(J, W, H, P, X + 1, Y + 1) * iw11,
                                                    W_BITS - 5);
                        const int df = jv - ip[y * ww + x];
                        ib1 += (int64_t)df * ix[y * ww + x];
                        ib2 += (int64_t)df * iy[y * ww + x];
                    }
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
            if (l == 0 && st) { /* L1 patch error at the final position */
                const int inx = (int)floorf(nx), iny = (int)floorf(ny);
                if (inx < -ww || inx >= W || iny < -wh || iny >= H) {
                    st = 0;
                } else {
                    a = nx - (float)inx;
                    b = ny - (float)iny;
                    iw00 = (int)lrintf((1.0f - a) * (1.0f - b) * (float)(1 << W_BITS));
                    iw01 = (int)lrintf(a * (1.0f - b) * (float)(1 << W_BITS));
                    iw10 = (int)lrintf((1.0f - a) * b * (float)(1 << W_BITS));
                    iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                    int64_t e = 0;
                    for (int y = 0; y < wh; ++y)
                        for (int x = 0; x < ww; ++x) {
                            const int X = inx + x, Y = iny + y;
                            const int jv = (int)descale((int64_t)pix(J, W, H, P, X, Y) * iw00 + (int64_t)pix(J, W, H, P, X + 1, Y) * iw01 +
                                                            (int64_t)pix(J, W, H, P, X, Y + 1) * iw10 + (int64_t)pix(J, W, H, P, X + 1, Y + 1) * iw11,
                                                        W_BITS - 5);
                            const int df = jv - ip[y * ww + x];
                            e += df < 0 ? -df : df;
                        }
                    er = (float)((double)e / (double)(32 * ww * wh));
                }
            }
            nx += hwx;
            ny += hwy;
        }
        out[2 * i] = nx;
        out[2 * i + 1] = ny;
        status[i] = (uint8_t)st;
        err[i] = st ? er : 0.0f;
    }
    free(ip);
}

/* ---- Shi-Tomasi corners -------------------------------------------------------------------------------------------------
 * eig [h][w] f64: minimum eigenvalue of the block_size x block_size Sobel structure tensor, scaled like OpenCV's
 * cornerMinEigenVal on 8-bit input (derivative scale 1 / (4 block_size 255)). */
static inline void sobel(const uint8_t *img, int w, int h, int p, int x, int y, int *dx, int *dy) {
    const int a = pix(img, w, h, p, x - 1, y - 1), b = pix(img, w, h, p, x, y - 1), c = pix(img, w, h, p, x + 1, y - 1);
    const int d = pix(img, w, h, p, x - 1, y), f = pix(img, w, h, p, x + 1, y);
    const int g = pix(img, w, h, p, x - 1, y + 1), hh = pix(img, w, h, p, x, y + 1), i = pix(img, w, h, p, x + 1, y + 1);
    *dx = (c - a) + 2 * (f - d) + (i - g);
    *dy = (g - a) + 2 * (hh - b) + (i - c);
}

void frc_min_eig(const uint8_t *img, int w, int h, int p, int bs, double *eig) {
    const double sc = 1.0 / (4.0 * (double)bs * 255.0), s2 = sc * sc;
    const int lo = -(bs / 2), hi = bs - 1 - bs / 2;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int64_t a = 0, b = 0, c = 0;
            for (int v = lo; v <= hi; ++v)
                for (int u = lo; u <= hi; ++u) {
                    int dx, dy;
                    sobel(img, w, h, p, refl(x + u, w), refl(y + v, h), &dx, &dy);
                    a += (int64_t)dx * dx;
                    b += (int64_t)dx * dy;
                    c += (int64_t)dy * dy;
                }
            const double A = 0.5 * (double)a, C = 0.5 * (double)c, B = (double)b;
            const double df = A - C;
            eig[(size_t)y * w + x] = ((A + C) - sqrt(df * df + B * B)) * s2;
        }
}

typedef struct {
    double v;
    int32_t y, x;
} cand_t;
static int cmp_cand(const void *pa, const void *pb) {
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->v != b->v) return a->v > b->v ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    return a->x < b->x ? -1 : (a->x > b->x ? 1 : 0);
}

/* -> number of corners written to out [max_out,2] f32 (x, y) */
int frc_good_features(const uint8_t *img, int w, int h, int p, int max_corners, double quality, double min_distance, int bs,
                      float *out, int max_out) {
    double *eig = (double *)malloc(sizeof(double) * (size_t)w * h);
    frc_min_eig(img, w, h, p, bs, eig);
    double mx = 0.0;
    for (size_t i = 0; i < (size_t)w * h; ++i)
        if (eig[i] > mx) mx = eig[i];
    const double thr = mx * quality;
    cand_t *cd = (cand_t *)malloc(sizeof(cand_t) * ((size_t)w * h + 1)); /* (a plateau of equal values keeps every pixel) */
    size_t nc = 0;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            const double v = eig[(size_t)y * w + x];
            if (!(v > thr) || v == 0.0) continue;
            int keep = 1;
            for (int dy = -1; dy <= 1 && keep; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    double q = eig[(size_t)(y + dy) * w + x + dx];
                    if (!(q > thr)) q = 0.0; /* (thresholded map) */
                    if (q > v) {
                        keep = 0;
                        break;
                    }
                }
            if (keep) {
                cd[nc].v = v;
                cd[nc].y = y;
                cd[nc].x = x;
                ++nc;
            }
        }
    qsort(cd, nc, sizeof(cand_t), cmp_cand);
    int n = 0;
    if (min_distance >= 1.0) {
        const int cell = (int)lrint(min_distance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        int32_t *head = (int32_t *)malloc(sizeof(int32_t) * (size_t)gw * gh);
        int32_t *nxt = (int32_t *)malloc(sizeof(int32_t) * (nc + 1));
        for (size_t i = 0; i < (size_t)gw * gh; ++i) head[i] = -1;
        const double md2 = min_distance * min_distance;
        int32_t *acc = (int32_t *)malloc(sizeof(int32_t) * (nc + 1));
        for (size_t i = 0; i < nc; ++i) {
            const int x = cd[i].x, y = cd[i].y, cx = x / cell, cy = y / cell;
            int good = 1;
            for (int yy = cy > 0 ? cy - 1 : 0; yy <= (cy + 1 < gh ? cy + 1 : gh - 1) && good; ++yy)
                for (int xx = cx > 0 ? cx - 1 : 0; xx <= (cx + 1 < gw ? cx + 1 : gw - 1) && good; ++xx)
                    for (int32_t e = head[yy * gw + xx]; e >= 0; e = nxt[e]) {
                        const double dx = x - cd[acc[e]].x, dy = y - cd[acc[e]].y;
                        if (dx * dx + dy * dy < md2) {
                            good = 0;
                            break;
                        }
                    }
            if (!good) continue;
            acc[n] = (int32_t)i;
            nxt[n] = head[cy * gw + cx];
            head[cy * gw + cx] = n;
            if (n < max_out) {
                out[2 * n] = (float)x;
                out[2 * n + 1] = (float)y;
            }
            ++n;
            if ((max_corners > 0 && n >= max_corners) || n >= max_out) break;
        }
        free(head);
        free(nxt);
        free(acc);
    } else {
        for (size_t i = 0; i < nc; ++i) {
            if (n >= max_out) break;
            out[2 * n] = (float)cd[i].x;
            out[2 * n + 1] = (float)cd[i].y;
            ++n;
            if (max_corners > 0 && n >= max_corners) break;
        }
    }
    free(cd);
    free(eig);
    return n;
}

/* ---- increaseContrast: BGR -> LAB (table-driven fixed point) -> CLAHE on L -> BGR ---------------------------------------
 * Tables (meatmodeler_amd/frame_tables.py): gamma [256] u16 (sRGB decode, 12 bit), cbrt_tab [4096] u16 (f(t), Q15),
 * gamma_inv [4096] u8 (sRGB encode); f^-1 is evaluated in integer arithmetic (cube / linear branch). */
typedef struct {
    const uint16_t *gamma, *cbrt_tab;
    const uint8_t *gamma_inv;
} lab_tables_t;

static inline void bgr_to_lab(const lab_tables_t *T, int B, int G, int R, int *L, int *A, int *Bb) {
    const int r = T->gamma[R], g = T->gamma[G], b = T->gamma[B];
    /* XYZ / white (D65), coefficients in 12-bit fixed point; rows sum to 4096 */
    int X = (1777 * r + 1541 * g + 778 * b + 2048) >> 12;
    int Y = (871 * r + 2929 * g + 296 * b + 2048) >> 12;
    int Z = (73 * r + 448 * g + 3575 * b + 2048) >> 12;
    X = X > 4095 ? 4095 : X;
    Y = Y > 4095 ? 4095 : Y;
    Z = Z > 4095 ? 4095 : Z;
    const int fx = T->cbrt_tab[X], fy = T->cbrt_tab[Y], fz = T->cbrt_tab[Z]; /* f * 32768 */
    /* L* = 116 f - 16 in [0,100] -> 8 bit: L = (116 fy - 16) 255 / 100 */
    int l = (int)(((int64_t)(116 * fy - 16 * 32768) * 255 + 50 * 32768) / (100 * 32768));
    int a = (int)((500 * (int64_t)(fx - fy) + 128 * 32768 + 16384) >> 15);
    int bb = (int)((200 * (int64_t)(fy - fz) + 128 * 32768 + 16384) >> 15);
    *L = l < 0 ? 0 : (l > 255 ? 255 : l);
    *A = a < 0 ? 0 : (a > 255 ? 255 : a);
    *Bb = bb < 0 ? 0 : (bb > 255 ? 255 : bb);
}

static inline int lab_finv(int f) { /* f in Q15 -> t = f^-1(f) as a 12-bit value (0..4095), integer exact */
    int64_t t;
    if (f > 6779)
        t = ((int64_t)f * f * f * 4095 + ((int64_t)1 << 44)) >> 45;
    else
        t = ((int64_t)(f - 4520) * 269254 + ((int64_t)1 << 23)) >> 24;
    return t < 0 ? 0 : (t > 4095 ? 4095 : (int)t);
}
static inline int div_round(int64_t num, int64_t den) { /* den > 0, round half away from zero */
    return (int)(num >= 0 ? (num + den / 2) / den : -((-num + den / 2) / den));
}

static inline void lab_to_bgr(const lab_tables_t *T, int L, int A, int Bb, int *B, int *G, int *R) {
    /* fy = (L* + 16) / 116 with L* = L 100 / 255, in Q15;  fx = fy + a* / 500,  fz = fy - b* / 200 */
    const int fy = div_round(((int64_t)L * 100 + 16 * 255) * 32768, 116 * 255);
    const int fx = fy + div_round((int64_t)(A - 128) * 65536, 1000);
    const int fz = fy - div_round((int64_t)(Bb - 128) * 16384, 100);
    const int X = lab_finv(fx), Y = lab_finv(fy), Z = lab_finv(fz);
    /* inverse matrix (XYZ / white -> linear sRGB), 12-bit fixed point */
    int r = (12621 * X - 6300 * Y - 2225 * Z + 2048) >> 12;   /* rows sum to 4096: white stays white */
    int g = (-3775 * X + 7686 * Y + 185 * Z + 2048) >> 12;
    int b = (215 * X - 834 * Y + 4715 * Z + 2048) >> 12;
    r = r < 0 ? 0 : (r > 4095 ? 4095 : r);
    g = g < 0 ? 0 : (g > 4095 ? 4095 : g);
    b = b < 0 ? 0 : (b > 4095 ? 4095 : b);
    *R = T->gamma_inv[r];
    *G = T->gamma_inv[g];
    *B = T->gamma_inv[b];
}

/* CLAHE on one 8-bit plane, in place semantics: src -> dst.  tiles (tx, ty), clip limit as OpenCV (float). */
void frc_clahe(const uint8_t *src, int w, int h, int p, uint8_t *dst, int pd, int tx, int ty, double clip) {
    /* tile size on the image padded (reflect-101) to a multiple of the grid */
    const int tw = (w + tx - 1) / tx, th = (h + ty - 1) / ty;
    const int area = tw * th;
    int climit = (int)(clip * area / 256.0);
    if (climit < 1) climit = 1;
    const float lut_scale = 255.0f / (float)area;
    uint8_t *lut = (uint8_t *)malloc((size_t)tx * ty * 256);
    for (int j = 0; j < ty; ++j)
        for (int i = 0; i < tx; ++i) {
            int hist[256];
            memset(hist, 0, sizeof(hist));
            for (int y = 0; y < th; ++y)
                for (int x = 0; x < tw; ++x) hist[pix(src, w, h, p, i * tw + x, j * th + y)]++;
            int clipped = 0;
            for (int k = 0; k < 256; ++k)
                if (hist[k] > climit) {
                    clipped += hist[k] - climit;
                    hist[k] = climit;
                }
            const int batch = clipped / 256;
            int residual = clipped - batch * 256;
            for (int k = 0; k < 256; ++k) hist[k] += batch;
            if (residual != 0) {
                int step = 256 / residual;
                if (step < 1) step = 1;
                for (int k = 0; k < 256 && residual > 0; k += step, --residual) hist[k]++;
            }
            int sum = 0;
            for (int k = 0; k < 256; ++k) {
                sum += hist[k];
                long v = lrintf((float)sum * lut_scale);
                lut[((size_t)j * tx + i) * 256 + k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    for (int y = 0; y < h; ++y) {
        const float tyf = (float)y * inv_th - 0.5f;
        int y1 = (int)floorf(tyf), y2 = y1 + 1;
        const float ya = tyf - (float)y1, ya1 = 1.0f - ya;
        y1 = y1 < 0 ? 0 : y1;
        y2 = y2 > ty - 1 ? ty - 1 : y2;
        for (int x = 0; x < w; ++x) {
            const float txf = (float)x * inv_tw - 0.5f;
            int x1 = (int)floorf(txf), x2 = x1 + 1;
            const float xa = txf - (float)x1, xa1 = 1.0f - xa;
            x1 = x1 < 0 ? 0 : x1;
            x2 = x2 > tx - 1 ? tx - 1 : x2;
            const int v = src[(size_t)y * p + x];
            const float r = ((float)lut[((size_t)y1 * tx + x1) * 256 + v] * xa1 + (float)lut[((size_t)y1 * tx + x2) * 256 + v] * xa) * ya1 +
                            ((float)lut[((size_t)y2 * tx + x1) * 256 + v] * xa1 + (float)lut[((size_t)y2 * tx + x2) * 256 + v] * xa) * ya;
            long q = lrintf(r);
            dst[(size_t)y * pd + x] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
    }
    free(lut);
}

/* bgr [h][w][3] u8 -> out [h][w][3] u8: increaseContrast (reference processor.py:12-26, clipLimit 3.5, 8 x 8 tiles) */
void frc_increase_contrast(const uint8_t *bgr, int w, int h, const uint16_t *gamma, const uint16_t *cbrt_tab,
                           const uint8_t *gamma_inv, double clip, int tx, int ty, uint8_t *out) {
    lab_tables_t T = {gamma, cbrt_tab, gamma_inv};
    uint8_t *L = (uint8_t *)malloc((size_t)w * h * 4), *A = L + (size_t)w * h, *B = A + (size_t)w * h, *L2 = B + (size_t)w * h;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        int l, a, b;
        bgr_to_lab(&T, bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2], &l, &a, &b);
        L[i] = (uint8_t)l;
        A[i] = (uint8_t)a;
        B[i] = (uint8_t)b;
    }
    frc_clahe(L, w, h, w, L2, w, tx, ty, clip);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        int b, g, r;
        lab_to_bgr(&T, L2[i], A[i], B[i], &b, &g, &r);
        out[3 * i] = (uint8_t)b;
        out[3 * i + 1] = (uint8_t)g;
        out[3 * i + 2] = (uint8_t)r;
    }
    free(L);
}

/* cv2.COLOR_BGR2GRAY on 8-bit input: (B 1868 + G 9617 + R 4899 + 8192) >> 14 (reference processor.py:357) */
void frc_bgr_to_grey(const uint8_t *bgr, size_t n, uint8_t *grey) {
    for (size_t i = 0; i < n; ++i)
        grey[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + 8192) >> 14);
}

/* BGR -> LAB -> BGR without CLAHE (accuracy check of the fixed-point conversion); lab [n,3] receives L, a, b */
void frc_lab_roundtrip(const uint8_t *bgr, size_t n, const uint16_t *gamma, const uint16_t *cbrt_tab,
                       const uint8_t *gamma_inv, uint8_t *lab, uint8_t *out) {
    lab_tables_t T = {gamma, cbrt_tab, gamma_inv};
    for (size_t i = 0; i < n; ++i) {
        int l, a, b, bb, g, r;
        bgr_to_lab(&T, bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2], &l, &a, &b);
        lab[3 * i] = (uint8_t)l;
        lab[3 * i + 1] = (uint8_t)a;
        lab[3 * i + 2] = (uint8_t)b;
        lab_to_bgr(&T, l, a, b, &bb, &g, &r);
        out[3 * i] = (uint8_t)bb;
        out[3 * i + 1] = (uint8_t)g;
        out[3 * i + 2] = (uint8_t)r;
    }
}
