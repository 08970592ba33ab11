/*
 * CPU ORACLE (test infrastructure only) — plain-C restatement of the feature front end:
 * ORB detect + describe and brute-force Hamming 2-NN + Lowe ratio test.
 *
 *   *** This file is the CHECKER.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 *   *** may build, load or call it.  Nothing under meatmodeler_amd/ may.
 *
 * PARITY UNPINNED: the reference reaches these stages through OpenCV (opencv-python~=4.5.2.54,
 * /root/reference/requirements.txt:4; call sites /root/reference/processor.py:129,132-137,308,328), which is not
 * vendored, not installed offline, and the reference holds no test, fixture or golden vector for it.  This file
 * restates the published algorithms with OpenCV's default parameters (cv2.ORB_create(nfeatures): scaleFactor 1.2,
 * nlevels 8, edgeThreshold 31, HARRIS score, patchSize 31, fastThreshold 20) in exact integer arithmetic; the two
 * implementation-defined points of OpenCV are fixed as documented in DESIGN.md §mm-ORB:
 *   - selection order (response desc, y asc, x asc) instead of std::nth_element's unspecified order;
 *   - the 256 rBRIEF point pairs are a caller-supplied table (OpenCV's learned bit_pattern_31_ is unavailable).
 * It is written for clarity (direct definitions, full sorts, non-separable blur), NOT the way the HIP kernels compute.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EDGE 31
#define HALF 15

/* ---- geometry: OpenCV orb.cpp — scale_l = (float)pow(1.2, l); size = cvRound(dim / scale_l);
 *      n_l geometric with factor 1/1.2, last level takes the remainder. ---- */
void orc_level_sizes(int H, int W, int nfeatures, int nlevels, float scale_factor, int *w, int *h, int *n,
                     float *scale) {
    float factor = (float)(1.0 / (double)scale_factor);
    float ndes = (float)nfeatures * (1.0f - factor) / (1.0f - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels; ++l) {
        float sc = (float)pow((double)scale_factor, (double)l);
        scale[l] = sc;
        w[l] = l == 0 ? W : (int)rint((double)((float)W / sc));
        h[l] = l == 0 ? H : (int)rint((double)((float)H / sc));
        if (l < nlevels - 1) {
            n[l] = (int)rint((double)ndes);
            sum += n[l];
            ndes *= factor;
        } else {
            n[l] = nfeatures - sum > 0 ? nfeatures - sum : 0;
        }
    }
}

/* ---- bilinear down-scaling, 11-bit fixed-point weights (the INTER_LINEAR sample position
 *      (d + 0.5) * ns/nd - 0.5, evaluated in exact integer arithmetic). ---- */
static void axis_coord(long long d, long long nd, long long ns, int *i0, int *wgt) {
    long long num = (2 * d + 1) * ns - nd;
    if (num < 0) num = 0;
    long long q = num / (2 * nd);
    long long rem = num - q * 2 * nd;
    long long wq = (rem * 2048 + nd) / (2 * nd);
    if (q >= ns - 1) {
        q = ns - 1;
        wq = 0;
    }
    *i0 = (int)q;
    *wgt = (int)wq;
}

void orc_resize(const uint8_t *src, int ws, int hs, int ps, uint8_t *dst, int wd, int hd, int pd) {
    for (int y = 0; y < hd; ++y) {
        int y0, wy;
        axis_coord(y, hd, hs, &y0, &wy);
        int y1 = y0 + 1 < hs ? y0 + 1 : hs - 1;
        for (int x = 0; x < wd; ++x) {
            int x0, wx;
            axis_coord(x, wd, ws, &x0, &wx);
            int x1 = x0 + 1 < ws ? x0 + 1 : ws - 1;
            int top = src[(size_t)y0 * ps + x0] * (2048 - wx) + src[(size_t)y0 * ps + x1] * wx;
            int bot = src[(size_t)y1 * ps + x0] * (2048 - wx) + src[(size_t)y1 * ps + x1] * wx;
            dst[(size_t)y * pd + x] = (uint8_t)(((unsigned)(top * (2048 - wy) + bot * wy + (1 << 21))) >> 22);
        }
    }
}

/* ---- FAST-9/16: V = max over the 16 arcs of 9 contiguous circle pixels and both polarities of the minimum
 *      signed contrast along the arc; corner iff V > t; score = V - 1 (largest threshold that still fires). ---- */
static const int CIRC[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int fast_score_at(const uint8_t *img, int pitch, int x, int y, int t) {
    int p = img[(size_t)y * pitch + x];
    int d[16];
    for (int i = 0; i < 16; ++i) d[i] = (int)img[(size_t)(y + CIRC[i][1]) * pitch + x + CIRC[i][0]] - p;
    int V = -1000;
    for (int pol = -1; pol <= 1; pol += 2) {
        for (int s = 0; s < 16; ++s) {
            int mn = 1000;
            for (int j = 0; j < 9; ++j) {
                int v = pol * d[(s + j) & 15];
                if (v < mn) mn = v;
            }
            if (mn > V) V = mn;
        }
    }
    return V > t ? V - 1 : 0;
}

typedef struct {
    int x, y, score;
    long long harris;
} Cand;

static int cmp_score(const void *a, const void *b) {
    const Cand *p = a, *q = b;
    if (p->score != q->score) return q->score - p->score;
    if (p->y != q->y) return p->y - q->y;
    return p->x - q->x;
}
static int cmp_harris(const void *a, const void *b) {
    const Cand *p = a, *q = b;
    if (p->harris != q->harris) return p->harris > q->harris ? -1 : 1;
    if (p->y != q->y) return p->y - q->y;
    return p->x - q->x;
}

/* Harris measure on the 7x7 block with 3x3 Sobel derivatives (OpenCV HarrisResponses), k = 0.04 = 1/25:
 * returns exactly 25 * (a b - c^2 - k (a + b)^2). */
static long long harris25(const uint8_t *img, int pitch, int x, int y) {
    long long a = 0, b = 0, c = 0;
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx) {
            const uint8_t *q = img + (size_t)(y + dy) * pitch + (x + dx);
            int ix = (q[1] - q[-1]) * 2 + (q[-pitch + 1] - q[-pitch - 1]) + (q[pitch + 1] - q[pitch - 1]);
            int iy = (q[pitch] - q[-pitch]) * 2 + (q[pitch - 1] - q[-pitch - 1]) + (q[pitch + 1] - q[-pitch + 1]);
            a += (long long)ix * ix;
            b += (long long)iy * iy;
            c += (long long)ix * iy;
        }
    return 25 * (a * b - c * c) - (a + b) * (a + b);
}

/* OpenCV's umax table for the intensity-centroid disc (orb.cpp). */
void orc_umax(int *umax /*[HALF+2]*/) {
    int v, v0, vmax = (int)floor(HALF * sqrt(2.0) / 2 + 1);
    int vmin = (int)ceil(HALF * sqrt(2.0) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = (int)rint(sqrt((double)HALF * HALF - v * v));
    for (v = HALF, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

static const int GW[7] = {18, 33, 49, 56, 49, 33, 18};

static int blurred(const uint8_t *img, int pitch, int x, int y) {
    int s = 0;
    for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 7; ++j) s += GW[i] * GW[j] * img[(size_t)(y + i - 3) * pitch + (x + j - 3)];
    return (s + 32768) >> 16;
}

void orc_describe(const uint8_t *img, int pitch, int x, int y, const int8_t *pattern, uint8_t *desc, int *m10_out,
                  int *m01_out) {
    int umax[HALF + 2];
    orc_umax(umax);
    int m10 = 0, m01 = 0;
    for (int v = -HALF; v <= HALF; ++v) {
        int um = umax[v < 0 ? -v : v];
        for (int u = -um; u <= um; ++u) {
            int val = img[(size_t)(y + v) * pitch + x + u];
            m10 += u * val;
            m01 += v * val;
        }
    }
    *m10_out = m10;
    *m01_out = m01;
    double cs = 1.0, sn = 0.0;
    long long q = (long long)m10 * m10 + (long long)m01 * m01;
    if (q > 0) {
        double r = sqrt((double)q);
        cs = (double)m10 / r;
        sn = (double)m01 / r;
    }
    memset(desc, 0, 32);
    for (int bit = 0; bit < 256; ++bit) {
        int val[2];
        for (int e = 0; e < 2; ++e) {
            double px = pattern[bit * 4 + 2 * e], py = pattern[bit * 4 + 2 * e + 1];
            double t1 = px * cs, t2 = py * sn, t3 = px * sn, t4 = py * cs;
            int ix = (int)rint(t1 - t2), iy = (int)rint(t3 + t4);
            val[e] = blurred(img, pitch, x + ix, y + iy);
        }
        if (val[0] < val[1]) desc[bit >> 3] |= (uint8_t)(1u << (bit & 7));
    }
}

/* Full chain on one frame.  Outputs have capacity nfeatures.  Returns the number of keypoints. */
int orc_detect_compute(const uint8_t *img, int H, int W, int stride, int nfeatures, int nlevels, float scale_factor,
                       int fast_t, const int8_t *pattern, float *kp_xy, int32_t *kp_meta, float *kp_resp,
                       int32_t *kp_mom, uint8_t *desc) {
    int w[16], h[16], n[16];
    float scale[16];
    orc_level_sizes(H, W, nfeatures, nlevels, scale_factor, w, h, n, scale);
    const uint8_t *cur = img;
    int cur_pitch = stride;
    uint8_t *owned = NULL;
    int out = 0;
    for (int l = 0; l < nlevels; ++l) {
        if (l > 0) {
            uint8_t *nxt = malloc((size_t)w[l] * h[l]);
            orc_resize(cur, w[l - 1], h[l - 1], cur_pitch, nxt, w[l], h[l], w[l]);
            free(owned);
            owned = nxt;
            cur = nxt;
            cur_pitch = w[l];
        }
        const int wl = w[l], hl = h[l];
        if (wl <= 2 * EDGE || hl <= 2 * EDGE || n[l] == 0) continue;
        uint8_t *score = calloc((size_t)wl * hl, 1);
        for (int y = 3; y < hl - 3; ++y)
            for (int x = 3; x < wl - 3; ++x) score[(size_t)y * wl + x] = (uint8_t)fast_score_at(cur, cur_pitch, x, y, fast_t);
        size_t cap = 1024, cnt = 0;
        Cand *c = malloc(cap * sizeof(Cand));
        for (int y = EDGE; y < hl - EDGE; ++y)
            for (int x = EDGE; x < wl - EDGE; ++x) {
                int s = score[(size_t)y * wl + x];
                if (!s) continue;
                int ok = 1;
                for (int dy = -1; dy <= 1 && ok; ++dy)
                    for (int dx = -1; dx <= 1; ++dx)
                        if ((dx || dy) && score[(size_t)(y + dy) * wl + x + dx] >= s) {
                            ok = 0;
                            break;
                        }
                if (!ok) continue;
                if (cnt == cap) {
                    cap *= 2;
                    c = realloc(c, cap * sizeof(Cand));
                }
                c[cnt].x = x;
                c[cnt].y = y;
                c[cnt].score = s;
                c[cnt].harris = 0;
                ++cnt;
            }
        qsort(c, cnt, sizeof(Cand), cmp_score);
        if (cnt > (size_t)2 * n[l]) cnt = (size_t)2 * n[l];
        for (size_t i = 0; i < cnt; ++i) c[i].harris = harris25(cur, cur_pitch, c[i].x, c[i].y);
        qsort(c, cnt, sizeof(Cand), cmp_harris);
        if (cnt > (size_t)n[l]) cnt = (size_t)n[l];
        for (size_t i = 0; i < cnt; ++i) {
            kp_xy[2 * out] = (float)c[i].x * scale[l];
            kp_xy[2 * out + 1] = (float)c[i].y * scale[l];
            kp_meta[4 * out] = l;
            kp_meta[4 * out + 1] = c[i].x;
            kp_meta[4 * out + 2] = c[i].y;
            kp_meta[4 * out + 3] = (int32_t)(c[i].harris & 0xFFFFFFFFll);
            kp_resp[out] = (float)c[i].harris * (float)(1.0 / (25.0 * 7140.0 * 7140.0 * 7140.0 * 7140.0));
            orc_describe(cur, cur_pitch, c[i].x, c[i].y, pattern, desc + (size_t)32 * out, &kp_mom[2 * out],
                         &kp_mom[2 * out + 1]);
            ++out;
        }
        free(c);
        free(score);
    }
    free(owned);
    return out;
}

/* ---- brute-force Hamming 2-NN (exact form of processor.py:132-133), ties -> lowest train index ---- */
void orc_bf_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist) {
    for (int i = 0; i < nq; ++i) {
        int b0 = -1, b1 = -1, d0 = 1 << 30, d1 = 1 << 30;
        uint64_t a[4];
        memcpy(a, q + (size_t)i * 32, 32);
        for (int j = 0; j < nt; ++j) {
            uint64_t b[4];
            memcpy(b, t + (size_t)j * 32, 32);
            int d = __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
                    __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
            if (d < d0) {
                d1 = d0;
                b1 = b0;
                d0 = d;
                b0 = j;
            } else if (d < d1) {
                d1 = d;
                b1 = j;
            }
        }
        idx[2 * i] = b0;
        idx[2 * i + 1] = b1;
        dist[2 * i] = b0 < 0 ? -1 : d0;
        dist[2 * i + 1] = b1 < 0 ? -1 : d1;
    }
}

/* Lowe ratio test (processor.py:136-137): two neighbours and d0 < threshold * d1, query order kept. */
int orc_ratio_filter(const int32_t *idx, const int32_t *dist, int nq, double threshold, int32_t *pairs) {
    int m = 0;
    for (int i = 0; i < nq; ++i) {
        if (idx[2 * i] < 0 || idx[2 * i + 1] < 0) continue;
        if ((double)dist[2 * i] < threshold * (double)dist[2 * i + 1]) {
            pairs[2 * m] = i;
            pairs[2 * m + 1] = idx[2 * i];
            ++m;
        }
    }
    return m;
}

/* ---- single-stage entry points for tests/test_oracle_definitions.py (the stages above, one at a time) ---- */
void orc_fast_score_map(const uint8_t *img, int H, int W, int pitch, int t, uint8_t *score /*[H*W], 0 on the 3-px rim*/) {
    memset(score, 0, (size_t)H * W);
    for (int y = 3; y < H - 3; ++y)
        for (int x = 3; x < W - 3; ++x) score[(size_t)y * W + x] = (uint8_t)fast_score_at(img, pitch, x, y, t);
}
long long orc_harris25_at(const uint8_t *img, int pitch, int x, int y) { return harris25(img, pitch, x, y); }
int orc_blurred_at(const uint8_t *img, int pitch, int x, int y) { return blurred(img, pitch, x, y); }
