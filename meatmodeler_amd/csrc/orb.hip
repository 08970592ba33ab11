// ORB detect + describe for batches of grey frames (gfx950).  Integer-exact definition: DESIGN.md §"mm-ORB".
//
// Replaces orb.detectAndCompute(img, None) (reference processor.py:129,328; cv2.ORB_create(nfeatures) defaults,
// processor.py:308): 8-level x1.2 pyramid, FAST-9/16 (t=20) + strict 3x3 NMS, 31-px border, per level keep 2n by FAST
// score then n by Harris (7x7, k=0.04), intensity-centroid orientation (radius 15), 7x7 sigma=2 blur, 256 steered
// intensity comparisons.  OpenCV itself is absent offline, so its two implementation-defined points are fixed here:
// selection order is (response desc, y asc, x asc) instead of std::nth_element's, and the rBRIEF pair table is a
// seeded table supplied by the caller (OpenCV's learned bit_pattern_31_ cannot be regenerated).
//
// Kernels (all batched over frames; image-sized passes are HBM/L2-bound streaming stencils):
//   orb_tables      resize coordinate/weight tables (11-bit fixed point)
//   orb_resize      level l-1 -> l, bilinear, 4 pixels / thread
//   orb_fast        LDS tile (72x24 px) -> FAST score tile (66x18) -> NMS -> candidate keys + score histogram
//   orb_select      exact radix select of the 2n smallest 32-bit keys (255-score | y | x) per (frame, level)
//   orb_harris      compaction of the selected candidates + exact integer Harris measure 25(ab-c^2)-(a+b)^2
//   orb_rank        rank by (Harris desc, y, x) (counting rank, O(m^2), m <= 2n), keep n, scatter in order
//   orb_describe    one wave per keypoint: 39x39 patch -> LDS, moments, separable integer blur, 256 tests
#include "mm_common.h"
#include <algorithm>
#include <vector>

#pragma clang fp contract(off)

namespace {

constexpr int MAXL = 8;
constexpr int EDGE = 31;

// products of operands below 2^23 whose result fits 32 bits: the full-rate 24-bit multiplier (v_mul_lo_u32 and
// v_mad_u64_u32, which the compiler picks for `int * int`, issue at a quarter of the rate)
// (as instructions: the intrinsics are widened back to 32-bit multiplies wherever the compiler cannot prove the operand range)
// 16 bytes from a 4-byte aligned address (image rows start 64-byte aligned, tiles and patches at multiples of 4 pixels): global
// memory takes a dwordx4 at dword alignment; the type tells the compiler so
struct __attribute__((packed, aligned(4))) U4A4 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ uint4 load16_a4(const uint8_t *p) {
    const U4A4 v = *reinterpret_cast<const U4A4 *>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int mul24(int a, int b) {
    int d;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ int mad24(int a, int b, int c) {
    int d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// A run of pyramid levels produced by ONE launch: a workgroup owns a tile of the run's source level and everything of
// the following levels whose left / upper bilinear tap falls into what it owns one level up (the taps are monotone, so
// the owned ranges are intervals that partition every level); the few columns / rows beyond, needed as right / lower taps
// further down, are recomputed (bit-identical) rather than exchanged.  Bounds per tile boundary, as int32 in the workspace:
//   X: lo[j][0..ntx] (j = 0..n: first owned column at run level j), nh[j][0..ntx-1] (last needed column); then Y likewise.
#ifndef MM_PY_THREADS
#define MM_PY_THREADS 512
#endif
constexpr int PY_MAXN = 4, PY_TW = 192, PY_TH = 64, PY_THREADS = MM_PY_THREADS;
constexpr int PY_PITCH_A = 216, PY_PITCH_B = 184;      // LDS pitches of the two rectangles (compile-time: row steps are immediates)
constexpr int PY_BUF_A = 15984, PY_BUF_B = 11960, PY_TABX = 224, PY_TABY = 96;      // 74 rows x 216, 65 rows x 184: 34 KB with the tables, 4 workgroups per CU
struct PyrChain {
    int src, n;              // source level, levels produced (src + 1 .. src + n)
    int ntx, nty;
    int fused;               // 0: the rectangles do not fit the kernel's LDS buffers -> one launch per level
    size_t bnd_off;
};
__host__ __device__ inline int py_x_lo(int j, int t, int ntx) { return j * (ntx + 1) + t; }
__host__ __device__ inline int py_x_nh(int j, int t, int n, int ntx) { return (n + 1) * (ntx + 1) + j * ntx + t; }
__host__ __device__ inline int py_dim_words(int n, int nt) { return (n + 1) * (2 * nt + 1); }

// source coordinate of destination coordinate d (nd destination, ns source samples): the integers of orb_tables_kernel
__host__ __device__ inline int rz_src0(long long d, long long nd, long long ns) {
    long long num = (2 * d + 1) * ns - nd;
    if (num < 0) num = 0;
    long long i0 = num / (2 * nd);
    if (i0 >= ns - 1) i0 = ns - 1;
    return (int)i0;
}
// smallest d in [0, nd] with rz_src0(d) >= a (nd: none)
__host__ __device__ inline int rz_lower(long long a, long long nd, long long ns) {
    if (a <= 0) return 0;
    if (a > ns - 1) return (int)nd;
    const long long num = (2 * a + 1) * nd - ns;
    long long d = num <= 0 ? 0 : (num + 2 * ns - 1) / (2 * ns);
    if (d > nd) d = nd;
    while (d > 0 && rz_src0(d - 1, nd, ns) >= a) --d;
    while (d < nd && rz_src0(d, nd, ns) < a) ++d;
    return (int)d;
}
// bounds of one tile boundary / tile `t` along one dimension; dims[j] = samples of run level j; out = the dimension's block
__host__ __device__ inline void py_bounds_lo(int t, int nt, int tile, int n, const int *dims, int32_t *out) {
    int lo = t * tile < dims[0] ? t * tile : dims[0];
    out[py_x_lo(0, t, nt)] = lo;
    for (int j = 1; j <= n; ++j) {
        lo = t == nt ? dims[j] : rz_lower(lo, dims[j], dims[j - 1]);
        out[py_x_lo(j, t, nt)] = lo;
    }
}
__host__ __device__ inline void py_bounds_nh(int t, int nt, int n, const int *dims, int32_t *out) {      // (after all lo)
    int nh = out[py_x_lo(n, t + 1, nt)] - 1;
    out[py_x_nh(n, t, n, nt)] = nh;
    for (int j = n; j >= 1; --j) {
        const int own = out[py_x_lo(j - 1, t + 1, nt)] - 1;
        int tap = -1;
        if (nh >= out[py_x_lo(j, t, nt)]) {
            tap = rz_src0(nh, dims[j], dims[j - 1]) + 1;
            if (tap > dims[j - 1] - 1) tap = dims[j - 1] - 1;
        }
        nh = own > tap ? own : tap;
        out[py_x_nh(j - 1, t, n, nt)] = nh;
    }
}

struct OrbGeom {
    int nlevels, batch, H, W, stride0;
    int w[MAXL], h[MAXL], pitch[MAXL];
    int nfeat[MAXL];
    float scale[MAXL];
    size_t pyr_off[MAXL];    // byte offset of level l (l>=1) for frame 0 inside the workspace
    size_t pyr_bytes[MAXL];  // bytes per frame of level l
    int tile_start[MAXL + 1];  // FAST tiles: cumulative count per level
    int tiles_x[MAXL];
    int cand_cap[MAXL];
    size_t cand_off[MAXL];  // byte offset of the candidate key array of level l, frame 0
    size_t tab_off[MAXL];   // resize tables of level l: x0[w], wx[w], y0[h], wy[h] (int32)
    PyrChain chain[2];      // the pyramid in (at most) two launches: levels 1..3 from the frame, 4..7 from level 3
    int nchains;
    int kcap;               // kept capacity per segment (>= 2 * nfeat[0])
    size_t cnt_off, hist_off, thr_off, kcnt_off, kkey_off, kh_off;
    int cap_out;  // nfeatures
};

__device__ __forceinline__ const uint8_t *level_ptr(const OrbGeom &g, const uint8_t *imgs, const uint8_t *ws, int l,
                                                    int b, int &pitch) {
    if (l == 0) {
        pitch = g.stride0;
        return imgs + (size_t)b * g.H * g.stride0;
    }
    pitch = g.pitch[l];
    return ws + g.pyr_off[l] + (size_t)b * g.pyr_bytes[l];
}

// ---- resize tables ------------------------------------------------------------------------------------------------------
__global__ void orb_tables_kernel(OrbGeom g, uint8_t *ws) {
    const int l = blockIdx.x + 1;
    if (l >= g.nlevels) {      // the blocks behind the levels: tile bounds of the fused pyramid launches
        const int c = l - g.nlevels;
        if (c >= g.nchains || !g.chain[c].fused) return;
        const PyrChain ch = g.chain[c];
        int32_t *bx = reinterpret_cast<int32_t *>(ws + ch.bnd_off), *by = bx + py_dim_words(ch.n, ch.ntx);
        int dw[PY_MAXN + 1], dh[PY_MAXN + 1];
        for (int j = 0; j <= ch.n; ++j) {
            dw[j] = g.w[ch.src + j];
            dh[j] = g.h[ch.src + j];
        }
        for (int t = threadIdx.x; t <= ch.ntx + ch.nty + 1; t += blockDim.x) {
            if (t <= ch.ntx) py_bounds_lo(t, ch.ntx, PY_TW, ch.n, dw, bx);
            else py_bounds_lo(t - ch.ntx - 1, ch.nty, PY_TH, ch.n, dh, by);
        }
        __syncthreads();
        for (int t = threadIdx.x; t < ch.ntx + ch.nty; t += blockDim.x) {
            if (t < ch.ntx) py_bounds_nh(t, ch.ntx, ch.n, dw, bx);
            else py_bounds_nh(t - ch.ntx, ch.nty, ch.n, dh, by);
        }
        return;
    }
    int32_t *tab = reinterpret_cast<int32_t *>(ws + g.tab_off[l]);
    const int wd = g.w[l], hd = g.h[l], wsrc = g.w[l - 1], hsrc = g.h[l - 1];
    for (int i = threadIdx.x; i < wd + hd; i += blockDim.x) {
        const bool isx = i < wd;
        const long long d = isx ? i : i - wd;
        const long long nd = isx ? wd : hd, ns = isx ? wsrc : hsrc;
        long long num = (2 * d + 1) * ns - nd;  // (d + 0.5) * ns/nd - 0.5, times 2 nd
        if (num < 0) num = 0;
        long long i0 = num / (2 * nd);
        long long rem = num - i0 * 2 * nd;
        long long wgt = (rem * 2048 + nd) / (2 * nd);
        if (i0 >= ns - 1) {
            i0 = ns - 1;
            wgt = 0;
        }
        if (isx) {
            tab[d] = (int32_t)i0;
            tab[wd + d] = (int32_t)wgt;
        } else {
            tab[2 * wd + d] = (int32_t)i0;
            tab[2 * wd + hd + d] = (int32_t)wgt;
        }
    }
}

// ---- pyramid level l-1 -> l ---------------------------------------------------------------------------------------------
// Output tile 256 x 16 per workgroup.  The source footprint (<= 24 rows x 320 bytes at any scale >= 1) is staged in LDS
// with coalesced 4-byte loads; a thread keeps the x positions / weights of its four output pixels in registers for all
// its rows (the x tables are read once per thread, not once per row), takes its bilinear taps from LDS and writes one
// packed word per row.  (Tapping global memory directly cost 24 dependent byte loads per four output pixels:
// 0.63 TB/s of algorithmic traffic; the arithmetic is unchanged, bit for bit.)
constexpr int RZ_TW = 256, RZ_TH = 16, RZ_SR = 24, RZ_SW = 320;

__global__ __launch_bounds__(256) void orb_resize_kernel(OrbGeom g, const uint8_t *__restrict__ imgs,
                                                         uint8_t *__restrict__ ws, int l, int tw, int th) {
    // tw x th: output tile (tw <= RZ_TW multiple of 4, th <= RZ_TH), chosen on the host so that the source footprint fits T
    __shared__ __attribute__((aligned(16))) uint8_t T[RZ_SR][RZ_SW];
    const int b = blockIdx.z;
    const int wd = g.w[l], hd = g.h[l], pd = g.pitch[l];
    const int tx0 = blockIdx.x * tw, ty0 = blockIdx.y * th;
    int ps;
    const uint8_t *src = level_ptr(g, imgs, ws, l - 1, b, ps);
    uint8_t *dst = ws + g.pyr_off[l] + (size_t)b * g.pyr_bytes[l];
    const int32_t *tab = reinterpret_cast<const int32_t *>(ws + g.tab_off[l]);
    const int ws_ = g.w[l - 1], hs_ = g.h[l - 1];
    const int txl = min(tx0 + tw, wd) - 1, tyl = min(ty0 + th, hd) - 1;   // last output column / row of the tile
    if (txl < tx0) {  // tile entirely inside the row padding: zero fill
        const int rl0 = threadIdx.x >> 6, cg0 = threadIdx.x & 63;
        for (int j = 0; j < RZ_TH / 4; ++j) {
            const int yy = rl0 + 4 * j;
            if (yy < th && ty0 + yy < hd && 4 * cg0 < tw && tx0 + 4 * cg0 < pd)
                *reinterpret_cast<uint32_t *>(dst + (size_t)(ty0 + yy) * pd + tx0 + 4 * cg0) = 0;
        }
        return;
    }
    const int sx_lo = tab[tx0] & ~3, sx_hi = min(tab[txl] + 1, ws_ - 1);
    const int sy_lo = tab[2 * wd + ty0], sy_hi = min(tab[2 * wd + tyl] + 1, hs_ - 1);
    const int nrow = sy_hi - sy_lo + 1, nw4 = (sx_hi - sx_lo) / 4 + 1;   // <= RZ_SR, <= RZ_SW / 4
    for (int e = threadIdx.x; e < nrow * nw4; e += 256) {
        const int r = e / nw4, c4 = e % nw4;
        const int x = sx_lo + 4 * c4;
        uint32_t v = 0;
        if (x + 3 < ps) v = *reinterpret_cast<const uint32_t *>(src + (size_t)(sy_lo + r) * ps + x);
        else
            for (int k = 0; k < 4; ++k)
                if (x + k < ps) v |= (uint32_t)src[(size_t)(sy_lo + r) * ps + x + k] << (8 * k);
        *reinterpret_cast<uint32_t *>(&T[r][4 * c4]) = v;
    }
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    int c0[4], c1[4], wx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = min(tx0 + 4 * cg + k, wd - 1);
        const int x0 = tab[x];
        wx[k] = tab[wd + x];
        c0[k] = x0 - sx_lo;
        c1[k] = min(x0 + 1, ws_ - 1) - sx_lo;
    }
    __syncthreads();
    if (4 * cg >= tw || tx0 + 4 * cg >= pd) return;
#pragma unroll
    for (int j = 0; j < RZ_TH / 4; ++j) {
        const int y = ty0 + rl + 4 * j;
        if (rl + 4 * j >= th || y >= hd) break;
        const int y0 = tab[2 * wd + y], wy = tab[2 * wd + hd + y];
        const uint8_t *r0 = T[y0 - sy_lo], *r1 = T[min(y0 + 1, hs_ - 1) - sy_lo];
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t v = 0;
            if (tx0 + 4 * cg + k < wd) {
                const int top = mad24(r0[c0[k]], 2048 - wx[k], mul24(r0[c1[k]], wx[k]));
                const int bot = mad24(r1[c0[k]], 2048 - wx[k], mul24(r1[c1[k]], wx[k]));
                v = (uint32_t)mad24(top, 2048 - wy, mad24(bot, wy, 1 << 21)) >> 22;
            }
            packed |= v << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(dst + (size_t)y * pd + tx0 + 4 * cg) = packed;
    }
}

// one entry of the resize tables, recomputed in 32-bit (dimensions < 4096: (2 d + 1) ns < 2^25): no table loads between levels
__device__ __forceinline__ void rz_coord(int d, int nd, int ns, int &i0, int &wgt) {
    const int num_ = (2 * d + 1) * ns - nd;
    const unsigned num = num_ < 0 ? 0u : (unsigned)num_, den = 2u * (unsigned)nd;
    // quotients of numbers below 2^25 by a divisor below 2^13: float estimate (off by at most one), then the exact fix-up
    const float rcp = 1.0f / (float)den;
    auto div = [&](unsigned n_) {
        unsigned q_ = (unsigned)((float)n_ * rcp);
        int r_ = (int)(n_ - q_ * den);
        if (r_ < 0) --q_;
        else if (r_ >= (int)den) ++q_;
        return q_;
    };
    unsigned q = div(num);
    const unsigned rem = num - q * den;
    unsigned w_ = div(rem * 2048u + (unsigned)nd);
    if (q >= (unsigned)(ns - 1)) {
        q = (unsigned)(ns - 1);
        w_ = 0;
    }
    i0 = (int)q;
    wgt = (int)w_;
}

// ---- a run of pyramid levels in one launch -------------------------------------------------------------------------------
// (see PyrChain.)  Two LDS rectangles hold alternate levels and the levels in between are never read back from memory
// (10.6 -> 7.1 MB of traffic per 1080p frame, 7 launches -> 2).  The arithmetic is orb_resize_kernel's, bit for bit.
// Measured (profiles/r03_orb_kernels.txt): 2.55 ms per 500 x 1080p against 3.2 ms for the seven per-level launches; with its phases
// switched off one at a time the kernel (256-wide tiles then) split into 0.6 ms of per-workgroup latency (launch, bounds,
// barriers: 93 k workgroups, three resident per CU), 0.6 ms source load, 1.4 ms level arithmetic, 0.25 ms stores -- none of them at a hardware roof.
// One level of a run: `prev` (pitch PP) -> `cur` (pitch CP) and memory.  lane = four adjacent columns, wave = a run of output
// rows; the wave walks the SOURCE rows of its run once, interpolating each horizontally as it comes (the bytes of the next row
// are requested before the arithmetic of the current one) and emitting an output row whenever its lower tap has just been
// interpolated -- its upper tap is then the row before (taps of consecutive output rows are strictly increasing, checked on
// the host).  All row decisions are wave-uniform.
struct PyLevel {
    int wd, pd, lox, ohx, ohy, loy, nhy, ax, nq, rows;
    uint8_t *dst;
};
template <int PP, int CP>
__device__ __forceinline__ void py_level(const PyLevel L, const uint8_t *prev, uint8_t *cur, const int4 *s_col, const int4 *s_row, int tid) {
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: the row loop is uniform control flow)
    constexpr int NW = PY_THREADS / 64;
    const int per = (L.rows + NW - 1) / NW;
    const int ya = wv * per, yb = min(ya + per, L.rows);
    if (ya >= yb) return;
    for (int q = lane; q < L.nq; q += 64) {
        int wx[4], ux[4];
        const uint8_t *p0[4], *p1[4];
        const int4 first = s_row[ya], last = s_row[yb - 1];
        int t = __builtin_amdgcn_readfirstlane(first.x);
        const int t_last = __builtin_amdgcn_readfirstlane(last.y);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int4 e = s_col[4 * q + k];
            p0[k] = prev + t * PP + e.x;
            p1[k] = prev + t * PP + e.y;
            wx[k] = e.z & 0xFFF;      // (weights are 0 .. 2047: known bits let the compiler pick the 24-bit multiplier)
            ux[k] = 2048 - wx[k];
        }
        const int x4 = L.ax + 4 * q;
        const bool full = x4 >= L.lox && x4 + 3 <= L.ohx;
        const int live = min(max(L.wd - x4, 0), 4);      // columns of the quad inside the image (the rest is zero padding)
        const uint32_t keep = live >= 4 ? 0xFFFFFFFFu : ((1u << (8 * live)) - 1u);
        uint8_t *curq = cur + 4 * q;
        int i = ya;
        int r0 = t, r1 = __builtin_amdgcn_readfirstlane(first.y), wy = __builtin_amdgcn_readfirstlane(first.z) & 0xFFF;
        int a[8], bb[8], h0[4] = {0, 0, 0, 0}, h1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = p0[k][0];
            a[4 + k] = p1[k][0];
        }
        // one source row: request the next one into `nx` at row offset OFF, interpolate `cu` into `hc`, emit if due
        auto step = [&](const int (&cu)[8], int (&nx)[8], const int (&hp)[4], int (&hc)[4], const int OFF) {
            if (t < t_last) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    nx[k] = p0[k][OFF];
                    nx[4 + k] = p1[k][OFF];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) hc[k] = mad24(cu[k], ux[k], mul24(cu[4 + k], wx[k]));
            if (t == r1) {
                const int uy = 2048 - wy;
                uint32_t packed = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int top = r0 == r1 ? hc[k] : hp[k];
                    const uint32_t v = (uint32_t)mad24(top, uy, mad24(hc[k], wy, 1 << 21)) >> 22;
                    packed |= v << (8 * k);
                }
                packed &= keep;
                *reinterpret_cast<uint32_t *>(curq + i * CP) = packed;
                const int y = L.loy + i;
                if (y <= L.ohy && full) *reinterpret_cast<uint32_t *>(L.dst + (size_t)y * L.pd + x4) = packed;
                ++i;
                if (i < yb) {
                    const int4 e = s_row[i];
                    r0 = __builtin_amdgcn_readfirstlane(e.x);
                    r1 = __builtin_amdgcn_readfirstlane(e.y);
                    wy = __builtin_amdgcn_readfirstlane(e.z) & 0xFFF;
                }
            }
            ++t;
        };
        while (t <= t_last) {
            step(a, bb, h0, h1, PP);
            if (t > t_last) break;
            step(bb, a, h1, h0, 2 * PP);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                p0[k] += 2 * PP;
                p1[k] += 2 * PP;
            }
        }
    }
}

__global__ __launch_bounds__(PY_THREADS) void orb_pyramid_kernel(OrbGeom g, const uint8_t *__restrict__ imgs, uint8_t *__restrict__ ws,
                                                                int c) {
    __shared__ __attribute__((aligned(16))) uint8_t bufA[PY_BUF_A];
    __shared__ __attribute__((aligned(16))) uint8_t bufB[PY_BUF_B];
    __shared__ int4 s_col[PY_TABX], s_row[PY_TABY];      // {tap 0, tap 1, weight, -} relative to the previous rectangle
    const PyrChain ch = g.chain[c];
    const int tx = blockIdx.x, ty = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int32_t *BX = reinterpret_cast<const int32_t *>(ws + ch.bnd_off), *BY = BX + py_dim_words(ch.n, ch.ntx);
    // all bounds of this tile up front (uniform: scalar loads, one latency instead of one per level)
    int LX[PY_MAXN + 1], NX[PY_MAXN + 1], OX[PY_MAXN + 1], LY[PY_MAXN + 1], NY[PY_MAXN + 1], OY[PY_MAXN + 1];
#pragma unroll
    for (int j = 0; j <= PY_MAXN; ++j) {
        const int jj = j <= ch.n ? j : ch.n;
        LX[j] = BX[py_x_lo(jj, tx, ch.ntx)];
        OX[j] = BX[py_x_lo(jj, tx + 1, ch.ntx)] - 1;
        NX[j] = BX[py_x_nh(jj, tx, ch.n, ch.ntx)];
        LY[j] = BY[py_x_lo(jj, ty, ch.nty)];
        OY[j] = BY[py_x_lo(jj, ty + 1, ch.nty)] - 1;
        NY[j] = BY[py_x_nh(jj, ty, ch.n, ch.nty)];
    }
    // the source rectangle
    int pax = LX[0] & ~3, ploy = LY[0];
    {
        int ps;
        const uint8_t *src = level_ptr(g, imgs, ws, ch.src, b, ps);
        const int nw4 = (NX[0] - pax) / 4 + 1, nrow = NY[0] - ploy + 1;
        const unsigned inv20 = ((1u << 20) + nw4 - 1) / nw4;      // e / nw4 for e < 2^20 / nw4 (e < 74 * 70)
        // (tried in round 4: 16-byte loads here, as in the FAST tile and the descriptor patch -- 2.62 vs 2.61 ms, nothing)
        for (int e = tid; e < nrow * nw4; e += PY_THREADS) {
            const int r = (int)(((unsigned)e * inv20) >> 20), c4 = e - r * nw4;
            const int x = pax + 4 * c4;
            const uint8_t *rowp = src + (size_t)(ploy + r) * ps;
            uint32_t v = 0;
            if (x + 3 < ps) v = *reinterpret_cast<const uint32_t *>(rowp + x);
            else
                for (int k = 0; k < 4; ++k)
                    if (x + k < ps) v |= (uint32_t)rowp[x + k] << (8 * k);
            *reinterpret_cast<uint32_t *>(bufA + r * PY_PITCH_A + 4 * c4) = v;
        }
    }
#pragma unroll
    for (int j = 1; j <= PY_MAXN; ++j) {
        if (j > ch.n) break;
        const int l = ch.src + j;
        const uint8_t *prev = (j & 1) ? bufA : bufB;
        uint8_t *cur = (j & 1) ? bufB : bufA;
        const int cp = (j & 1) ? PY_PITCH_B : PY_PITCH_A;
        const int wd = g.w[l], hd = g.h[l], pd = g.pitch[l], ws_ = g.w[l - 1], hs_ = g.h[l - 1];
        PyLevel L;
        L.wd = wd;
        L.pd = pd;
        L.lox = LX[j];
        L.loy = LY[j];
        L.nhy = NY[j];
        L.ohy = OY[j];
        // what this workgroup writes to memory: its own columns (the last tile also the zero padding of the word wd - 1 is in)
        L.ohx = tx == ch.ntx - 1 ? min(pd - 1, (wd - 1) | 3) : OX[j];
        const int chx = max(NX[j], L.ohx);      // last column computed
        L.ax = L.lox & ~3;
        L.nq = chx >= L.lox ? (chx - L.ax) / 4 + 1 : 0;
        L.rows = L.nhy - L.loy + 1;
        L.dst = ws + g.pyr_off[l] + (size_t)b * g.pyr_bytes[l];
        // taps (row taps as row INDICES of the previous rectangle) and weights
        for (int i = tid; i < 4 * L.nq + L.rows; i += PY_THREADS) {
            if (i < 4 * L.nq) {
                int x0, wx_;
                rz_coord(min(max(L.ax + i, L.lox), wd - 1), wd, ws_, x0, wx_);
                s_col[i] = make_int4(x0 - pax, min(x0 + 1, ws_ - 1) - pax, wx_, 0);
            } else {
                int y0, wy_;
                rz_coord(L.loy + i - 4 * L.nq, hd, hs_, y0, wy_);
                s_row[i - 4 * L.nq] = make_int4(y0 - ploy, min(y0 + 1, hs_ - 1) - ploy, wy_, 0);
            }
        }
        __syncthreads();      // (also: the previous level is complete in `prev`)
        if (L.nq > 0 && L.rows > 0) {
            if (j & 1) py_level<PY_PITCH_A, PY_PITCH_B>(L, prev, cur, s_col, s_row, tid);
            else py_level<PY_PITCH_B, PY_PITCH_A>(L, prev, cur, s_col, s_row, tid);
            // the (at most two) words of a row that straddle the boundary to a neighbouring tile: byte by byte, from LDS
            __syncthreads();
            const int nown = min(L.ohy, L.nhy) - L.loy + 1;
            for (int e = tid; e < 2 * nown; e += PY_THREADS) {
                const int i = e >> 1, x4 = (e & 1) ? (L.ohx & ~3) : L.ax;
                if ((e & 1) && x4 == L.ax) continue;
                if (x4 >= L.lox && x4 + 3 <= L.ohx) continue;      // a full word: written above
                const uint32_t v = *reinterpret_cast<const uint32_t *>(cur + i * cp + (x4 - L.ax));
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x4 + k >= L.lox && x4 + k <= L.ohx) L.dst[(size_t)(L.loy + i) * pd + x4 + k] = (uint8_t)(v >> (8 * k));
            }
        }
        // the rest of the row padding (last tile of a row): zeros, as the per-level kernel leaves it
        if (tx == ch.ntx - 1) {
            const int z0 = ((wd - 1) | 3) + 1, nz4 = (pd - z0) / 4, nown = L.ohy - L.loy + 1;
            for (int e = tid; e < nz4 * nown; e += PY_THREADS) {
                const int r = e / nz4, c4 = e - r * nz4;
                *reinterpret_cast<uint32_t *>(L.dst + (size_t)(L.loy + r) * pd + z0 + 4 * c4) = 0;
            }
        }
        pax = L.ax;
        ploy = L.loy;
        __syncthreads();      // (the tables are rewritten, `prev` becomes the next target)
    }
}

// ---- FAST-9/16 + NMS ----------------------------------------------------------------------------------------------------
#ifndef MM_FT_H
#define MM_FT_H 64
#endif
constexpr int FT_W = 64, FT_H = MM_FT_H;        // keypoint tile (the width is the wave: lane = column)
constexpr int FP_W = 80, FP_H = FT_H + 8;  // pixel tile in LDS (80 = 72 needed + alignment slack)
constexpr int FS_W = FT_W + 2, FS_H = FT_H + 2;

// the 16 ring pixels of FAST around P[r][c], clockwise from (row + 3, col)
__device__ __forceinline__ void fast_ring(const uint8_t (*P)[FP_W], int r, int c, int (&v)[16]) {
    v[0] = P[r + 3][c];      v[1] = P[r + 3][c + 1];  v[2] = P[r + 2][c + 2];  v[3] = P[r + 1][c + 3];
    v[4] = P[r][c + 3];      v[5] = P[r - 1][c + 3];  v[6] = P[r - 2][c + 2];  v[7] = P[r - 3][c + 1];
    v[8] = P[r - 3][c];      v[9] = P[r - 3][c - 1];  v[10] = P[r - 2][c - 2]; v[11] = P[r - 1][c - 3];
    v[12] = P[r][c - 3];     v[13] = P[r + 1][c - 3]; v[14] = P[r + 2][c - 2]; v[15] = P[r + 3][c - 1];
}

// 0: no corner, 1: nine contiguous ring pixels brighter than p + t, 2: darker than p - t (never both: 9 + 9 > 16).
// (Per-lane bit masks: the same logic on wave-wide ballot masks in scalar registers was 1.7x slower here -- 158
// dependent scalar operations per step -- while it pays for the four-pixel pre-test of stage 1.)
__device__ __forceinline__ int fast_corner_kind(const uint8_t (*P)[FP_W], int r, int c, int t) {
    const int p = P[r][c];
    int v[16];
    fast_ring(P, r, c, v);
    uint32_t mb = 0, md = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        mb |= (uint32_t)(v[i] > p + t) << i;
        md |= (uint32_t)(v[i] < p - t) << i;
    }
    auto has9 = [](uint32_t m) {
        m |= m << 16;
        uint32_t x = m & (m >> 1);
        x &= x >> 2;
        x &= x >> 4;
        x &= m >> 8;
        return (x & 0xFFFFu) != 0;
    };
    return has9(mb) ? 1 : (has9(md) ? 2 : 0);
}

// largest threshold for which the pixel is still a corner of the given kind: max over the 16 arcs of nine of the
// smallest signed difference in the arc, minus one
__device__ __forceinline__ int fast_score(const uint8_t (*P)[FP_W], int r, int c, int kind) {
    const int p = P[r][c];
    int v[16], d[16], m1[16], m2[16], m4[16];
    fast_ring(P, r, c, v);
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = kind == 1 ? v[i] - p : p - v[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) m1[i] = min(d[i], d[(i + 1) & 15]);
#pragma unroll
    for (int i = 0; i < 16; ++i) m2[i] = min(m1[i], m1[(i + 2) & 15]);
#pragma unroll
    for (int i = 0; i < 16; ++i) m4[i] = min(m2[i], m2[(i + 4) & 15]);
    int best = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) best = max(best, min(m4[i], d[(i + 8) & 15]));
    return best - 1;
}

// append the lanes with `pass` to an LDS list (one LDS atomic per wave)
__device__ __forceinline__ void wave_append(bool pass, uint16_t value, uint16_t *list, int *count) {
    const unsigned long long m = __ballot(pass);
    if (m == 0) return;
    int base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, 0, 64);
    if (pass) list[base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1))] = value;
}

__global__ __launch_bounds__(256) void orb_fast_kernel(OrbGeom g, const uint8_t *__restrict__ imgs,
                                                       uint8_t *__restrict__ ws, int fast_t) {
    __shared__ __attribute__((aligned(16))) uint8_t P[FP_H][FP_W];
    __shared__ __attribute__((aligned(16))) uint8_t Sc[FS_H][FS_W + 2];
    const int b = blockIdx.y;
    int l = 0;
    while (l + 1 < g.nlevels && (int)blockIdx.x >= g.tile_start[l + 1]) ++l;
    const int tl = blockIdx.x - g.tile_start[l];
    const int tx = tl % g.tiles_x[l], ty = tl / g.tiles_x[l];
    const int ox = EDGE + tx * FT_W, oy = EDGE + ty * FT_H;
    const int w = g.w[l], h = g.h[l];
    int pitch;
    const uint8_t *img = level_ptr(g, imgs, ws, l, b, pitch);
    // pixel tile: rows oy-4 .. oy+67, columns ox-7 .. ox+72 (ox-7 is a multiple of 4)
    const int bx = ox - 7, by = oy - 4;
    // (16 bytes per lane and access: 360 instead of 1440 loads per tile; the address is only 4-byte aligned, which global
    // loads accept; a piece that would cross the pitch falls back to words)
    static_assert(FP_W % 16 == 0, "pixel tile rows are whole 16-byte pieces");
    for (int e = threadIdx.x; e < FP_H * (FP_W / 16); e += 256) {
        const int r = e / (FP_W / 16), c16 = (e % (FP_W / 16)) * 16;
        const int y = by + r, x = bx + c16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (y < h) {
            const uint8_t *src = img + (size_t)y * pitch + x;
            if (x + 15 < pitch) {
                v = load16_a4(src);
            } else {
                if (x + 3 < pitch) v.x = *reinterpret_cast<const uint32_t *>(src);
                if (x + 7 < pitch) v.y = *reinterpret_cast<const uint32_t *>(src + 4);
                if (x + 11 < pitch) v.z = *reinterpret_cast<const uint32_t *>(src + 8);
            }
        }
        *reinterpret_cast<uint4 *>(&P[r][c16]) = v;
    }
    static_assert(sizeof(Sc) % 4 == 0, "score tile is cleared by 16-byte pieces and a tail of words");
    for (int e = threadIdx.x; e < (int)sizeof(Sc) / 16; e += 256) reinterpret_cast<uint4 *>(&Sc[0][0])[e] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < (int)(sizeof(Sc) % 16) / 4) reinterpret_cast<uint32_t *>(&Sc[0][0])[(sizeof(Sc) / 16) * 4 + threadIdx.x] = 0;
    // A cascade over the score positions, each stage compacting its survivors into an LDS list so that the next one runs
    // on full waves (in one pass, a wave pays for the most expensive stage as soon as one of its 64 pixels gets there):
    //   1. every position: the exact compass pre-test (any 9 contiguous ring pixels contain two of the four compass
    //      pixels, so a corner needs two of them brighter than p + t or two darker than p - t)          ~7 % survive
    //   2. survivors: the 16 comparisons and the nine-in-a-row test                                     ~1 % are corners
    //   3. corners: the score (16 arcs of nine), written into the score tile
    //   4. corners again: strict 3 x 3 non-maximum suppression against the score tile -> key points
    constexpr int TODO_SEG = ((FS_H + 3) / 4 + 1) * 64;      // rows of a wave + its share of the two right-most columns
    __shared__ uint16_t todo[4 * TODO_SEG];
    uint16_t *corners = todo;
    __shared__ int seg_n[4], n_corners;
    if (threadIdx.x == 0) n_corners = 0;
    __syncthreads();
    {
        // Stage 1, one row of score positions per wave and step (lane = column; the two columns 64, 65 are left to one
        // extra step): addresses advance by a constant, the survivors of all steps are kept as a bit mask per lane and
        // appended to the list at the end with ONE LDS atomic per wave.
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        // The test itself runs on wave-wide masks (one bit per lane, in scalar registers): per position 8 vector
        // compares; the and / or logic is scalar work, which issues beside the vector instructions of the other waves.
        // Any 9 contiguous ring pixels contain pixel 0 or 8 AND pixel 4 or 12, so a corner needs
        // (b0 | b8) & (b4 | b12) among the brighter-than-p+t bits or the same among the darker-than-p-t bits.
        auto compass = [&](int sr, int sc, bool ok) -> unsigned long long {
            // score position: image (ox-1+sc, oy-1+sr) -> P[r = sr+3][c = sc+6]
            const int r = sr + 3, c = sc + 6;
            const int p = P[r][c], hi = ok ? p + fast_t : 256, lo = ok ? p - fast_t : -1;
            const int a0 = P[r + 3][c], a4 = P[r][c + 3], a8 = P[r - 3][c], a12 = P[r][c - 3];
            const unsigned long long b = (__ballot(a0 > hi) | __ballot(a8 > hi)) & (__ballot(a4 > hi) | __ballot(a12 > hi));
            const unsigned long long d = (__ballot(a0 < lo) | __ballot(a8 < lo)) & (__ballot(a4 < lo) | __ballot(a12 < lo));
            return b | d;
        };
        // every wave appends to its own segment of the list (neighbouring entries = neighbouring pixels: stage 2 reads P
        // without bank conflicts); the running count is a scalar
        const bool col_ok = ox - 1 + lane < w - 3;
        const unsigned long long below = (1ull << lane) - 1;
        uint16_t *seg = todo + wv * TODO_SEG;
        int cnt = 0;
        const int sr_end = min(FS_H, h - 3 - (oy - 1));
        for (int sr = wv; sr < sr_end; sr += 4) {
            const unsigned long long m = compass(sr, lane, col_ok);
            if ((m >> lane) & 1) seg[cnt + __popcll(m & below)] = (uint16_t)(sr * FS_W + lane);
            cnt += __popcll(m);
        }
        const int q = wv * 64 + lane, qr = q >> 1, qc = 64 + (q & 1);   // the two right-most columns
        if (wv * 64 < 2 * FS_H) {
            const unsigned long long m = compass(min(qr, FS_H - 1), qc, q < 2 * FS_H && ox - 1 + qc < w - 3 && oy - 1 + qr < h - 3);
            if ((m >> lane) & 1) seg[cnt + __popcll(m & below)] = (uint16_t)(qr * FS_W + qc);
            cnt += __popcll(m);
        }
        if (lane == 0) seg_n[wv] = cnt;
    }
    __syncthreads();
    const int pre1 = seg_n[0], pre2 = pre1 + seg_n[1], pre3 = pre2 + seg_n[2], nt = pre3 + seg_n[3];
    for (int i0 = 0; i0 < nt; i0 += 256) {
        const int i = i0 + threadIdx.x;
        int e = 0;
        if (i < nt) {
            const int sg = (i >= pre1) + (i >= pre2) + (i >= pre3);
            e = todo[sg * TODO_SEG + i - (sg == 0 ? 0 : sg == 1 ? pre1 : sg == 2 ? pre2 : pre3)];
        }
        // the corner list overwrites the list of survivors from its start: it holds fewer entries than have been read,
        // and an entry's position in its segment is never before its rank in the order of reading
        __syncthreads();
        const int kind = i < nt ? fast_corner_kind(P, e / FS_W + 3, e % FS_W + 6, fast_t) : 0;
        wave_append(kind != 0, (uint16_t)(e | (kind << 14)), corners, &n_corners);   // e < 4356 < 2^13
    }
    __syncthreads();
    const int nc = n_corners;
    for (int i = threadIdx.x; i < nc; i += 256) {
        const int e = corners[i] & 0x3FFF, kind = corners[i] >> 14;
        const int sr = e / FS_W, sc = e % FS_W;
        Sc[sr][sc] = (uint8_t)fast_score(P, sr + 3, sc + 6, kind);
    }
    // Survivors of the strict 3x3 NMS are collected in LDS first (at most 1 in 4 pixels can survive), so the tile costs
    // ONE returning global atomic on the per-(frame, level) counter plus one add per occupied histogram bin, instead of
    // one of each per key point on the same few words.
    // (keys and histogram reuse the pixel tile, which nobody reads after the scores: 19 KB of LDS, 8 workgroups per CU)
    static_assert(sizeof(P) >= (FT_H * FT_W / 4 + 8) * 4 + 256 * 4, "key list + histogram must fit into the pixel tile");
    uint32_t *keys = reinterpret_cast<uint32_t *>(&P[0][0]);
    int *lhist = reinterpret_cast<int *>(keys + FT_H * FT_W / 4 + 8);
    __shared__ int lcount, lbase;
    __syncthreads();
    lhist[threadIdx.x] = 0;
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nc; i += 256) {
        const int e = corners[i] & 0x3FFF;
        const int r = e / FS_W - 1, c = e % FS_W - 1;       // key point tile coordinates (the halo ring only feeds the NMS)
        if (r < 0 || r >= FT_H || c < 0 || c >= FT_W) continue;
        const int x = ox + c, y = oy + r;
        if (x >= w - EDGE || y >= h - EDGE) continue;
        const int s = Sc[r + 1][c + 1];
        if (s == 0) continue;
        const bool keep = s > Sc[r][c] && s > Sc[r][c + 1] && s > Sc[r][c + 2] && s > Sc[r + 1][c] &&
                          s > Sc[r + 1][c + 2] && s > Sc[r + 2][c] && s > Sc[r + 2][c + 1] && s > Sc[r + 2][c + 2];
        if (!keep) continue;
        const int slot = atomicAdd(&lcount, 1);
        keys[slot] = ((uint32_t)(255 - s) << 24) | ((uint32_t)y << 12) | (uint32_t)x;
        atomicAdd(&lhist[255 - s], 1);
    }
    __syncthreads();
    const int nloc = lcount;
    if (nloc == 0) return;
    const int seg = b * g.nlevels + l;
    int32_t *cnt = reinterpret_cast<int32_t *>(ws + g.cnt_off) + seg;
    int32_t *hist = reinterpret_cast<int32_t *>(ws + g.hist_off) + (size_t)seg * 256;
    uint32_t *cand = reinterpret_cast<uint32_t *>(ws + g.cand_off[l]) + (size_t)b * g.cand_cap[l];
    if (threadIdx.x == 0) lbase = atomicAdd(cnt, nloc);
    if (lhist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lhist[threadIdx.x]);
    __syncthreads();
    const int base = lbase;
    for (int e = threadIdx.x; e < nloc; e += 256)
        if (base + e < g.cand_cap[l]) cand[base + e] = keys[e];
}

// ---- exact selection of the 2n smallest keys per (frame, level) ----------------------------------------------------------
__global__ __launch_bounds__(256) void orb_select_kernel(OrbGeom g, uint8_t *__restrict__ ws) {
    __shared__ int hist[256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_k;
    const int seg = blockIdx.x;
    const int b = seg / g.nlevels, l = seg % g.nlevels;
    const int C = min(reinterpret_cast<int32_t *>(ws + g.cnt_off)[seg], g.cand_cap[l]);
    uint32_t *thr = reinterpret_cast<uint32_t *>(ws + g.thr_off) + seg;
    const int k = 2 * g.nfeat[l];
    if (C <= k) {
        if (threadIdx.x == 0) *thr = (k == 0) ? 0u : 0xFFFFFFFFu;
        return;
    }
    const uint32_t *cand = reinterpret_cast<const uint32_t *>(ws + g.cand_off[l]) + (size_t)b * g.cand_cap[l];
    const int32_t *hist0 = reinterpret_cast<const int32_t *>(ws + g.hist_off) + (size_t)seg * 256;
    hist[threadIdx.x] = hist0[threadIdx.x];
    if (threadIdx.x == 0) {
        s_prefix = 0;
        s_k = k;
    }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (pass > 0) {
            hist[threadIdx.x] = 0;
            __syncthreads();
            const uint32_t pre = s_prefix;
            for (int i = threadIdx.x; i < C; i += 256) {
                const uint32_t key = cand[i];
                if ((key >> (shift + 8)) == pre) atomicAdd(&hist[(key >> shift) & 255], 1);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            int kk = s_k, d = 0, cum = 0;
            for (; d < 256; ++d) {
                if (cum + hist[d] >= kk) break;
                cum += hist[d];
            }
            s_k = kk - cum;
            s_prefix = (s_prefix << 8) | (uint32_t)d;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *thr = s_prefix;  // the k-th smallest key; keys are unique
}

// ---- compaction + Harris ------------------------------------------------------------------------------------------------
// 25 x the Harris bracket of the 7 x 7 block around (x, y): a = sum Ix^2, b = sum Iy^2, c = sum Ix Iy with 3 x 3 Sobel
// derivatives.  The 9 x 9 window is read as three (unaligned) words per row; per row the horizontal differences
// D[c] = p[c+1] - p[c-1] and smoothings S[c] = p[c-1] + 2 p[c] + p[c+1] are formed once, Ix = D[r-1] + 2 D[r] + D[r+1],
// Iy = S[r+1] - S[r-1]; the sums fit 32 bits (|I| <= 1020, 49 terms).  (One thread per candidate reading 8 bytes per
// derivative pair from global memory was latency bound: 1.4 ms per clip.)
__device__ __forceinline__ long long harris25(const uint8_t *img, int pitch, int x, int y) {
    int D[3][7], S[3][7];      // rows r-1, r, r+1 (rotating)
    int a = 0, bb = 0, c = 0;
    auto load_row = [&](int r, int slot) __attribute__((always_inline)) {      // r = 0..8 <-> image row y - 4 + r
        const uint8_t *p = img + (size_t)(y - 4 + r) * pitch + (x - 4);
        uint32_t w0, w1, w2;
        __builtin_memcpy(&w0, p, 4);
        __builtin_memcpy(&w1, p + 4, 4);
        __builtin_memcpy(&w2, p + 8, 4);
        int v[9];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = (int)((w0 >> (8 * k)) & 255u);
            v[4 + k] = (int)((w1 >> (8 * k)) & 255u);
        }
        v[8] = (int)(w2 & 255u);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            D[slot][k] = v[k + 2] - v[k];
            S[slot][k] = v[k] + 2 * v[k + 1] + v[k + 2];
        }
    };
    load_row(0, 0);
    load_row(1, 1);
#pragma unroll
    for (int r = 1; r <= 7; ++r) {      // centre rows y - 3 .. y + 3
        load_row(r + 1, (r + 1) % 3);
        const int up = (r - 1) % 3, mid = r % 3, dn = (r + 1) % 3;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int ix = D[up][k] + 2 * D[mid][k] + D[dn][k];
            const int iy = S[dn][k] - S[up][k];
            a += ix * ix;
            bb += iy * iy;
            c += ix * iy;
        }
    }
    const long long A = a, B = bb, C = c;
    return 25 * (A * B - C * C) - (A + B) * (A + B);
}

__global__ __launch_bounds__(256) void orb_harris_kernel(OrbGeom g, const uint8_t *__restrict__ imgs,
                                                         uint8_t *__restrict__ ws) {
    const int seg = blockIdx.y;
    const int b = seg / g.nlevels, l = seg % g.nlevels;
    const int C = min(reinterpret_cast<int32_t *>(ws + g.cnt_off)[seg], g.cand_cap[l]);
    const uint32_t thr = reinterpret_cast<uint32_t *>(ws + g.thr_off)[seg];
    const uint32_t *cand = reinterpret_cast<const uint32_t *>(ws + g.cand_off[l]) + (size_t)b * g.cand_cap[l];
    int32_t *kcnt = reinterpret_cast<int32_t *>(ws + g.kcnt_off) + seg;
    uint32_t *kkey = reinterpret_cast<uint32_t *>(ws + g.kkey_off) + (size_t)seg * g.kcap;
    long long *kh = reinterpret_cast<long long *>(ws + g.kh_off) + (size_t)seg * g.kcap;
    int pitch;
    const uint8_t *img = level_ptr(g, imgs, ws, l, b, pitch);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < C; i += gridDim.x * 256) {
        const uint32_t key = cand[i];
        if (key > thr || g.nfeat[l] == 0) continue;
        const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
        const long long hv = harris25(img, pitch, x, y);
        const int slot = atomicAdd(kcnt, 1);
        if (slot < g.kcap) {
            kkey[slot] = key;
            kh[slot] = hv;
        }
    }
}

// ---- rank by (Harris desc, y asc, x asc), keep n, write keypoints in order ------------------------------------------------
__global__ __launch_bounds__(256) void orb_rank_kernel(OrbGeom g, uint8_t *__restrict__ ws, float *__restrict__ kp_xy,
                                                       int32_t *__restrict__ kp_meta, float *__restrict__ kp_resp,
                                                       int32_t *__restrict__ n_out) {
    __shared__ long long sh[256];
    __shared__ uint32_t sp[256];
    const int seg = blockIdx.y;
    const int b = seg / g.nlevels, l = seg % g.nlevels;
    const int32_t *kcnt = reinterpret_cast<const int32_t *>(ws + g.kcnt_off) + (size_t)b * g.nlevels;
    const int m = min(kcnt[l], g.kcap);
    int base = 0, total = 0;
    for (int q = 0; q < g.nlevels; ++q) {
        const int mq = min(min(kcnt[q], g.kcap), g.nfeat[q]);
        if (q < l) base += mq;
        total += mq;
    }
    if (l == 0 && blockIdx.x == 0 && threadIdx.x == 0) n_out[b] = total;
    if ((int)blockIdx.x * 256 >= m) return;
    const uint32_t *kkey = reinterpret_cast<const uint32_t *>(ws + g.kkey_off) + (size_t)seg * g.kcap;
    const long long *kh = reinterpret_cast<const long long *>(ws + g.kh_off) + (size_t)seg * g.kcap;
    const int i = blockIdx.x * 256 + threadIdx.x;
    long long hi = 0;
    uint32_t pi = 0;
    if (i < m) {
        hi = kh[i];
        pi = kkey[i] & 0xFFFFFFu;
    }
    // rank = #{j : H_j > H_i} + #{j : H_j == H_i and position_j < position_i}.  Equal Harris values at different positions
    // are rare, so the main loop only counts the first term and NOTES whether an equal value was seen (itself included:
    // more than one); the few threads that saw one redo the pass with the tie rule.
    int rank = 0, n_eq = 0;
    for (int j0 = 0; j0 < m; j0 += 256) {
        __syncthreads();
        if (j0 + (int)threadIdx.x < m) sh[threadIdx.x] = kh[j0 + threadIdx.x];
        __syncthreads();
        const int nj = min(256, m - j0);
        for (int j = 0; j < nj; ++j) {
            rank += sh[j] > hi;
            n_eq += sh[j] == hi;
        }
    }
    if (__syncthreads_or(i < m && n_eq > 1)) {      // (workgroup-uniform) ties: second pass over the positions
        for (int j0 = 0; j0 < m; j0 += 256) {
            __syncthreads();
            if (j0 + (int)threadIdx.x < m) {
                sh[threadIdx.x] = kh[j0 + threadIdx.x];
                sp[threadIdx.x] = kkey[j0 + threadIdx.x] & 0xFFFFFFu;
            }
            __syncthreads();
            if (n_eq > 1) {
                const int nj = min(256, m - j0);
                for (int j = 0; j < nj; ++j) rank += sh[j] == hi && sp[j] < pi;
            }
        }
    }
    if (i < m && rank < g.nfeat[l]) {
        const size_t o = (size_t)b * g.cap_out + base + rank;
        const int x = pi & 0xFFF, y = pi >> 12;
        kp_xy[o * 2] = (float)x * g.scale[l];
        kp_xy[o * 2 + 1] = (float)y * g.scale[l];
        kp_meta[o * 4] = l;
        kp_meta[o * 4 + 1] = x;
        kp_meta[o * 4 + 2] = y;
        kp_meta[o * 4 + 3] = (int32_t)(hi & 0xFFFFFFFFll);
        // cv2 response = (ab - c^2 - 0.04 (a+b)^2) * (1/(4*7*255))^4 ; `hi` is exactly 25x the bracket
        kp_resp[o] = (float)hi * (float)(1.0 / (25.0 * 7140.0 * 7140.0 * 7140.0 * 7140.0));
    }
}

// ---- the same ranking by SORTING (round 4): one workgroup per (frame, level) ----------------------------------------------------
// The counting kernel above compares every candidate with every other: (2 n_l)^2 comparisons per segment -- 12 M at level 0
// of an 8000-key-point frame (3.7x the time per frame for 2x the key points).  Here the (<= 4096) candidates of a segment
// are sorted in LDS by (Harris desc, position asc) -- a bitonic network, m log^2 m / 4 compare-exchanges per thread pair --
// and the element at sorted position r IS the key point of rank r: same order, same records, bit for bit.
constexpr int RK_MAX = 4096, RK_THREADS = 1024;
__global__ __launch_bounds__(RK_THREADS) void orb_rank_sort_kernel(OrbGeom g, uint8_t *__restrict__ ws, float *__restrict__ kp_xy,
                                                                   int32_t *__restrict__ kp_meta, float *__restrict__ kp_resp,
                                                                   int32_t *__restrict__ n_out) {
    __shared__ long long sH[RK_MAX];
    __shared__ uint32_t sP[RK_MAX];
    const int seg = blockIdx.x;
    const int b = seg / g.nlevels, l = seg % g.nlevels;
    const int32_t *kcnt = reinterpret_cast<const int32_t *>(ws + g.kcnt_off) + (size_t)b * g.nlevels;
    const int m = min(kcnt[l], g.kcap);
    int base = 0, total = 0;
    for (int q = 0; q < g.nlevels; ++q) {
        const int mq = min(min(kcnt[q], g.kcap), g.nfeat[q]);
        if (q < l) base += mq;
        total += mq;
    }
    if (l == 0 && threadIdx.x == 0) n_out[b] = total;
    if (m <= 0) return;
    const uint32_t *kkey = reinterpret_cast<const uint32_t *>(ws + g.kkey_off) + (size_t)seg * g.kcap;
    const long long *kh = reinterpret_cast<const long long *>(ws + g.kh_off) + (size_t)seg * g.kcap;
    int np2 = 64;
    while (np2 < m) np2 <<= 1;      // (<= RK_MAX: checked by the launcher)
    for (int i = threadIdx.x; i < np2; i += RK_THREADS) {
        // padding sorts behind every real entry: smallest Harris value, largest position
        sH[i] = i < m ? kh[i] : (long long)0x8000000000000000ull;
        sP[i] = i < m ? (kkey[i] & 0xFFFFFFu) : 0xFFFFFFFFu;
    }
    __syncthreads();
    // "a before b": larger Harris value first, then the smaller position (two candidates never share a position)
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (np2 >> 1); t += RK_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;      // the pair (i, i ^ j), i without bit j
                const bool up = (i & k) == 0;
                const long long hi_ = sH[i], hx = sH[ixj];
                const uint32_t pi_ = sP[i], px = sP[ixj];
                const bool x_first = hx > hi_ || (hx == hi_ && px < pi_);      // the element at ixj belongs in front
                if (x_first == up) {
                    sH[i] = hx;
                    sH[ixj] = hi_;
                    sP[i] = px;
                    sP[ixj] = pi_;
                }
            }
            __syncthreads();
        }
    }
    const int keep = min(m, g.nfeat[l]);
    for (int r = threadIdx.x; r < keep; r += RK_THREADS) {
        const long long hi = sH[r];
        const uint32_t pi = sP[r];
        const size_t o = (size_t)b * g.cap_out + base + r;
        const int x = pi & 0xFFF, y = pi >> 12;
        kp_xy[o * 2] = (float)x * g.scale[l];
        kp_xy[o * 2 + 1] = (float)y * g.scale[l];
        kp_meta[o * 4] = l;
        kp_meta[o * 4 + 1] = x;
        kp_meta[o * 4 + 2] = y;
        kp_meta[o * 4 + 3] = (int32_t)(hi & 0xFFFFFFFFll);
        kp_resp[o] = (float)hi * (float)(1.0 / (25.0 * 7140.0 * 7140.0 * 7140.0 * 7140.0));
    }
}

// ---- orientation + descriptor ----------------------------------------------------------------------------------------------
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
// blur: 7-tap sigma=2 Gaussian {18, 33, 49, 56, 49, 33, 18} / 256 in 8-bit fixed point (weights inlined in the kernel)

constexpr int DP = 39;  // raw patch side (31 + 2*(1 rotation slack) + 2*3 blur) -> offsets -19..19
constexpr int DB = 33;  // blurred patch side, offsets -16..16
// (waves per workgroup / key points per wave / waves per SIMD the register allocation aims at: 1 / 8 / 8 since the sweeps of
// round 4 -- 3.22 -> 3.05 ms; 4 / 8 / 5 before)
#ifndef MM_DESC_WAVES
#define MM_DESC_WAVES 1
#endif
#ifndef MM_DESC_KP
#define MM_DESC_KP 8
#endif
#ifndef MM_DESC_OCC
#define MM_DESC_OCC 8
#endif
constexpr int DESC_WAVES = MM_DESC_WAVES;
constexpr int DESC_KP_PER_WAVE = MM_DESC_KP;  // key points a wave describes one after the other (the next patch prefetched)

__global__ __launch_bounds__(64 * DESC_WAVES, MM_DESC_OCC) void orb_describe_kernel(OrbGeom g, const uint8_t *__restrict__ imgs,
                                                                       const uint8_t *__restrict__ ws,
                                                                       const int8_t *__restrict__ pattern,
                                                                       const int32_t *__restrict__ kp_meta,
                                                                       const int32_t *__restrict__ n_out,
                                                                       int32_t *__restrict__ kp_mom,
                                                                       uint8_t *__restrict__ desc) {
    // Per wave: the raw 39 x 39 patch as aligned words (byte j of a row = image column xa + j, xa = (x - 19) & ~3, so
    // patch column c sits at byte c + sh), the horizontally blurred patch TRANSPOSED as u16 (so that the vertical pass
    // reads its seven taps as four words), and the blurred 33 x 33 patch, which reuses the raw patch's storage.
    constexpr int RW = 12, HT = 40, BLP = 36;
    __shared__ __attribute__((aligned(16))) uint32_t raw_w[DESC_WAVES][DP * RW];
    static_assert(RW == 12 && (DP * RW * 4) % 16 == 0, "a patch row is three 16-byte pieces");
    __shared__ uint16_t hbt[DESC_WAVES][36 * HT];
    static_assert(DP * RW * 4 >= DB * BLP, "blurred patch reuses the raw patch");
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_kp = n_out[b];
    const int n_waves = gridDim.x * DESC_WAVES;
    int kp = blockIdx.x * DESC_WAVES + wv;
    if (kp >= n_kp) return;  // wave-uniform; no workgroup barriers below
    uint32_t *rw = raw_w[wv];
    const uint8_t *raw = reinterpret_cast<const uint8_t *>(rw);     // raw[r * 48 + c + sh]
    uint8_t *bl = reinterpret_cast<uint8_t *>(rw);                   // bl[r * BLP + c], written after the last read of raw
    uint16_t *ht = hbt[wv];
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // what does not depend on the key point: this lane's four sampling pairs and its row of the moment disc
    uint32_t patw[4];      // (kept packed: unpacked and converted to f64 inside the loop, or 32 registers are pinned)
#pragma unroll
    for (int q = 0; q < 4; ++q) patw[q] = reinterpret_cast<const uint32_t *>(pattern)[lane * 4 + q];
    const int mv = lane - 15, um = lane < 31 ? c_umax[mv < 0 ? -mv : mv] : -1;
    // A wave describes every n_waves-th key point of its frame.  The patch of the NEXT key point is requested (into
    // registers) before the current one is processed: the two dependent global accesses (key point record, then 39 row
    // segments scattered over a pyramid level) were 36 % of a wave's life and more on clips that do not fit the caches.
    // (16 bytes per lane and load: 39 rows x 3 pieces = 117 loads per patch instead of 429 words; an LDS row is 12 words, the
    // twelfth -- bytes 44 .. 47 -- is never read but lies inside the image row: key points keep 31 pixels from the border)
    constexpr int NPRE = (DP * 3 + 63) / 64;
    uint4 pre[NPRE];
    int x = 0, y = 0;
    auto request = [&](int k) {
        const size_t oo = (size_t)b * g.cap_out + k;
        const int l = kp_meta[oo * 4];
        x = kp_meta[oo * 4 + 1];
        y = kp_meta[oo * 4 + 2];
        int pitch;
        const uint8_t *img = level_ptr(g, imgs, ws, l, b, pitch);
        const int xa = (x - 19) & ~3;
        // 11 words per row cover bytes 0 .. 43 >= 38 + 3
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int e = min(lane + 64 * j, DP * 3 - 1), r = e / 3, pc = e % 3;
            pre[j] = load16_a4(img + (size_t)(y - 19 + r) * pitch + xa + 16 * pc);
        }
    };
    request(kp);
    const int lane_invariant = lane;
    for (;;) {
    // (opaque copy: otherwise every index expression below is hoisted out of the key point loop -- 118 registers, half
    // the occupancy)
    int lane = lane_invariant;
    asm volatile("" : "+v"(lane));
    const size_t o = (size_t)b * g.cap_out + kp;
    const int sh = (x - 19) & 3;
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
        const int e = lane + 64 * j;
        if (e < DP * 3) *reinterpret_cast<uint4 *>(rw + (e / 3) * RW + 4 * (e % 3)) = pre[j];
    }
    wave_sync();
    const int kp_next = kp + n_waves;
    if (kp_next < n_kp) request(kp_next);      // (x, y now belong to the next key point)
    // intensity-centroid moments over the radius-15 disc (rows v = -15..15 over lanes 0..30)
    int m10 = 0, m01 = 0;
    if (lane < 31) {
        int rs = 0;
        const uint8_t *row = raw + (19 + mv) * (4 * RW) + 19 + sh;
        for (int u = -um; u <= um; ++u) {
            const int val = row[u];
            m10 += u * val;
            rs += val;
        }
        m01 = mv * rs;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m10 += __shfl_xor(m10, off, 64);
        m01 += __shfl_xor(m01, off, 64);
    }
    if (lane == 0) {
        kp_mom[o * 2] = m10;
        kp_mom[o * 2 + 1] = m01;
    }
    // horizontal blur, four outputs (bytes 4k .. 4k+3 of the row, i.e. patch columns 4k - sh ..) per lane and step from
    // three words: taps 0..3 and 4..6 are one v_dot4_u32_u8 each on byte windows cut out with v_alignbyte.
    // {18,33,49,56 | 49,33,18,0}: exact integer sums <= 255 * 256.
    constexpr uint32_t GW_LO = 18u | (33u << 8) | (49u << 16) | (56u << 24), GW_HI = 49u | (33u << 8) | (18u << 16);
    for (int e = lane; e < DP * 9; e += 64) {
        const int r = e / 9, k = e % 9;
        const uint32_t w0 = rw[r * RW + k], w1 = rw[r * RW + k + 1], w2 = rw[r * RW + k + 2];
        uint32_t s[4];
        s[0] = __builtin_amdgcn_udot4(w1, GW_HI, __builtin_amdgcn_udot4(w0, GW_LO, 0u, false), false);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)q), hi = __builtin_amdgcn_alignbyte(w2, w1, (uint32_t)q);
            s[q] = __builtin_amdgcn_udot4(hi, GW_HI, __builtin_amdgcn_udot4(lo, GW_LO, 0u, false), false);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ht[(4 * k + q) * HT + r] = (uint16_t)s[q];
    }
    wave_sync();
    // vertical blur: two output rows (2 rp, 2 rp + 1) of one column per lane and step from the same four words of the
    // transposed rows (u16 pairs, v_dot2_u32_u16): the odd row only shifts the weights.  Rows 33 / taps past row 38 are
    // computed from padding and never read.
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 E0 = {18, 33}, E1 = {49, 56}, E2 = {49, 33}, E3 = {18, 0};
    const u16x2 O0 = {0, 18}, O1 = {33, 49}, O2 = {56, 49}, O3 = {33, 18};
    for (int e = lane; e < DB * 17; e += 64) {
        const int c = e / 17, rp = e % 17;
        const uint32_t *col = reinterpret_cast<const uint32_t *>(ht + (c + sh) * HT) + rp;
        const u16x2 a0 = __builtin_bit_cast(u16x2, col[0]), a1 = __builtin_bit_cast(u16x2, col[1]),
                    a2 = __builtin_bit_cast(u16x2, col[2]), a3 = __builtin_bit_cast(u16x2, col[3]);
        uint32_t se = __builtin_amdgcn_udot2(a0, E0, 32768u, false);
        se = __builtin_amdgcn_udot2(a1, E1, se, false);
        se = __builtin_amdgcn_udot2(a2, E2, se, false);
        se = __builtin_amdgcn_udot2(a3, E3, se, false);
        uint32_t so = __builtin_amdgcn_udot2(a0, O0, 32768u, false);
        so = __builtin_amdgcn_udot2(a1, O1, so, false);
        so = __builtin_amdgcn_udot2(a2, O2, so, false);
        so = __builtin_amdgcn_udot2(a3, O3, so, false);
        bl[(2 * rp) * BLP + c] = (uint8_t)(se >> 16);
        if (2 * rp + 1 < DB) bl[(2 * rp + 1) * BLP + c] = (uint8_t)(so >> 16);
    }
    wave_sync();
    // steering: cos = m10 / |m|, sin = m01 / |m| in IEEE f64 (no trig), offsets rounded half-to-even.
    // No FMA contraction here: the CPU oracle must reproduce every rounding (file is built with -ffp-contract=off).
    double cs = 1.0, sn = 0.0;
    {
        const long long q = (long long)m10 * m10 + (long long)m01 * m01;
        if (q > 0) {
            const double rr = __dsqrt_rn((double)q);
            cs = __ddiv_rn((double)m10, rr);
            sn = __ddiv_rn((double)m01, rr);
        }
    }
    // lane computes 4 bits: pairs 4*lane .. 4*lane+3
    uint32_t nib = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t pw = patw[k];
        asm volatile("" : "+v"(pw));      // not loop invariant as far as the optimiser is concerned
        const int8_t pp[4] = {(int8_t)pw, (int8_t)(pw >> 8), (int8_t)(pw >> 16), (int8_t)(pw >> 24)};
        int val[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const double px = (double)pp[2 * e], py = (double)pp[2 * e + 1];
            const double fx = __dsub_rn(__dmul_rn(px, cs), __dmul_rn(py, sn));
            const double fy = __dadd_rn(__dmul_rn(px, sn), __dmul_rn(py, cs));
            const int ix = (int)rint(fx), iy = (int)rint(fy);
            val[e] = bl[(16 + iy) * BLP + 16 + ix];
        }
        nib |= (uint32_t)(val[0] < val[1]) << k;
    }
    // two lanes per byte: low nibble from the even lane
    const uint32_t other = __shfl_xor(nib, 1, 64);
    if ((lane & 1) == 0) desc[o * 32 + (lane >> 1)] = (uint8_t)(nib | (other << 4));
    if (kp_next >= n_kp) break;
    kp = kp_next;
    wave_sync();      // every lane is done with the blurred patch before the next raw patch overwrites it
    }
}

// ---- host side geometry ---------------------------------------------------------------------------------------------------
int cv_round(double v) { return (int)__builtin_rint(v); }

int build_geom(int batch, int H, int W, int stride, const mm_orb_params *prm, OrbGeom &g, size_t &total) {
    if (!prm || prm->nlevels < 1 || prm->nlevels > MAXL || prm->nfeatures < 0 || H <= 0 || W <= 0 || batch <= 0)
        return MM_ERR_ARG;
    if (H >= 4096 || W >= 4096) return MM_ERR_ARG;  // 12-bit coordinates in the candidate key
    if (prm->edge_threshold != EDGE) return MM_ERR_ARG;
    memset(&g, 0, sizeof(g));
    g.nlevels = prm->nlevels;
    g.batch = batch;
    g.H = H;
    g.W = W;
    g.stride0 = stride;
    g.cap_out = prm->nfeatures;
    // OpenCV orb.cpp: scale_l = (float)pow(scaleFactor, l); size = cvRound(dim / scale_l)
    const float factor = (float)(1.0 / (double)prm->scale_factor);
    float ndes = (float)prm->nfeatures * (1.0f - factor) /
                 (1.0f - (float)__builtin_pow((double)factor, (double)prm->nlevels));
    int sum = 0;
    for (int l = 0; l < g.nlevels; ++l) {
        const float sc = (float)__builtin_pow((double)prm->scale_factor, (double)l);
        g.scale[l] = sc;
        g.w[l] = l == 0 ? W : cv_round((double)((float)W / sc));
        g.h[l] = l == 0 ? H : cv_round((double)((float)H / sc));
        g.pitch[l] = l == 0 ? stride : (int)mm_align_up((size_t)g.w[l], 64);
        if (l < g.nlevels - 1) {
            g.nfeat[l] = cv_round((double)ndes);
            sum += g.nfeat[l];
            ndes *= factor;
        } else {
            g.nfeat[l] = prm->nfeatures - sum > 0 ? prm->nfeatures - sum : 0;
        }
    }
    size_t off = 0;
    for (int l = 1; l < g.nlevels; ++l) {
        g.pyr_off[l] = off;
        g.pyr_bytes[l] = mm_align_up((size_t)g.pitch[l] * g.h[l] + 64, 256);
        off += g.pyr_bytes[l] * batch;
    }
    g.tile_start[0] = 0;
    for (int l = 0; l < g.nlevels; ++l) {
        const int vw = g.w[l] - 2 * EDGE, vh = g.h[l] - 2 * EDGE;
        int tx = 0, ty = 0;
        if (vw > 0 && vh > 0) {
            tx = (vw + FT_W - 1) / FT_W;
            ty = (vh + FT_H - 1) / FT_H;
        }
        g.tiles_x[l] = tx > 0 ? tx : 1;
        g.tile_start[l + 1] = g.tile_start[l] + tx * ty;
        g.cand_cap[l] = (vw > 0 && vh > 0) ? (int)(((size_t)vw * vh + 3) / 4 + 64) : 64;
        g.cand_off[l] = off;
        off += mm_align_up((size_t)g.cand_cap[l] * 4, 256) * batch;
        // per-frame stride of the candidate array must equal cand_cap entries: keep arrays dense per frame
    }
    for (int l = 1; l < g.nlevels; ++l) {
        g.tab_off[l] = off;
        off += mm_align_up((size_t)(2 * g.w[l] + 2 * g.h[l]) * 4, 256);
    }
    // the pyramid as runs of levels per launch, where the rectangles of a run fit the fused kernel's LDS buffers
    g.nchains = 0;
    for (int src = 0; src + 1 < g.nlevels; ) {
        PyrChain &ch = g.chain[g.nchains++];
        ch.src = src;
        ch.n = src == 0 ? (g.nlevels - 1 < 3 ? g.nlevels - 1 : 3) : g.nlevels - 1 - src;      // (MAXL = 8: 3 + 4)
        ch.ntx = (g.w[src] + PY_TW - 1) / PY_TW;
        ch.nty = (g.h[src] + PY_TH - 1) / PY_TH;
        ch.bnd_off = off;
        const int wx_ = py_dim_words(ch.n, ch.ntx), wy_ = py_dim_words(ch.n, ch.nty);
        off += mm_align_up((size_t)(wx_ + wy_) * 4, 256);
        std::vector<int32_t> bx(wx_), by(wy_);
        int dw[PY_MAXN + 1], dh[PY_MAXN + 1];
        for (int j = 0; j <= ch.n; ++j) {
            dw[j] = g.w[src + j];
            dh[j] = g.h[src + j];
        }
        for (int t = 0; t <= ch.ntx; ++t) py_bounds_lo(t, ch.ntx, PY_TW, ch.n, dw, bx.data());
        for (int t = 0; t <= ch.nty; ++t) py_bounds_lo(t, ch.nty, PY_TH, ch.n, dh, by.data());
        for (int t = 0; t < ch.ntx; ++t) py_bounds_nh(t, ch.ntx, ch.n, dw, bx.data());
        for (int t = 0; t < ch.nty; ++t) py_bounds_nh(t, ch.nty, ch.n, dh, by.data());
        bool fits = ch.ntx <= 65535 && ch.nty <= 65535;
        for (int j = 1; j <= ch.n && fits; ++j) {      // the kernel walks source rows: lower taps strictly increasing, one apart
            int prev_r1 = -1;
            for (int y = 0; y < dh[j] && fits; ++y) {
                const int y0 = rz_src0(y, dh[j], dh[j - 1]), r1 = std::min(y0 + 1, dh[j - 1] - 1);
                if (r1 <= prev_r1) fits = false;
                prev_r1 = r1;
            }
        }
        for (int j = 0; j <= ch.n && fits; ++j) {
            const int pitch_ = (j & 1) ? PY_PITCH_B : PY_PITCH_A, cap_ = (j & 1) ? PY_BUF_B : PY_BUF_A;
            for (int tx = 0; tx < ch.ntx && fits; ++tx) {
                const int lo = bx[py_x_lo(j, tx, ch.ntx)], nh = bx[py_x_nh(j, tx, ch.n, ch.ntx)];
                int ohx = bx[py_x_lo(j, tx + 1, ch.ntx)] - 1;
                if (j > 0 && tx == ch.ntx - 1) ohx = std::min(g.pitch[src + j] - 1, (dw[j] - 1) | 3);
                const int chx = std::max(nh, ohx);
                const int nq = chx >= lo ? (chx - (lo & ~3)) / 4 + 1 : 0;
                if (4 * nq + 4 > pitch_ || (j > 0 && 4 * nq > PY_TABX)) fits = false;
            }
            for (int ty = 0; ty < ch.nty && fits; ++ty) {
                const int rows = by[py_x_nh(j, ty, ch.n, ch.nty)] - by[py_x_lo(j, ty, ch.nty)] + 1;
                if ((j > 0 && rows > PY_TABY) || rows * pitch_ > cap_) fits = false;
            }
        }
        ch.fused = fits ? 1 : 0;
        src += ch.n;
        if (g.nchains == 2) break;
    }
    int nmax = 0;
    for (int l = 0; l < g.nlevels; ++l) nmax = g.nfeat[l] > nmax ? g.nfeat[l] : nmax;
    g.kcap = 2 * nmax + 8;
    const size_t segs = (size_t)batch * g.nlevels;
    g.cnt_off = off;
    off += mm_align_up(segs * 4, 256);
    g.kcnt_off = off;
    off += mm_align_up(segs * 4, 256);
    g.hist_off = off;
    off += mm_align_up(segs * 256 * 4, 256);
    g.thr_off = off;
    off += mm_align_up(segs * 4, 256);
    g.kkey_off = off;
    off += mm_align_up(segs * g.kcap * 4, 256);
    g.kh_off = off;
    off += mm_align_up(segs * g.kcap * 8, 256);
    total = off;
    return MM_OK;
}

}  // namespace

extern "C" {

int mm_orb_level_sizes(int height, int width, const mm_orb_params *prm, int32_t *lvl_w, int32_t *lvl_h, int32_t *lvl_n,
                       float *lvl_scale) {
    OrbGeom g;
    size_t total;
    int rc = build_geom(1, height, width, width, prm, g, total);
    if (rc) return rc;
    for (int l = 0; l < g.nlevels; ++l) {
        if (lvl_w) lvl_w[l] = g.w[l];
        if (lvl_h) lvl_h[l] = g.h[l];
        if (lvl_n) lvl_n[l] = g.nfeat[l];
        if (lvl_scale) lvl_scale[l] = g.scale[l];
    }
    return MM_OK;
}

size_t mm_orb_workspace_bytes(int batch, int height, int width, const mm_orb_params *prm) {
    OrbGeom g;
    size_t total = 0;
    if (build_geom(batch, height, width, width, prm, g, total)) return 0;
    return total;
}

int mm_orb_detect_compute(mm_ctx *ctx, const uint8_t *imgs, int batch, int height, int width, int stride,
                          const mm_orb_params *prm, const int8_t *pattern, void *ws, size_t ws_bytes, float *kp_xy,
                          int32_t *kp_meta, float *kp_resp, int32_t *kp_mom, uint8_t *desc, int32_t *n_out) {
    if (!ctx) return MM_ERR_ARG;
    if (batch == 0) return MM_OK;
    OrbGeom g;
    size_t total = 0;
    if (build_geom(batch, height, width, stride, prm, g, total))
        return mm_fail(ctx, MM_ERR_ARG, "mm_orb_detect_compute: bad geometry / parameters");
    if (!imgs || !pattern || !ws || !kp_xy || !kp_meta || !kp_resp || !kp_mom || !desc || !n_out)
        return mm_fail(ctx, MM_ERR_ARG, "mm_orb_detect_compute: null pointer");
    if (stride < width || (stride & 3) || ((uintptr_t)imgs & 3) || ((uintptr_t)ws & 255))
        return mm_fail(ctx, MM_ERR_ARG, "mm_orb_detect_compute: stride must be a multiple of 4 >= width, imgs 4-byte and ws 256-byte aligned");
    if (ws_bytes < total) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_orb_detect_compute: workspace %zu < %zu", ws_bytes, total);
    if (batch > 65535) return mm_fail(ctx, MM_ERR_ARG, "mm_orb_detect_compute: batch > 65535");
    uint8_t *w8 = (uint8_t *)ws;
    hipStream_t st = ctx->stream;
    // counters, kept counters and histograms are contiguous: [cnt_off, thr_off)
    MM_HIP(ctx, hipMemsetAsync(w8 + g.cnt_off, 0, g.thr_off - g.cnt_off, st));
    if (g.nlevels > 1) {
        const char *pyr_env = getenv("MM_ORB_PYRAMID");
        const bool per_level = pyr_env && pyr_env[0] == 'l';      // MM_ORB_PYRAMID=levels: one launch per level (cross-check)
        MM_LAUNCH(ctx, "orb_tables_kernel", orb_tables_kernel, dim3(g.nlevels - 1 + g.nchains), dim3(256), 0, g, w8);
        for (int c = 0; c < g.nchains; ++c) {
            const PyrChain &ch = g.chain[c];
            if (ch.fused && !per_level) {
                MM_LAUNCH(ctx, "orb_pyramid_kernel", orb_pyramid_kernel, dim3(ch.ntx, ch.nty, batch), dim3(PY_THREADS), 0, g, imgs, w8, c);
                continue;
            }
            for (int l = ch.src + 1; l <= ch.src + ch.n; ++l) {
                // output tile whose source footprint fits the kernel's LDS tile (256 x 16 for any scale factor <= 1.2)
                const double rx = (double)g.w[l - 1] / g.w[l], ry = (double)g.h[l - 1] / g.h[l];
                int tw = (int)((RZ_SW - 8) / rx) & ~3, th = (int)((RZ_SR - 2) / ry);
                tw = tw > RZ_TW ? RZ_TW : (tw < 4 ? 4 : tw);
                th = th > RZ_TH ? RZ_TH : (th < 1 ? 1 : th);
                if (rx > (RZ_SW - 8) / 4.0 || ry > RZ_SR - 2) return mm_fail(ctx, MM_ERR_ARG, "mm_orb_detect_compute: scale factor too large");
                dim3 grid((g.pitch[l] + tw - 1) / tw, (g.h[l] + th - 1) / th, batch);
                MM_LAUNCH(ctx, "orb_resize_kernel", orb_resize_kernel, grid, dim3(256), 0, g, imgs, w8, l, tw, th);
            }
        }
    }
    if (g.tile_start[g.nlevels] > 0) {
        MM_LAUNCH(ctx, "orb_fast_kernel", orb_fast_kernel, dim3(g.tile_start[g.nlevels], batch), dim3(256), 0, g, imgs, w8, prm->fast_threshold);
    }
    const int segs = batch * g.nlevels;
    MM_LAUNCH(ctx, "orb_select_kernel", orb_select_kernel, dim3(segs), dim3(256), 0, g, w8);
    MM_LAUNCH(ctx, "orb_harris_kernel", orb_harris_kernel, dim3(32, segs), dim3(256), 0, g, imgs, w8);
    // MM_ORB_RANK=count: the O(m^2) counting kernel (kept as the cross-check; also what larger capacities fall back to)
    static const bool rank_by_count = [] {
        const char *e = getenv("MM_ORB_RANK");
        return e && e[0] == 'c';
    }();
    if (g.kcap <= RK_MAX && !rank_by_count)
        MM_LAUNCH(ctx, "orb_rank_kernel", orb_rank_sort_kernel, dim3(segs), dim3(RK_THREADS), 0, g, w8, kp_xy, kp_meta, kp_resp, n_out);
    else
        MM_LAUNCH(ctx, "orb_rank_kernel", orb_rank_kernel, dim3((g.kcap + 255) / 256, segs), dim3(256), 0, g, w8, kp_xy, kp_meta, kp_resp, n_out);
    if (g.cap_out > 0) {
        MM_LAUNCH(ctx, "orb_describe_kernel", orb_describe_kernel, dim3((g.cap_out + DESC_WAVES * DESC_KP_PER_WAVE - 1) / (DESC_WAVES * DESC_KP_PER_WAVE), batch), dim3(64 * DESC_WAVES), 0, g, imgs, (const uint8_t *)w8, pattern, (const int32_t *)kp_meta, (const int32_t *)n_out, kp_mom, desc);
    }
    return MM_OK;
}

}  // extern "C"
