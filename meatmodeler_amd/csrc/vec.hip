// Vector reductions of the trust-region driver (gfx950): up to eight inner products over the parameter vector in ONE
// launch.  SciPy's TRF loop (reached through bundleAdjuster.py:180-192) takes ~13 inner products per iteration
// (|g_h|, the Gram-Schmidt coefficients of the 2-D subspace, the 2x2 model matrix, step norms); one BLAS dot per
// product costs two launches each and leaves the GPU idle in between, a fused pass reads each operand once per product
// at streaming rate.  Deterministic: fixed slices per workgroup, fixed trees, the workgroup that finishes last adds the
// per-workgroup partials in index order.  Every sum is reported in two parts (indices < split and >= split: the
// camera block is replicated across ranks, the point block is sharded).
#include "mm_common.h"
#include "ba_eval.h"

namespace {

constexpr int MD_MAX = 8;
#ifndef MM_MD_THREADS
#define MM_MD_THREADS 512      // (17.2 us per pass with 256 workgroups of 512 threads; 256 threads: 17.8, 128: 21.5, 1024: 19.4 -- round 4)
#endif
constexpr int MD_THREADS = MM_MD_THREADS;
constexpr int MD_GRID = 2048;      // capacity of the per-workgroup partials (workspace size); the launch uses md_grid_cap() of them
inline int md_grid_cap() {
    static const int cap = [] {
        const char *e = getenv("MM_VEC_GRID");
        const int v = e ? atoi(e) : 256;      // (17.9 us per pass at the bench shape; 384: 17.9, 512: 19.0, 768: 21.4, 128: 22.0 -- round 4)
        return v < 1 ? 1 : (v > MD_GRID ? MD_GRID : v);
    }();
    return cap;
}

struct MultiDotArgs {
    const double *a[MD_MAX];
    const double *b[MD_MAX];
};

template <int K>
__global__ __launch_bounds__(MD_THREADS) void multi_dot_kernel(MultiDotArgs args, int64_t n, int64_t split,
                                                               double *__restrict__ partial, unsigned *__restrict__ counter,
                                                               double *__restrict__ out) {
    __shared__ double sm[(MD_THREADS / 64) * 2 * K];
    __shared__ int s_last;
    double acc[2 * K];
#pragma unroll
    for (int q = 0; q < 2 * K; ++q) acc[q] = 0.0;
    // contiguous slice per workgroup, strided by the workgroup inside it: coalesced and independent of the grid mapping
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = per * blockIdx.x, hi = min(n, lo + per);
    for (int64_t i = lo + threadIdx.x; i < hi; i += MD_THREADS) {
        const bool pts = i >= split;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const double p = args.a[q][i] * args.b[q][i];
            acc[2 * q] += pts ? 0.0 : p;
            acc[2 * q + 1] += pts ? p : 0.0;
        }
    }
    block_sum_n<2 * K, MD_THREADS>(acc, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 2 * K; ++q)
            __hip_atomic_store(partial + (size_t)blockIdx.x * 2 * K + q, acc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have landed
        s_last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    // the last workgroup adds the per-workgroup partials: strided over the threads (all loads in flight at once), then
    // the same fixed tree as above
#pragma unroll
    for (int q = 0; q < 2 * K; ++q) acc[q] = 0.0;
    for (unsigned g = threadIdx.x; g < gridDim.x; g += MD_THREADS) {
#pragma unroll
        for (int q = 0; q < 2 * K; ++q)
            acc[q] += __hip_atomic_load(partial + (size_t)g * 2 * K + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    block_sum_n<2 * K, MD_THREADS>(acc, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            out[3 * q] = acc[2 * q];
            out[3 * q + 1] = acc[2 * q + 1];
            out[3 * q + 2] = acc[2 * q] + acc[2 * q + 1];
        }
    }
    if (threadIdx.x == 0) *counter = 0;  // ready for the next call on the same workspace
}

template <int K>
int launch_multi_dot(mm_ctx *ctx, const MultiDotArgs &args, int64_t n, int64_t split, double *partial, unsigned *counter,
                     double *out) {
    int64_t g = (n + 4 * MD_THREADS - 1) / (4 * MD_THREADS);
    const int grid = (int)(g < 1 ? 1 : (g > md_grid_cap() ? md_grid_cap() : g));
    MM_LAUNCH(ctx, "multi_dot_kernel", multi_dot_kernel<K>, dim3(grid), dim3(MD_THREADS), 0, args, n, split, partial, counter, out);
    return MM_OK;
}

}  // namespace

extern "C" {

size_t mm_multi_dot_workspace_bytes(void) { return 256 + (size_t)MD_GRID * 2 * MD_MAX * sizeof(double); }

int mm_multi_dot(mm_ctx *ctx, int k, const double *const *a, const double *const *b, int64_t n, int64_t split, double *out,
                 void *ws, size_t ws_bytes) {
    if (!ctx) return MM_ERR_ARG;
    if (k < 1 || k > MD_MAX || !a || !b || n < 0 || split < 0 || !out)
        return mm_fail(ctx, MM_ERR_ARG, "mm_multi_dot: bad argument (1 <= k <= %d)", MD_MAX);
    if (!ws || ws_bytes < mm_multi_dot_workspace_bytes() || ((uintptr_t)ws & 255))
        return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_multi_dot: workspace too small or misaligned");
    MultiDotArgs args;
    for (int q = 0; q < MD_MAX; ++q) {
        args.a[q] = a[q < k ? q : 0];
        args.b[q] = b[q < k ? q : 0];
        if (!args.a[q] || !args.b[q]) return mm_fail(ctx, MM_ERR_ARG, "mm_multi_dot: null operand");
    }
    unsigned *counter = (unsigned *)ws;  // zero on first use (the caller zero-fills the workspace once), reset by the kernel
    double *partial = (double *)((char *)ws + 256);
    switch (k) {
        case 1: return launch_multi_dot<1>(ctx, args, n, split, partial, counter, out);
        case 2: return launch_multi_dot<2>(ctx, args, n, split, partial, counter, out);
        case 3: return launch_multi_dot<3>(ctx, args, n, split, partial, counter, out);
        case 4: return launch_multi_dot<4>(ctx, args, n, split, partial, counter, out);
        case 5: return launch_multi_dot<5>(ctx, args, n, split, partial, counter, out);
        case 6: return launch_multi_dot<6>(ctx, args, n, split, partial, counter, out);
        case 7: return launch_multi_dot<7>(ctx, args, n, split, partial, counter, out);
        default: return launch_multi_dot<8>(ctx, args, n, split, partial, counter, out);
    }
}

}  // extern "C"

// ---- fused vector updates of the trust-region driver -------------------------------------------------------------------
// The 2-D subspace step of SciPy's TRF (trf.py:478-494) is a dozen element-wise passes over the parameter vector, each
// followed by an inner product that the next pass needs.  Written with framework calls that is ~40 small launches per
// iteration; here every pass also accumulates the inner products it is followed by (same deterministic reduction as
// mm_multi_dot: fixed slices, fixed trees, last workgroup adds the partials in index order; sums reported as camera
// part / point part / total so that a sharded run can all-reduce the point part).
namespace {

constexpr int FV_MAXK = 5;

struct FusedArgs {
    const double *in[6];
    double *out[3];
    const double *scalar[2];   // device scalars
    double h0, h1;             // host scalars
    void *ctab;                // OP 7: rotation coefficients of the F cameras at the head of out[0] (ba_eval.h's CamCoef rows)
    int F;
};

// OP 0: gh = g / si, ghs = gh / si;                      sums: gh.gh           (+ max |g| in the extra slot)
// OP 1: gn = q * si, q1 = gh / sqrt(gh2);                sums: q1.gn, gn.gn    (q = [v ; dp]: in[0] cameras, in[1] points)
// OP 2: w = gn - sc * q1;                                sums: w.w
// OP 3: q2 = w / sqrt(wn2), s1 = q1 / si, s2 = q2 / si;  sums: s1.s1, s1.s2, s2.s2, q2.gh, x.x
// OP 4: x_new = x + h0 * s1 + h1 * s2;                   no sums
// OP 5: the same with (h0, h1) = scalar[0][0..1] read from device memory (mm_trf_step2d); h1 == 0 skips s2 entirely
//       (a one-dimensional subspace leaves s2 = w / |w| with |w| = 0 undefined)
// OP 6: OP 0 preceded by mm_ba_scale_update(first = 0) on the same elements: si = max(sqrt(diag), si) from the camera
//       blocks in[2] [F,6,6] / packed point blocks in[3] [P,6], stored to out[2] (= in[1]) -- idempotent, so an iteration
//       that kept its blocks (rejected step) leaves si as it is
// OP 7: OP 5, and the cameras' rotation coefficients at x_new into `ctab` (what cam_coef_kernel would compute from out[0]:
//       the workgroups recompute those 3 F elements with the same fma chain)
template <int OP>
struct FusedTraits;
template <> struct FusedTraits<6> { static constexpr int K = 1; };
template <> struct FusedTraits<7> { static constexpr int K = 0; };
template <> struct FusedTraits<0> { static constexpr int K = 1; };
template <> struct FusedTraits<1> { static constexpr int K = 2; };
template <> struct FusedTraits<2> { static constexpr int K = 1; };
template <> struct FusedTraits<3> { static constexpr int K = 5; };
template <> struct FusedTraits<4> { static constexpr int K = 0; };
template <> struct FusedTraits<5> { static constexpr int K = 0; };

__device__ __forceinline__ double scaled_si(const FusedArgs &a, int64_t i, int64_t split, double si_old) {
    double d;
    if (i < split) {
        const int64_t f = i / 6, e = i % 6;
        d = a.in[2][f * 36 + e * 7];
    } else {
        const int64_t j = i - split, q = j / 3, e = j % 3;
        d = a.in[3][q * 6 + (e == 0 ? 0 : (e == 1 ? 3 : 5))];
    }
    return fmax(sqrt(d), si_old);
}
// x + h0 s1 (+ h1 s2): explicit fmas, the cameras' rotation coefficients (OP 7) are computed from the same chain
__device__ __forceinline__ double step_elem(double x, double s1, double s2, double h0, double h1) {
    double v = fma(h0, s1, x);
    if (h1 != 0.0) v = fma(h1, s2, v);
    return v;
}
template <int OP>
__device__ __forceinline__ void fused_elem(const FusedArgs &a, int64_t i, int64_t split, double (&p)[FV_MAXK], double &mx) {
    if constexpr (OP == 0 || OP == 6) {
        const double g = a.in[0][i];
        double si = a.in[1][i];
        if constexpr (OP == 6) {
            si = scaled_si(a, i, split, si);
            a.out[2][i] = si;
        }
        const double gh = g / si;
        a.out[0][i] = gh;
        a.out[1][i] = gh / si;
        p[0] = gh * gh;
        mx = fmax(mx, fabs(g));
    } else if constexpr (OP == 1) {
        const double q = i < split ? a.in[0][i] : a.in[1][i - split];
        const double gn = q * a.in[2][i];
        const double q1 = a.in[3][i] / sqrt(a.scalar[0][0]);
        a.out[0][i] = gn;
        a.out[1][i] = q1;
        p[0] = q1 * gn;
        p[1] = gn * gn;
    } else if constexpr (OP == 2) {
        const double w = a.in[0][i] - a.scalar[0][0] * a.in[1][i];
        a.out[0][i] = w;
        p[0] = w * w;
    } else if constexpr (OP == 3) {
        const double si = a.in[2][i];
        const double q2 = a.in[0][i] / sqrt(a.scalar[0][0]);
        const double s1 = a.in[1][i] / si, s2 = q2 / si;
        const double x = a.in[4][i];
        a.out[0][i] = q2;
        a.out[1][i] = s1;
        a.out[2][i] = s2;
        p[0] = s1 * s1;
        p[1] = s1 * s2;
        p[2] = s2 * s2;
        p[3] = q2 * a.in[3][i];
        p[4] = x * x;
    } else if constexpr (OP == 4) {
        a.out[0][i] = a.in[0][i] + a.h0 * a.in[1][i] + a.h1 * a.in[2][i];
    } else {
        const double h0 = a.scalar[0][0], h1 = a.scalar[0][1];
        a.out[0][i] = step_elem(a.in[0][i], a.in[1][i], h1 != 0.0 ? a.in[2][i] : 0.0, h0, h1);
    }
}

// two neighbouring elements (i even, both on the same side of `split`, which is even) with 16-byte loads / stores: half the
// memory instructions and twice the bytes in flight per lane -- the passes are bound by how much one workgroup per half CU
// keeps in flight, not by HBM (19 us for 40-80 MB)
__device__ __forceinline__ double2 ld2(const double *p, int64_t i) { return *reinterpret_cast<const double2 *>(p + i); }
__device__ __forceinline__ void st2(double *p, int64_t i, double x, double y) { *reinterpret_cast<double2 *>(p + i) = make_double2(x, y); }
template <int OP>
__device__ __forceinline__ void fused_elem2(const FusedArgs &a, int64_t i, int64_t split, double (&p)[FV_MAXK], double &mx) {
    if constexpr (OP == 0 || OP == 6) {
        const double2 g = ld2(a.in[0], i);
        double2 si = ld2(a.in[1], i);
        if constexpr (OP == 6) {
            si.x = scaled_si(a, i, split, si.x);
            si.y = scaled_si(a, i + 1, split, si.y);
            st2(a.out[2], i, si.x, si.y);
        }
        const double gh0 = g.x / si.x, gh1 = g.y / si.y;
        st2(a.out[0], i, gh0, gh1);
        st2(a.out[1], i, gh0 / si.x, gh1 / si.y);
        p[0] = gh0 * gh0;
        p[0] += gh1 * gh1;
        mx = fmax(mx, fmax(fabs(g.x), fabs(g.y)));
    } else if constexpr (OP == 1) {
        const double2 q = i < split ? ld2(a.in[0], i) : ld2(a.in[1], i - split);
        const double2 si = ld2(a.in[2], i), gh = ld2(a.in[3], i);
        const double rt = sqrt(a.scalar[0][0]);
        const double gn0 = q.x * si.x, gn1 = q.y * si.y;
        const double q10 = gh.x / rt, q11 = gh.y / rt;
        st2(a.out[0], i, gn0, gn1);
        st2(a.out[1], i, q10, q11);
        p[0] = q10 * gn0;
        p[0] += q11 * gn1;
        p[1] = gn0 * gn0;
        p[1] += gn1 * gn1;
    } else if constexpr (OP == 2) {
        const double2 gn = ld2(a.in[0], i), q1 = ld2(a.in[1], i);
        const double sc = a.scalar[0][0];
        const double w0 = gn.x - sc * q1.x, w1 = gn.y - sc * q1.y;
        st2(a.out[0], i, w0, w1);
        p[0] = w0 * w0;
        p[0] += w1 * w1;
    } else if constexpr (OP == 3) {
        const double2 w = ld2(a.in[0], i), q1 = ld2(a.in[1], i), si = ld2(a.in[2], i), gh = ld2(a.in[3], i), x = ld2(a.in[4], i);
        const double rt = sqrt(a.scalar[0][0]);
        const double q20 = w.x / rt, q21 = w.y / rt;
        const double s10 = q1.x / si.x, s11 = q1.y / si.y, s20 = q20 / si.x, s21 = q21 / si.y;
        st2(a.out[0], i, q20, q21);
        st2(a.out[1], i, s10, s11);
        st2(a.out[2], i, s20, s21);
        p[0] = s10 * s10; p[0] += s11 * s11;
        p[1] = s10 * s20; p[1] += s11 * s21;
        p[2] = s20 * s20; p[2] += s21 * s21;
        p[3] = q20 * gh.x; p[3] += q21 * gh.y;
        p[4] = x.x * x.x;  p[4] += x.y * x.y;
    } else if constexpr (OP == 4) {
        const double2 x = ld2(a.in[0], i), s1 = ld2(a.in[1], i), s2 = ld2(a.in[2], i);
        st2(a.out[0], i, x.x + a.h0 * s1.x + a.h1 * s2.x, x.y + a.h0 * s1.y + a.h1 * s2.y);
    } else {
        const double h0 = a.scalar[0][0], h1 = a.scalar[0][1];
        const double2 x = ld2(a.in[0], i), s1 = ld2(a.in[1], i);
        double2 s2 = make_double2(0.0, 0.0);
        if (h1 != 0.0) s2 = ld2(a.in[2], i);
        st2(a.out[0], i, step_elem(x.x, s1.x, s2.x, h0, h1), step_elem(x.y, s1.y, s2.y, h0, h1));
    }
}

// (body shared by the single-problem kernel and the batched one: bx / gx stand for blockIdx.x / gridDim.x)
template <int OP>
__device__ __forceinline__ void fused_vec_body(const FusedArgs &args, int64_t n, int64_t split, double *__restrict__ partial,
                                               unsigned *__restrict__ counter, double *__restrict__ out, const unsigned bx,
                                               const unsigned gx, const bool vec2) {
    constexpr int K = FusedTraits<OP>::K;
    constexpr int NA = 2 * K + 2;            // sums (camera, point) + the two maxima
    __shared__ double sm[(MD_THREADS / 64) * (NA > 0 ? NA : 1)];
    __shared__ int s_last;
    double acc[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) acc[q] = 0.0;
    // contiguous slice per workgroup (an even number of elements; pairs never straddle `split` when it is even)
    const int64_t per = ((n + gx - 1) / gx + 1) & ~(int64_t)1;
    const int64_t lo = per * bx, hi = min(n, lo + per);
    auto add = [&](int64_t i, const double (&p)[FV_MAXK], double mx) {
        const bool pts = i >= split;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            acc[2 * q] += pts ? 0.0 : p[q];
            acc[2 * q + 1] += pts ? p[q] : 0.0;
        }
        acc[2 * K + (pts ? 1 : 0)] = fmax(acc[2 * K + (pts ? 1 : 0)], mx);
    };
    for (int64_t i = lo + 2 * (int64_t)threadIdx.x; i < hi; i += 2 * MD_THREADS) {
        if (vec2 && i + 1 < hi) {
            double p[FV_MAXK] = {0, 0, 0, 0, 0}, mx = 0.0;
            fused_elem2<OP>(args, i, split, p, mx);
            add(i, p, mx);
        } else {      // (unaligned operands, an odd split, or the last element of an odd slice: element by element)
            for (int64_t j = i; j < i + 2 && j < hi; ++j) {
                double p[FV_MAXK] = {0, 0, 0, 0, 0}, mx = 0.0;
                fused_elem<OP>(args, j, split, p, mx);
                add(j, p, mx);
            }
        }
    }
    if constexpr (OP == 7) {
        const double h0 = args.scalar[0][0], h1 = args.scalar[0][1];
        for (int64_t f = (int64_t)bx * MD_THREADS + threadIdx.x; f < args.F; f += (int64_t)gx * MD_THREADS) {
            double r[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                r[k] = step_elem(args.in[0][f * 6 + k], args.in[1][f * 6 + k], h1 != 0.0 ? args.in[2][f * 6 + k] : 0.0, h0, h1);
            ((CamCoef *)args.ctab)[f] = cam_coef_of(r);
        }
    }
    if constexpr (K == 0) return;
    // maxima: reduce with max, sums with the fixed tree (the maxima are order-independent anyway)
    double m0 = acc[2 * K], m1 = acc[2 * K + 1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m0 = fmax(m0, __shfl_down(m0, off, 64));
        m1 = fmax(m1, __shfl_down(m1, off, 64));
    }
    block_sum_n<NA, MD_THREADS>(acc, sm);     // slots 2K, 2K+1 of `acc` are overwritten below
    __shared__ double smax[2][MD_THREADS / 64];
    if ((threadIdx.x & 63) == 0) {
        smax[0][threadIdx.x >> 6] = m0;
        smax[1][threadIdx.x >> 6] = m1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double b0 = 0.0, b1 = 0.0;
        for (int w = 0; w < MD_THREADS / 64; ++w) {
            b0 = fmax(b0, smax[0][w]);
            b1 = fmax(b1, smax[1][w]);
        }
        acc[2 * K] = b0;
        acc[2 * K + 1] = b1;
#pragma unroll
        for (int q = 0; q < NA; ++q)
            __hip_atomic_store(partial + (size_t)bx * NA + q, acc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(counter, 1u) == gx - 1;
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int q = 0; q < NA; ++q) acc[q] = 0.0;
    double g0 = 0.0, g1 = 0.0;
    for (unsigned g = threadIdx.x; g < gx; g += MD_THREADS) {
#pragma unroll
        for (int q = 0; q < 2 * K; ++q)
            acc[q] += __hip_atomic_load(partial + (size_t)g * NA + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g0 = fmax(g0, __hip_atomic_load(partial + (size_t)g * NA + 2 * K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        g1 = fmax(g1, __hip_atomic_load(partial + (size_t)g * NA + 2 * K + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        g0 = fmax(g0, __shfl_down(g0, off, 64));
        g1 = fmax(g1, __shfl_down(g1, off, 64));
    }
    acc[2 * K] = 0.0;
    acc[2 * K + 1] = 0.0;
    block_sum_n<NA, MD_THREADS>(acc, sm);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        smax[0][threadIdx.x >> 6] = g0;
        smax[1][threadIdx.x >> 6] = g1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            out[3 * q] = acc[2 * q];
            out[3 * q + 1] = acc[2 * q + 1];
            out[3 * q + 2] = acc[2 * q] + acc[2 * q + 1];
        }
        double b0 = 0.0, b1 = 0.0;
        for (int w = 0; w < MD_THREADS / 64; ++w) {
            b0 = fmax(b0, smax[0][w]);
            b1 = fmax(b1, smax[1][w]);
        }
        out[3 * K] = b0;
        out[3 * K + 1] = b1;
        out[3 * K + 2] = fmax(b0, b1);
        *counter = 0;
    }
}

template <int OP>
__global__ __launch_bounds__(MD_THREADS) void fused_vec_kernel(FusedArgs args, int64_t n, int64_t split,
                                                               double *__restrict__ partial, unsigned *__restrict__ counter,
                                                               double *__restrict__ out, bool vec2) {
    fused_vec_body<OP>(args, n, split, partial, counter, out, blockIdx.x, gridDim.x, vec2);
}

// the passes of mm_ba_trf's loop on the buffers of a batch record (operands as trf.hip passes them)
template <int OP>
__global__ __launch_bounds__(MD_THREADS) void fused_vec_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    const mm_batch_prob &bp = tab[list[blockIdx.y]];
    if (blockIdx.x >= bp.g_vec) return;
    FusedArgs a = {};
    double *rows = nullptr;
    if constexpr (OP == 0) {
        a.in[0] = bp.g; a.in[1] = bp.si; a.out[0] = bp.gh; a.out[1] = bp.ghs; rows = bp.r0;
    } else if constexpr (OP == 1) {
        a.in[0] = bp.v; a.in[1] = bp.dp; a.in[2] = bp.si; a.in[3] = bp.gh; a.out[0] = bp.gn; a.out[1] = bp.q1;
        a.scalar[0] = bp.r0 + 2; rows = bp.r1;
    } else if constexpr (OP == 2) {
        a.in[0] = bp.gn; a.in[1] = bp.q1; a.out[0] = bp.w; a.scalar[0] = bp.r1 + 2; rows = bp.r2;
    } else if constexpr (OP == 3) {
        a.in[0] = bp.w; a.in[1] = bp.q1; a.in[2] = bp.si; a.in[3] = bp.gh; a.in[4] = bp.x;
        a.out[0] = bp.q2; a.out[1] = bp.s1; a.out[2] = bp.s2; a.scalar[0] = bp.r2 + 2; rows = bp.r3;
    } else {
        static_assert(OP == 5, "ops 0-3 and 5 are the ones the loop issues");
        a.in[0] = bp.x; a.in[1] = bp.s1; a.in[2] = bp.s2; a.out[0] = bp.x_new; a.scalar[0] = bp.board;
    }
    fused_vec_body<OP>(a, bp.n, bp.nc, bp.md_partial, bp.md_counter, rows, blockIdx.x, bp.g_vec, true);
}

inline int fused_grid_of(int64_t n) {
    int64_t g = (n + 4 * MD_THREADS - 1) / (4 * MD_THREADS);
    return (int)(g < 1 ? 1 : (g > md_grid_cap() ? md_grid_cap() : g));
}
template <int OP>
int launch_fused(mm_ctx *ctx, const FusedArgs &args, int64_t n, int64_t split, double *partial, unsigned *counter, double *out) {
    const int grid = fused_grid_of(n);
    // 16-byte accesses need 16-byte aligned operands and an even split (the library's own buffers always are)
    bool vec2 = (split & 1) == 0;
    for (int q = 0; q < 6; ++q) vec2 = vec2 && (((uintptr_t)args.in[q]) & 15) == 0;
    for (int q = 0; q < 3; ++q) vec2 = vec2 && (((uintptr_t)args.out[q]) & 15) == 0;
    MM_LAUNCH(ctx, "fused_vec_kernel", fused_vec_kernel<OP>, dim3(grid), dim3(MD_THREADS), 0, args, n, split, partial, counter, out, vec2);
    return MM_OK;
}

}  // namespace

extern "C" int mm_trf_fused(mm_ctx *ctx, int op, const double *const *in, double *const *outv, const double *const *scalars,
                            double h0, double h1, int64_t n, int64_t split, double *out, void *ws, size_t ws_bytes) {
    if (!ctx) return MM_ERR_ARG;
    static const int n_in[6] = {2, 4, 2, 5, 3, 3}, n_out[6] = {2, 2, 1, 3, 1, 1}, n_sc[6] = {0, 1, 1, 1, 0, 1};
    if (op < 0 || op > 5 || !in || !outv || n < 0 || split < 0 || split > n) return mm_fail(ctx, MM_ERR_ARG, "mm_trf_fused: bad argument");
    if (!ws || ws_bytes < mm_multi_dot_workspace_bytes() || ((uintptr_t)ws & 255))
        return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_trf_fused: workspace too small or misaligned");
    FusedArgs a = {};
    for (int q = 0; q < n_in[op]; ++q) {
        if (!in[q]) return mm_fail(ctx, MM_ERR_ARG, "mm_trf_fused: null input %d", q);
        a.in[q] = in[q];
    }
    for (int q = 0; q < n_out[op]; ++q) {
        if (!outv[q]) return mm_fail(ctx, MM_ERR_ARG, "mm_trf_fused: null output %d", q);
        a.out[q] = outv[q];
    }
    for (int q = 0; q < n_sc[op]; ++q) {
        if (!scalars || !scalars[q]) return mm_fail(ctx, MM_ERR_ARG, "mm_trf_fused: null scalar %d", q);
        a.scalar[q] = scalars[q];
    }
    if (op < 4 && !out) return mm_fail(ctx, MM_ERR_ARG, "mm_trf_fused: null result");
    a.h0 = h0;
    a.h1 = h1;
    if (n == 0) {  // nothing to map; the sums (and maxima) of an empty vector are zero
        static const int rows[6] = {2, 3, 2, 6, 0, 0};
        if (rows[op]) MM_HIP(ctx, hipMemsetAsync(out, 0, (size_t)rows[op] * 3 * sizeof(double), ctx->stream));
        return MM_OK;
    }
    unsigned *counter = (unsigned *)ws;
    double *partial = (double *)((char *)ws + 256);
    switch (op) {
        case 0: return launch_fused<0>(ctx, a, n, split, partial, counter, out);
        case 1: return launch_fused<1>(ctx, a, n, split, partial, counter, out);
        case 2: return launch_fused<2>(ctx, a, n, split, partial, counter, out);
        case 3: return launch_fused<3>(ctx, a, n, split, partial, counter, out);
        case 4: return launch_fused<4>(ctx, a, n, split, partial, counter, out);
        default: return launch_fused<5>(ctx, a, n, split, partial, counter, out);
    }
}


// mm_common.h: the two passes of mm_ba_trf's loop that absorb a neighbouring small launch
int mm_trf_fused0_scaled(mm_ctx *ctx, const double *g, double *si, const double *B, const double *C, double *gh, double *ghs, int64_t n,
                         int64_t split, double *out, void *ws, size_t ws_bytes) {
    if (!ws || ws_bytes < mm_multi_dot_workspace_bytes() || ((uintptr_t)ws & 255))
        return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_trf_fused: workspace too small or misaligned");
    if (n <= 0 || (split % 6) || ((n - split) % 3)) return mm_fail(ctx, MM_ERR_ARG, "mm_trf_fused0_scaled: bad argument");
    FusedArgs a = {};
    a.in[0] = g; a.in[1] = si; a.in[2] = B; a.in[3] = C;
    a.out[0] = gh; a.out[1] = ghs; a.out[2] = si;
    FusedArgs al = a;      // (the block arrays are gathered element by element: only the vectors need the alignment)
    al.in[2] = al.in[3] = nullptr;
    bool vec2 = (split & 1) == 0;
    for (int q = 0; q < 6; ++q) vec2 = vec2 && (((uintptr_t)al.in[q]) & 15) == 0;
    for (int q = 0; q < 3; ++q) vec2 = vec2 && (((uintptr_t)al.out[q]) & 15) == 0;
    MM_LAUNCH(ctx, "fused_vec_kernel", fused_vec_kernel<6>, dim3(fused_grid_of(n)), dim3(MD_THREADS), 0, a, n, split,
              (double *)((char *)ws + 256), (unsigned *)ws, out, vec2);
    return MM_OK;
}
int mm_trf_fused5_coef(mm_ctx *ctx, const double *x, const double *s1, const double *s2, double *x_new, const double *h01, int64_t n,
                       int64_t split, void *ctab, int F) {
    if (n <= 0 || !ctab || (int64_t)F * 6 > n) return mm_fail(ctx, MM_ERR_ARG, "mm_trf_fused5_coef: bad argument");
    FusedArgs a = {};
    a.in[0] = x; a.in[1] = s1; a.in[2] = s2;
    a.out[0] = x_new;
    a.scalar[0] = h01;
    a.ctab = ctab;
    a.F = F;
    bool vec2 = (split & 1) == 0;
    for (int q = 0; q < 6; ++q) vec2 = vec2 && (((uintptr_t)a.in[q]) & 15) == 0;
    for (int q = 0; q < 3; ++q) vec2 = vec2 && (((uintptr_t)a.out[q]) & 15) == 0;
    MM_LAUNCH(ctx, "fused_vec_kernel", fused_vec_kernel<7>, dim3(fused_grid_of(n)), dim3(MD_THREADS), 0, a, n, split, (double *)nullptr,
              (unsigned *)nullptr, (double *)nullptr, vec2);
    return MM_OK;
}

// ---- block glue of the trust-region driver ---------------------------------------------------------------------------------
// scale_inv (SciPy x_scale='jac', common.py:598-610): si_i = sqrt((J^T J)_ii) from the diagonals of the camera blocks
// B [F,6,6] and of the packed point blocks C [P,6] (xx,xy,xz,yy,yz,zz); first call: zeros become 1, later calls: running
// maximum with the previous value.  One launch instead of diagonal / index / cat / sqrt / maximum.
namespace {
__device__ __forceinline__ void ba_scale_update_body(int64_t nc, int64_t n, const double *__restrict__ B,
                                                     const double *__restrict__ C, double *__restrict__ si, int first,
                                                     const unsigned bx) {
    const int64_t i = (int64_t)bx * 256 + threadIdx.x;
    if (i >= n) return;
    double d;
    if (i < nc) {
        const int64_t f = i / 6, a = i % 6;
        d = B[f * 36 + a * 7];
    } else {
        const int64_t j = i - nc, p = j / 3, a = j % 3;
        d = C[p * 6 + (a == 0 ? 0 : (a == 1 ? 3 : 5))];
    }
    double v = sqrt(d);
    if (first)
        v = v == 0.0 ? 1.0 : v;
    else
        v = fmax(v, si[i]);
    si[i] = v;
}

__global__ __launch_bounds__(256) void ba_scale_update_kernel(int64_t nc, int64_t n, const double *__restrict__ B,
                                                              const double *__restrict__ C, double *__restrict__ si,
                                                              int first) {
    ba_scale_update_body(nc, n, B, C, si, first, blockIdx.x);
}
__global__ __launch_bounds__(256) void ba_scale_update_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    const mm_batch_prob &bp = tab[list[blockIdx.y]];
    if (blockIdx.x >= bp.g_scale) return;
    ba_scale_update_body(bp.nc, bp.n, bp.B, bp.C, bp.si, 0, blockIdx.x);
}

// damped blocks  Bd = B + reg diag(si_c^2),  Cd = C + reg diag(si_p^2)  (reg read from device memory)
// dmp != nullptr: the damping itself is computed here from the iteration's scalars (every thread the same value, thread 0
// of workgroup 0 records {reg, max(reg, floor)} where trf_damping_kernel would have): one launch less per iteration
struct DampingIn {
    const double *gh2, *d11;
    double Delta, min_damping;
    double *out;
};
__device__ __forceinline__ void ba_damp_body(int64_t F, int64_t P, const double *__restrict__ B, const double *__restrict__ C,
                                             const double *__restrict__ si, const double *__restrict__ reg_p,
                                             double *__restrict__ Bd, double *__restrict__ Cd, const unsigned bx,
                                             const DampingIn *dmp = nullptr) {
    const int64_t i = (int64_t)bx * 256 + threadIdx.x;
    double reg;
    if (dmp) {
        const double r = trf_damping_value(dmp->gh2[0], dmp->d11[0], dmp->Delta);
        reg = fmax(r, dmp->min_damping);
        if (i == 0) {
            dmp->out[0] = r;
            dmp->out[1] = reg;
        }
    } else {
        reg = reg_p[0];
    }
    const int64_t nb = F * 36;
    if (i < nb) {
        const int64_t f = i / 36, e = i % 36;
        double v = B[i];
        if (e % 7 == 0) {
            const double s = si[f * 6 + e / 7];
            v = fma(s * s, reg, v);
        }
        Bd[i] = v;
    } else if (i < nb + P * 6) {
        const int64_t j = i - nb, p = j / 6, e = j % 6;
        double v = C[j];
        if (e == 0 || e == 3 || e == 5) {
            const double s = si[F * 6 + p * 3 + (e == 0 ? 0 : (e == 3 ? 1 : 2))];
            v = fma(s * s, reg, v);
        }
        Cd[j] = v;
    }
}
__global__ __launch_bounds__(256) void ba_damp_kernel(int64_t F, int64_t P, const double *__restrict__ B,
                                                      const double *__restrict__ C, const double *__restrict__ si,
                                                      const double *__restrict__ reg_p, double *__restrict__ Bd,
                                                      double *__restrict__ Cd) {
    ba_damp_body(F, P, B, C, si, reg_p, Bd, Cd, blockIdx.x);
}
__global__ __launch_bounds__(256) void ba_damp_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    const mm_batch_prob &bp = tab[list[blockIdx.y]];
    if (blockIdx.x >= bp.g_damp) return;
    ba_damp_body(bp.pb.F, bp.pb.P, bp.B, bp.C, bp.si, bp.damp + 1, bp.Bd, bp.Cd, blockIdx.x);
}
__global__ __launch_bounds__(256) void ba_damp_damping_kernel(int64_t F, int64_t P, const double *__restrict__ B,
                                                              const double *__restrict__ C, const double *__restrict__ si,
                                                              DampingIn dmp, double *__restrict__ Bd, double *__restrict__ Cd) {
    ba_damp_body(F, P, B, C, si, nullptr, Bd, Cd, blockIdx.x, &dmp);
}
// an accepted trial point becomes the iterate: x <- x_new, and the cameras' rotation coefficients with it (40 bytes per camera)
__global__ __launch_bounds__(256) void batch_accept_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    const mm_batch_prob &bp = tab[list[blockIdx.y]];
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < bp.n; i += stride) bp.x[i] = bp.x_new[i];
    const int64_t nt = (int64_t)bp.pb.F * 5;
    double *tx = (double *)bp.ctab_x;
    const double *tn = (const double *)bp.ctab_new;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nt; i += stride) tx[i] = tn[i];
}
}  // namespace

extern "C" int mm_ba_scale_update(mm_ctx *ctx, int F, int P, const double *B, const double *C, double *scale_inv, int first) {
    if (!ctx) return MM_ERR_ARG;
    if (F < 0 || P < 0 || (F > 0 && !B) || (P > 0 && !C) || !scale_inv) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_scale_update: bad argument");
    const int64_t n = (int64_t)F * 6 + (int64_t)P * 3;
    if (n == 0) return MM_OK;
    MM_LAUNCH(ctx, "ba_scale_update_kernel", ba_scale_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (int64_t)F * 6,
              n, B, C, scale_inv, first);
    return MM_OK;
}

// mm_trf_damping + mm_ba_damp in one launch (mm_ba_trf's first attempt of an iteration): damp_out [2] receives what
// mm_trf_damping writes
int mm_ba_damp_damping(mm_ctx *ctx, int F, int P, const double *B, const double *C, const double *scale_inv, const double *gh2,
                       const double *d11, double Delta, double min_damping, double *damp_out, double *Bd, double *Cd) {
    const int64_t n = (int64_t)F * 36 + (int64_t)P * 6;
    if (n == 0) return MM_OK;
    DampingIn dmp = {gh2, d11, Delta, min_damping, damp_out};
    MM_LAUNCH(ctx, "ba_damp_kernel", ba_damp_damping_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (int64_t)F, (int64_t)P, B, C,
              scale_inv, dmp, Bd, Cd);
    return MM_OK;
}

extern "C" int mm_ba_damp(mm_ctx *ctx, int F, int P, const double *B, const double *C, const double *scale_inv,
                          const double *reg, double *Bd, double *Cd) {
    if (!ctx) return MM_ERR_ARG;
    if (F < 0 || P < 0 || !scale_inv || !reg || (F > 0 && (!B || !Bd)) || (P > 0 && (!C || !Cd)))
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_damp: bad argument");
    const int64_t n = (int64_t)F * 36 + (int64_t)P * 6;
    if (n == 0) return MM_OK;
    MM_LAUNCH(ctx, "ba_damp_kernel", ba_damp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (int64_t)F, (int64_t)P, B, C,
              scale_inv, reg, Bd, Cd);
    return MM_OK;
}


// ---- the 2-D trust-region subproblem on the device ---------------------------------------------------------------------------
// SciPy solves  min 0.5 p^T B p + g^T p,  |p| <= Delta  in the plane span{g_h, gn_h} (trf.py:481-494 via
// common.py:171-219): Cholesky attempt for the interior solution, otherwise the stationary points of the model on the
// circle as the real roots of a quartic in tan(phi / 2).  Here the scalars that define B and g stay in device memory
// (they are results of the fused passes), one wave solves the subproblem and the trial point is formed from its
// output, so the host no longer has to read them back between building the subspace and the trial step: ONE host
// synchronisation per trial step instead of two.  Boundary case: see the comment in the kernel (the quartic's real
// roots by sampling + 64-ary search in two rational charts of the circle) -- the same minimiser as np.roots', to
// rounding.
namespace {
struct Step2dIn {
    const double *r0, *d11, *r1, *r2, *r3, *bs, *reg;
    const int32_t *info;
};

__device__ __forceinline__ double model_2d(double b00, double b01, double b11, double g0, double g1, double p0, double p1) {
    return 0.5 * (p0 * (b00 * p0 + b01 * p1) + p1 * (b01 * p0 + b11 * p1)) + (g0 * p0 + g1 * p1);
}

__device__ __forceinline__ void trf_step2d_body(const Step2dIn &in, double Delta, double *__restrict__ board, const double bs2, const double bs5) {
    const int lane = threadIdx.x;
    const double gh2 = in.r0[2], gmax = in.r0[5], d11 = in.d11[2], gn2 = in.r1[5], wn2 = in.r2[2];
    double n11 = in.r3[2], n12 = in.r3[5], n22 = in.r3[8], g2 = in.r3[11];
    const double xx = in.r3[14];
    const double u1Jq2 = bs2;
    double b22 = bs5;
    const double gh_norm = sqrt(gh2);
    const double b11 = d11 / gh2;
    double b12 = u1Jq2 / gh_norm;
    const bool degenerate = !(wn2 > 1e-28 * fmax(gn2, 1e-300));  // gn_h parallel to g_h: the subspace is one-dimensional
    if (degenerate) {
        b12 = 0.0;
        b22 = 1.0;
        n12 = 0.0;
        n22 = 0.0;
        g2 = 0.0;
    }
    const double b00 = b11, b01 = b12, bb = b22, g0 = gh_norm, g1 = g2;
    double p0 = 0.0, p1 = 0.0;
    bool interior = false;
    {   // interior solution if B is positive definite and the Newton point lies inside
        const double det = b00 * bb - b01 * b01;
        if (b00 > 0.0 && det > 0.0) {
            const double l00 = sqrt(b00), l10 = b01 / l00, l11sq = bb - l10 * l10;
            if (l11sq > 0.0) {
                const double l11 = sqrt(l11sq);
                const double y0 = -g0 / l00, y1 = (-g1 - l10 * y0) / l11;   // L y = -g
                p1 = y1 / l11;
                p0 = (y0 - l10 * p1) / l00;                                   // L^T p = y
                interior = p0 * p0 + p1 * p1 <= Delta * Delta;
            }
        }
    }
    if (!interior) {
        // Boundary: stationary points of the model on the circle |p| = Delta.  Two rational charts cover the circle,
        //   p(t) = Delta (2t, s (1 - t^2)) / (1 + t^2),  t in [-1, 1],  s = +1 (upper half), -1 (lower half);
        // d model / dt has the sign of the quartic N_s(t) below (the same quartic SciPy hands to np.roots, in both
        // charts instead of one chart on the whole real line).  No trigonometry: each lane samples N_s at t = -1 + l / 32,
        // every sign change is narrowed by 64-ary search across the lanes (9 rounds: double resolution), and the
        // stationary point with the smallest model value wins; the two chart junctions (+-Delta, 0) are candidates too.
        double best_val = 1.0e308, best_p0 = Delta, best_p1 = 0.0;
        auto consider = [&](double q0, double q1) {
            const double val = model_2d(b00, b01, bb, g0, g1, q0, q1);
            if (val < best_val) {
                best_val = val;
                best_p0 = q0;
                best_p1 = q1;
            }
        };
        consider(Delta, 0.0);
        consider(-Delta, 0.0);
        for (int chart = 0; chart < 2; ++chart) {
            const double sg = chart == 0 ? 1.0 : -1.0;
            auto N = [&](double t) {
                const double t2 = t * t, q = 1.0 + t2, u = 1.0 - t2;
                const double a0 = b00 * (2.0 * t * Delta) + b01 * (sg * Delta * u) + g0 * q;
                const double a1 = b01 * (2.0 * t * Delta) + bb * (sg * Delta * u) + g1 * q;
                return a0 * (2.0 * u) - sg * a1 * (4.0 * t);
            };
            const double hstep = 2.0 / 64.0;
            const double f_a = N(-1.0 + hstep * lane), f_b = N(-1.0 + hstep * (lane + 1));
            unsigned long long m = __ballot((f_a <= 0.0) != (f_b <= 0.0));
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                double lo = -1.0 + hstep * src, hi = lo + hstep;
                const bool lo_neg = N(lo) <= 0.0;   // (wave-uniform: every lane evaluates the same point)
                for (int round = 0; round < 9; ++round) {
                    const double wdt = (hi - lo) / 64.0;
                    const bool neg = N(lo + wdt * lane) <= 0.0;
                    // the sign change sits behind the LAST lane of the leading run that still has the sign of `lo`
                    const unsigned long long same = __ballot(neg == lo_neg);   // bit 0 (the sample at `lo`) is always set
                    const int k = (~same ? (int)__builtin_ctzll(~same) : 64) - 1;  // length of the leading run of ones - 1
                    lo = lo + wdt * k;
                    hi = lo + wdt;
                }
                const double t = 0.5 * (lo + hi), q = 1.0 + t * t;
                consider(Delta * (2.0 * t) / q, sg * Delta * (1.0 - t * t) / q);
            }
        }
        p0 = best_p0;
        p1 = best_p1;
    }
    if (degenerate) p1 = 0.0;
    if (lane == 0) {
        const double predicted = -model_2d(b00, b01, bb, g0, g1, p0, p1);
        board[0] = p0;
        board[1] = p1;
        board[2] = predicted;
        board[3] = sqrt(p0 * p0 + p1 * p1);
        board[4] = sqrt(fmax(p0 * p0 * n11 + 2.0 * p0 * p1 * n12 + p1 * p1 * n22, 0.0));
        board[5] = degenerate ? 1.0 : 0.0;
        board[6] = (double)in.info[0];
        board[7] = wn2;
        board[8] = gn2;
        board[9] = xx;
        board[10] = gmax;
        board[11] = gh2;
        board[12] = d11;
        board[13] = in.reg[0];
    }
}
__global__ __launch_bounds__(64) void trf_step2d_kernel(Step2dIn in, double Delta, double *__restrict__ board) {
    trf_step2d_body(in, Delta, board, in.bs[2], in.bs[5]);
}
// the same behind the final sum of the second Jacobian product's partials (jvp_rows_kernel's tree: 256 threads), one launch
// instead of two on the chain to the trial point; the sums reach the first wave through shared memory
__global__ __launch_bounds__(256) void trf_rows_step2d_kernel(const double *__restrict__ partial, unsigned n_wg, double *__restrict__ rows,
                                                              Step2dIn in, double Delta, double *__restrict__ board) {
    __shared__ double s_bs[2];
    const double2 t = jvp_rows_body(partial, n_wg, rows, 0, 1);
    if (threadIdx.x == 0) {
        s_bs[0] = t.x;
        s_bs[1] = t.y;
    }
    __syncthreads();
    if (threadIdx.x < 64) trf_step2d_body(in, Delta, board, s_bs[0], s_bs[1]);
}
__global__ __launch_bounds__(64) void trf_step2d_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list,
                                                              const mm_batch_dyn *__restrict__ dyn) {
    const int pid = list[blockIdx.x];
    const mm_batch_prob &bp = tab[pid];
    const Step2dIn in = {bp.r0, bp.d11, bp.r1, bp.r2, bp.r3, bp.bs, bp.damp + 1, bp.info};
    trf_step2d_body(in, dyn[pid].Delta, bp.board, in.bs[2], in.bs[5]);
}
}  // namespace

// ---- batched launches (mm_ba_trf_batched, trf.hip): grid = (largest per-problem grid, problems listed) ----------------------
int mm_batch_fused(mm_ctx *ctx, int op, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g) {
    if (n_list <= 0) return MM_OK;
    const dim3 grid(max_g, (unsigned)n_list);
    switch (op) {
        case 0: MM_LAUNCH(ctx, "fused_vec_kernel", fused_vec_batch_kernel<0>, grid, dim3(MD_THREADS), 0, tab, list); break;
        case 1: MM_LAUNCH(ctx, "fused_vec_kernel", fused_vec_batch_kernel<1>, grid, dim3(MD_THREADS), 0, tab, list); break;
        case 2: MM_LAUNCH(ctx, "fused_vec_kernel", fused_vec_batch_kernel<2>, grid, dim3(MD_THREADS), 0, tab, list); break;
        case 3: MM_LAUNCH(ctx, "fused_vec_kernel", fused_vec_batch_kernel<3>, grid, dim3(MD_THREADS), 0, tab, list); break;
        case 5: MM_LAUNCH(ctx, "fused_vec_kernel", fused_vec_batch_kernel<5>, grid, dim3(MD_THREADS), 0, tab, list); break;
        default: return mm_fail(ctx, MM_ERR_ARG, "mm_batch_fused: op %d", op);
    }
    return MM_OK;
}
unsigned mm_batch_fused_grid(int64_t n) { return (unsigned)fused_grid_of(n); }
int mm_batch_scale_update(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "ba_scale_update_kernel", ba_scale_update_batch_kernel, dim3(max_g, (unsigned)n_list), dim3(256), 0, tab, list);
    return MM_OK;
}
int mm_batch_damp(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "ba_damp_kernel", ba_damp_batch_kernel, dim3(max_g, (unsigned)n_list), dim3(256), 0, tab, list);
    return MM_OK;
}
int mm_batch_accept(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "batch_accept_kernel", batch_accept_kernel, dim3(max_g, (unsigned)n_list), dim3(256), 0, tab, list);
    return MM_OK;
}
int mm_batch_step2d(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, const mm_batch_dyn *dyn) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "trf_step2d_kernel", trf_step2d_batch_kernel, dim3((unsigned)n_list), dim3(64), 0, tab, list, dyn);
    return MM_OK;
}

int mm_trf_rows_step2d(mm_ctx *ctx, const double *partial, unsigned n_wg, double *rows, const double *r0, const double *d11,
                       const double *r1, const double *r2, const double *r3, const double *reg, const int32_t *info, double Delta,
                       double *board) {
    if (!partial || !n_wg || !rows || !(Delta >= 0.0)) return mm_fail(ctx, MM_ERR_ARG, "mm_trf_rows_step2d: bad argument");
    Step2dIn in = {r0, d11, r1, r2, r3, rows, reg, info};
    MM_LAUNCH(ctx, "trf_step2d_kernel", trf_rows_step2d_kernel, dim3(1), dim3(256), 0, partial, n_wg, rows, in, Delta, board);
    return MM_OK;
}

extern "C" int mm_trf_step2d(mm_ctx *ctx, const double *r0, const double *d11, const double *r1, const double *r2, const double *r3,
                             const double *bs, const double *reg, const int32_t *info, double Delta, double *board) {
    if (!ctx) return MM_ERR_ARG;
    if (!r0 || !d11 || !r1 || !r2 || !r3 || !bs || !reg || !info || !board || !(Delta >= 0.0))
        return mm_fail(ctx, MM_ERR_ARG, "mm_trf_step2d: bad argument");
    Step2dIn in = {r0, d11, r1, r2, r3, bs, reg, info};
    MM_LAUNCH(ctx, "trf_step2d_kernel", trf_step2d_kernel, dim3(1), dim3(64), 0, in, Delta, board);
    return MM_OK;
}
