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
constexpr int MD_THREADS = 256;
constexpr int MD_GRID = 512;

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
    const int grid = (int)(g < 1 ? 1 : (g > MD_GRID ? MD_GRID : g));
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
