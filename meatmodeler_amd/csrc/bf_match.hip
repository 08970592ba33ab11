// Brute-force Hamming 2-NN + Lowe ratio test for 256-bit descriptors (gfx950).
//
// Replaces cv2.FlannBasedMatcher(...).knnMatch(prev_desc, new_desc, k=2) and the ratio filter of
// processor.featureTracking (reference processor.py:132-137) by the exact search FLANN-LSH approximates.
//
// Mapping to the hardware:
//  * one lane owns one query descriptor (8 VGPRs); a wave covers 64 queries, a workgroup 256;
//  * the workgroup stages 128 train descriptors at a time in LDS (double buffered, the next chunk's global loads fly
//    during the current chunk's compute); every lane reads the same train with two same-address ds_read_b128 (LDS
//    broadcast) so that the eight v_xor_b32 take VGPR operands (2.6 cycles per wave instruction; 4.1 with an SGPR
//    operand, measured: profiles/r01_valu_issue_rates.txt); 8 v_bcnt_u32_b32 accumulate the popcounts;
//  * trains are handled in groups of four: a candidate can only enter the two smallest keys (key = dist << 20 | train
//    index, ties -> lowest train index) if its distance is below the current second-best distance, so the group's
//    minimum is tested and the min / med3 bookkeeping sits behind a wave-uniform, rarely taken branch:
//    18.8 VALU instructions per descriptor pair, 16 of them the xor / popcount floor;
//  * no cross-lane traffic, 0.02 B of HBM per pair: the bound is VALU integer issue.  Measured
//    with the SQ counters (profiles/r02_bf_pmc.txt): 3.84 cycles per VALU instruction at the 2.1 GHz the chip holds
//    under this load = 91 % of the issue roof of this instruction mix (v_bcnt / v_min issue at quarter rate);
//  * small launches split the train range over blockIdx.y and merge partial top-2 keys in a second kernel;
//  * the SGPR-fed variants (train descriptor by scalar loads, bf_knn2_kernel) are kept for MM_BF_VARIANT tuning runs.
#include "mm_common.h"
#include <cstdlib>

namespace {

constexpr int BF_THREADS = 256;
constexpr uint32_t BF_NONE = 0xFFFFFFFFu;
constexpr int BF_IDX_BITS = 20;

// popcount(x) + acc in ONE instruction.  Written as asm because hipcc otherwise re-associates the eight adds
// into v_bcnt + v_add3 trees (+3 VALU per pair, measured in the .s).
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

__device__ __forceinline__ uint32_t ham256(const uint32_t (&a)[8], const uint4 &t0, const uint4 &t1) {
    uint32_t d = bcnt_acc(a[0] ^ t0.x, 0u);
    d = bcnt_acc(a[1] ^ t0.y, d);
    d = bcnt_acc(a[2] ^ t0.z, d);
    d = bcnt_acc(a[3] ^ t0.w, d);
    d = bcnt_acc(a[4] ^ t1.x, d);
    d = bcnt_acc(a[5] ^ t1.y, d);
    d = bcnt_acc(a[6] ^ t1.z, d);
    d = bcnt_acc(a[7] ^ t1.w, d);
    return d;
}

__device__ __forceinline__ void top2_insert(uint32_t key, uint32_t &b0, uint32_t &b1) {
    // (b0 <= b1) and key  ->  two smallest of the three
    uint32_t hi = max(b0, key);
    b0 = min(b0, key);
    b1 = min(b1, hi);
}

template <int BF_QPL, int BF_UNROLL>
__global__ __launch_bounds__(BF_THREADS) void bf_knn2_kernel(
    const uint8_t *__restrict__ q, const int32_t *__restrict__ nq_dev, int nq_cap, size_t q_stride,
    const uint8_t *__restrict__ t, const int32_t *__restrict__ nt_dev, int nt_cap, size_t t_stride, int n_splits,
    uint32_t *__restrict__ part, int32_t *__restrict__ idx, int32_t *__restrict__ dist) {
    const int pair = blockIdx.z;
    const int split = blockIdx.y;
    int nq = nq_dev ? min(nq_dev[pair], nq_cap) : nq_cap;
    int nt = nt_dev ? min(nt_dev[pair], nt_cap) : nt_cap;
    constexpr int BF_QTILE = BF_THREADS * BF_QPL;  // queries per workgroup
    const int qbase = blockIdx.x * BF_QTILE;
    if (qbase >= nq) return;  // workgroup-uniform

    uint32_t qa[BF_QPL][8];
    const uint8_t *qp = q + (size_t)pair * q_stride;
#pragma unroll
    for (int u = 0; u < BF_QPL; ++u) {
        int qi = qbase + u * BF_THREADS + threadIdx.x;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (qi < nq) {
            const uint4 *p = reinterpret_cast<const uint4 *>(qp + (size_t)qi * 32);
            lo = p[0];
            hi = p[1];
        }
        qa[u][0] = lo.x; qa[u][1] = lo.y; qa[u][2] = lo.z; qa[u][3] = lo.w;
        qa[u][4] = hi.x; qa[u][5] = hi.y; qa[u][6] = hi.z; qa[u][7] = hi.w;
    }

    // train range of this split (multiples of the unroll so the unrolled body needs one tail only at the very end)
    int chunk = (nt + n_splits - 1) / n_splits;
    chunk = (chunk + BF_UNROLL - 1) / BF_UNROLL * BF_UNROLL;
    const int j0 = split * chunk;
    const int j1 = min(nt, j0 + chunk);

    uint32_t b0[BF_QPL], b1[BF_QPL];
#pragma unroll
    for (int u = 0; u < BF_QPL; ++u) b0[u] = b1[u] = BF_NONE;

    const uint4 *__restrict__ tp = reinterpret_cast<const uint4 *>(t + (size_t)pair * t_stride);
    int j = j0;
    for (; j + BF_UNROLL <= j1; j += BF_UNROLL) {
        uint4 tv[2 * BF_UNROLL];
#pragma unroll
        for (int k = 0; k < 2 * BF_UNROLL; ++k) tv[k] = tp[2 * j + k];  // wave-uniform address -> scalar loads
#pragma unroll
        for (int k = 0; k < BF_UNROLL; ++k) {
#pragma unroll
            for (int u = 0; u < BF_QPL; ++u) {
                uint32_t d = ham256(qa[u], tv[2 * k], tv[2 * k + 1]);
                top2_insert((d << BF_IDX_BITS) | (uint32_t)(j + k), b0[u], b1[u]);
            }
        }
    }
    for (; j < j1; ++j) {
        uint4 t0 = tp[2 * j], t1 = tp[2 * j + 1];
#pragma unroll
        for (int u = 0; u < BF_QPL; ++u) {
            uint32_t d = ham256(qa[u], t0, t1);
            top2_insert((d << BF_IDX_BITS) | (uint32_t)j, b0[u], b1[u]);
        }
    }

#pragma unroll
    for (int u = 0; u < BF_QPL; ++u) {
        int qi = qbase + u * BF_THREADS + threadIdx.x;
        if (qi >= nq) continue;
        if (n_splits == 1) {
            size_t o = ((size_t)pair * nq_cap + qi) * 2;
            idx[o] = b0[u] == BF_NONE ? -1 : (int32_t)(b0[u] & ((1u << BF_IDX_BITS) - 1));
            idx[o + 1] = b1[u] == BF_NONE ? -1 : (int32_t)(b1[u] & ((1u << BF_IDX_BITS) - 1));
            dist[o] = b0[u] == BF_NONE ? -1 : (int32_t)(b0[u] >> BF_IDX_BITS);
            dist[o + 1] = b1[u] == BF_NONE ? -1 : (int32_t)(b1[u] >> BF_IDX_BITS);
        } else {
            size_t o = (((size_t)pair * n_splits + split) * nq_cap + qi) * 2;
            part[o] = b0[u];
            part[o + 1] = b1[u];
        }
    }
}

// LDS-fed variant.  Measured on MI355X (profiles/r01_valu_issue_rates.txt): v_xor_b32 issues in 2 cycles per wave with
// VGPR operands but in 4 with an SGPR operand; v_bcnt_u32_b32 / v_min / v_med3 / v_lshl_or take 4.  Feeding the train
// descriptor from VGPRs (two same-address ds_read_b128 = LDS broadcast, on the LDS pipe, not the VALU) makes the eight
// xors half price: 8*2 + 8*4 + 3*4 = 60 cycles per descriptor pair instead of ~82.  The workgroup stages BF_TCHUNK train
// descriptors at a time (registers -> LDS, double buffered: the next chunk's global loads fly during the current
// chunk's compute); all four waves read the same chunk.
constexpr int BF_TCHUNK = 128;  // trains per LDS stage: 4 KB, one 16-byte piece per thread

__device__ __forceinline__ void top2_insert_med3(uint32_t key, uint32_t &b0, uint32_t &b1) {
    uint32_t m;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(b0), "v"(b1), "v"(key));  // middle of three = new second smallest
    b0 = min(b0, key);
    b1 = m;
}

template <int BF_QPL, int TCH = BF_TCHUNK>
__global__ __launch_bounds__(BF_THREADS) void bf_knn2_lds_kernel(
    const uint8_t *__restrict__ q, const int32_t *__restrict__ nq_dev, int nq_cap, size_t q_stride,
    const uint8_t *__restrict__ t, const int32_t *__restrict__ nt_dev, int nt_cap, size_t t_stride, int n_splits,
    uint32_t *__restrict__ part, int32_t *__restrict__ idx, int32_t *__restrict__ dist) {
    constexpr int NST = TCH / 128;  // 16-byte pieces per thread and stage
    __shared__ uint4 tl[2][TCH * 2];
    const int pair = blockIdx.z;
    const int split = blockIdx.y;
    int nq = nq_dev ? min(nq_dev[pair], nq_cap) : nq_cap;
    int nt = nt_dev ? min(nt_dev[pair], nt_cap) : nt_cap;
    constexpr int BF_QTILE = BF_THREADS * BF_QPL;
    const int qbase = blockIdx.x * BF_QTILE;
    if (qbase >= nq) return;  // workgroup-uniform

    uint32_t qa[BF_QPL][8];
    const uint8_t *qp = q + (size_t)pair * q_stride;
#pragma unroll
    for (int u = 0; u < BF_QPL; ++u) {
        int qi = qbase + u * BF_THREADS + threadIdx.x;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (qi < nq) {
            const uint4 *p = reinterpret_cast<const uint4 *>(qp + (size_t)qi * 32);
            lo = p[0];
            hi = p[1];
        }
        qa[u][0] = lo.x; qa[u][1] = lo.y; qa[u][2] = lo.z; qa[u][3] = lo.w;
        qa[u][4] = hi.x; qa[u][5] = hi.y; qa[u][6] = hi.z; qa[u][7] = hi.w;
    }
    int chunk = (nt + n_splits - 1) / n_splits;
    chunk = (chunk + 3) & ~3;
    const int j0 = split * chunk;
    const int j1 = min(nt, j0 + chunk);
    uint32_t b0[BF_QPL], b1[BF_QPL];
#pragma unroll
    for (int u = 0; u < BF_QPL; ++u) b0[u] = b1[u] = BF_NONE;

    const uint4 *__restrict__ tp = reinterpret_cast<const uint4 *>(t + (size_t)pair * t_stride);
    // stage 0
    uint4 stage[NST];
#pragma unroll
    for (int z = 0; z < NST; ++z) {
        const int e = 2 * j0 + z * BF_THREADS + (int)threadIdx.x;
        stage[z] = e < 2 * j1 ? tp[e] : make_uint4(0, 0, 0, 0);
        tl[0][z * BF_THREADS + threadIdx.x] = stage[z];
    }
    __syncthreads();
    int buf = 0;
    for (int c0 = j0; c0 < j1; c0 += TCH) {
        const int cn = min(TCH, j1 - c0);
        const int nxt = c0 + TCH;
        if (nxt < j1) {  // prefetch the next chunk into registers (in flight during the compute below)
#pragma unroll
            for (int z = 0; z < NST; ++z) {
                const int e = 2 * nxt + z * BF_THREADS + (int)threadIdx.x;
                stage[z] = e < 2 * j1 ? tp[e] : make_uint4(0, 0, 0, 0);
            }
        }
        const uint4 *tb = tl[buf];
        int k = 0;
        for (; k + 4 <= cn; k += 4) {
            uint4 tv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) tv[r] = tb[2 * k + r];  // same address in every lane: LDS broadcast
            uint32_t d[4][BF_QPL];
            bool hit = false;
#pragma unroll
            for (int u = 0; u < BF_QPL; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r][u] = ham256(qa[u], tv[2 * r], tv[2 * r + 1]);
                // train indices only grow, so a candidate enters the two smallest keys iff its DISTANCE is strictly
                // below the current second-best distance; test the group's minimum and skip the bookkeeping otherwise
                const uint32_t m = min(min(d[0][u], d[1][u]), min(d[2][u], d[3][u]));
                hit |= m < (b1[u] >> BF_IDX_BITS);
            }
            if (__builtin_amdgcn_ballot_w64(hit) != 0) {  // wave-uniform branch, rarely taken after the first trains
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int u = 0; u < BF_QPL; ++u)
                        top2_insert_med3((d[r][u] << BF_IDX_BITS) | (uint32_t)(c0 + k + r), b0[u], b1[u]);
                }
            }
        }
        for (; k < cn; ++k) {
            const uint4 t0 = tb[2 * k], t1 = tb[2 * k + 1];
#pragma unroll
            for (int u = 0; u < BF_QPL; ++u) {
                uint32_t d = ham256(qa[u], t0, t1);
                top2_insert_med3((d << BF_IDX_BITS) | (uint32_t)(c0 + k), b0[u], b1[u]);
            }
        }
        if (nxt < j1) {
#pragma unroll
            for (int z = 0; z < NST; ++z) tl[buf ^ 1][z * BF_THREADS + threadIdx.x] = stage[z];
            __syncthreads();
            buf ^= 1;
        }
    }
#pragma unroll
    for (int u = 0; u < BF_QPL; ++u) {
        int qi = qbase + u * BF_THREADS + threadIdx.x;
        if (qi >= nq) continue;
        if (n_splits == 1) {
            size_t o = ((size_t)pair * nq_cap + qi) * 2;
            idx[o] = b0[u] == BF_NONE ? -1 : (int32_t)(b0[u] & ((1u << BF_IDX_BITS) - 1));
            idx[o + 1] = b1[u] == BF_NONE ? -1 : (int32_t)(b1[u] & ((1u << BF_IDX_BITS) - 1));
            dist[o] = b0[u] == BF_NONE ? -1 : (int32_t)(b0[u] >> BF_IDX_BITS);
            dist[o + 1] = b1[u] == BF_NONE ? -1 : (int32_t)(b1[u] >> BF_IDX_BITS);
        } else {
            size_t o = (((size_t)pair * n_splits + split) * nq_cap + qi) * 2;
            part[o] = b0[u];
            part[o + 1] = b1[u];
        }
    }
}

__global__ __launch_bounds__(256) void bf_merge_kernel(const uint32_t *__restrict__ part,
                                                       const int32_t *__restrict__ nq_dev, int nq_cap, int n_splits,
                                                       int32_t *__restrict__ idx, int32_t *__restrict__ dist) {
    const int pair = blockIdx.y;
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    int nq = nq_dev ? min(nq_dev[pair], nq_cap) : nq_cap;
    if (qi >= nq) return;
    uint32_t b0 = BF_NONE, b1 = BF_NONE;
    for (int s = 0; s < n_splits; ++s) {
        size_t o = (((size_t)pair * n_splits + s) * nq_cap + qi) * 2;
        top2_insert(part[o], b0, b1);
        top2_insert(part[o + 1], b0, b1);
    }
    size_t o = ((size_t)pair * nq_cap + qi) * 2;
    idx[o] = b0 == BF_NONE ? -1 : (int32_t)(b0 & ((1u << BF_IDX_BITS) - 1));
    idx[o + 1] = b1 == BF_NONE ? -1 : (int32_t)(b1 & ((1u << BF_IDX_BITS) - 1));
    dist[o] = b0 == BF_NONE ? -1 : (int32_t)(b0 >> BF_IDX_BITS);
    dist[o + 1] = b1 == BF_NONE ? -1 : (int32_t)(b1 >> BF_IDX_BITS);
}

// Order-preserving compaction of the queries that pass the ratio test; one workgroup per pair.
__global__ __launch_bounds__(256) void ratio_filter_kernel(const int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ dist,
                                                           const int32_t *__restrict__ nq_dev, int nq_cap,
                                                           double threshold, int32_t *__restrict__ pairs,
                                                           int32_t *__restrict__ m_out) {
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int pair = blockIdx.x;
    int nq = nq_dev ? min(nq_dev[pair], nq_cap) : nq_cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 256) {
        int qi = q0 + threadIdx.x;
        bool keep = false;
        int ti = -1;
        if (qi < nq) {
            size_t o = ((size_t)pair * nq_cap + qi) * 2;
            int d0 = dist[o], d1 = dist[o + 1];
            ti = idx[o];
            keep = (d0 >= 0) && (d1 >= 0) && ((double)d0 < threshold * (double)d1);
        }
        unsigned long long m = __ballot(keep);
        int before = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __builtin_popcountll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (keep) {
            size_t o = ((size_t)pair * nq_cap + (off + before)) * 2;
            pairs[o] = qi;
            pairs[o + 1] = ti;
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) m_out[pair] = base_s;
}

// queries per lane: 2 amortises the scalar train loads over two descriptor pairs; small launches use 1 to get more waves.
// MM_BF_VARIANT=<qpl><unroll> (e.g. 24, 28, 44, 14) overrides for tuning runs.
int bf_variant(int n_pairs, int nq_cap) {
    const char *e = getenv("MM_BF_VARIANT");
    const int forced = e ? atoi(e) : 0;
    if (forced) return forced;
    return 114;  // LDS-fed, one query per lane, skip-branch bookkeeping: fastest of the measured variants (profiles/r01_bf_variants.txt)
}

int bf_choose_splits(int n_pairs, int nq_cap, int nt_cap) {
    const int BF_QTILE = BF_THREADS * ((bf_variant(n_pairs, nq_cap) / 10) % 10);
    long waves = (long)n_pairs * ((nq_cap + BF_QTILE - 1) / BF_QTILE) * (BF_THREADS / 64);
    if (waves <= 0) return 1;
    long s = (2048 + waves - 1) / waves;  // aim for >= 2 waves per SIMD on 256 CUs
    long smax = (nt_cap + 127) / 128;     // keep >= 128 trains per split
    if (s > smax) s = smax;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return (int)s;
}

}  // namespace

extern "C" {

size_t mm_bf_workspace_bytes(int n_pairs, int nq_cap, int nt_cap) {
    int s = bf_choose_splits(n_pairs, nq_cap, nt_cap);
    if (s == 1) return 256;
    return mm_align_up((size_t)n_pairs * s * nq_cap * 2 * sizeof(uint32_t), 256);
}

int mm_bf_knn2_batched(mm_ctx *ctx, const uint8_t *q, const int32_t *nq, int nq_cap, size_t q_set_stride,
                       const uint8_t *t, const int32_t *nt, int nt_cap, size_t t_set_stride, int n_pairs,
                       int32_t *idx, int32_t *dist, void *ws, size_t ws_bytes) {
    if (!ctx) return MM_ERR_ARG;
    if (n_pairs == 0 || nq_cap == 0) return MM_OK;
    if (!q || (!t && nt_cap > 0) || !idx || !dist || n_pairs < 0 || nq_cap < 0 || nt_cap < 0)
        return mm_fail(ctx, MM_ERR_ARG, "mm_bf_knn2_batched: bad argument");
    if (nt_cap >= (1 << BF_IDX_BITS)) return mm_fail(ctx, MM_ERR_ARG, "mm_bf_knn2_batched: nt_cap must be < 2^20");
    if (((uintptr_t)q | (uintptr_t)t | q_set_stride | t_set_stride) & 15)
        return mm_fail(ctx, MM_ERR_ARG, "mm_bf_knn2_batched: descriptors must be 16-byte aligned");
    if (n_pairs > 65535) return mm_fail(ctx, MM_ERR_ARG, "mm_bf_knn2_batched: at most 65535 pairs per call");
    int s = bf_choose_splits(n_pairs, nq_cap, nt_cap);
    if (s > 1 && (!ws || ws_bytes < mm_bf_workspace_bytes(n_pairs, nq_cap, nt_cap)))
        return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_bf_knn2_batched: workspace too small");
    const int var = bf_variant(n_pairs, nq_cap);
    const int qtile = BF_THREADS * ((var / 10) % 10);
    dim3 grid((nq_cap + qtile - 1) / qtile, s, n_pairs);
#define BF_GO(Q, U)                                                                                              \
    MM_LAUNCH(ctx, "bf_knn2_kernel", (bf_knn2_kernel<Q, U>), grid, dim3(BF_THREADS), 0, q, nq, nq_cap, q_set_stride, t, \
              nt, nt_cap, t_set_stride, s, (uint32_t *)ws, idx, dist)
    switch (var) {
        case 14: BF_GO(1, 4); break;
        case 18: BF_GO(1, 8); break;
        case 24: BF_GO(2, 4); break;
        case 28: BF_GO(2, 8); break;
        case 34: BF_GO(3, 4); break;
        case 44: BF_GO(4, 4); break;
        case 48: BF_GO(4, 8); break;
#define BF_GO_LDS(Q)                                                                                             \
    MM_LAUNCH(ctx, "bf_knn2_kernel", (bf_knn2_lds_kernel<Q>), grid, dim3(BF_THREADS), 0, q, nq, nq_cap, q_set_stride, t, \
              nt, nt_cap, t_set_stride, s, (uint32_t *)ws, idx, dist)
        case 114: BF_GO_LDS(1); break;
        case 115:
            MM_LAUNCH(ctx, "bf_knn2_kernel", (bf_knn2_lds_kernel<1, 512>), grid, dim3(BF_THREADS), 0, q, nq, nq_cap, q_set_stride,
                      t, nt, nt_cap, t_set_stride, s, (uint32_t *)ws, idx, dist);
            break;
        case 125:
            MM_LAUNCH(ctx, "bf_knn2_kernel", (bf_knn2_lds_kernel<2, 512>), grid, dim3(BF_THREADS), 0, q, nq, nq_cap, q_set_stride,
                      t, nt, nt_cap, t_set_stride, s, (uint32_t *)ws, idx, dist);
            break;
        case 124: BF_GO_LDS(2); break;
        case 134: BF_GO_LDS(3); break;
        case 144: BF_GO_LDS(4); break;
#undef BF_GO_LDS
        default: return mm_fail(ctx, MM_ERR_ARG, "mm_bf_knn2_batched: unknown MM_BF_VARIANT %d", var);
    }
#undef BF_GO
    if (s > 1) {
        dim3 g2((nq_cap + 255) / 256, n_pairs);
        MM_LAUNCH(ctx, "bf_merge_kernel", bf_merge_kernel, g2, dim3(256), 0, (const uint32_t *)ws, nq, nq_cap, s, idx, dist);
    }
    return MM_OK;
}

int mm_bf_knn2_hamming(mm_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist,
                       void *ws, size_t ws_bytes) {
    return mm_bf_knn2_batched(ctx, q, nullptr, nq, 0, t, nullptr, nt, 0, 1, idx, dist, ws, ws_bytes);
}

int mm_ratio_filter_batched(mm_ctx *ctx, const int32_t *idx, const int32_t *dist, const int32_t *nq, int nq_cap,
                            int n_pairs, double threshold, int32_t *pairs, int32_t *m_out) {
    if (!ctx) return MM_ERR_ARG;
    if (n_pairs == 0) return MM_OK;
    if (!idx || !dist || !pairs || !m_out || n_pairs < 0 || nq_cap < 0)
        return mm_fail(ctx, MM_ERR_ARG, "mm_ratio_filter_batched: bad argument");
    MM_LAUNCH(ctx, "ratio_filter_kernel", ratio_filter_kernel, dim3(n_pairs), dim3(256), 0, idx, dist, nq, nq_cap, threshold, pairs, m_out);
    return MM_OK;
}

}  // extern "C"
