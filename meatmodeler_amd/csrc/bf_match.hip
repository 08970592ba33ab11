// Brute-force Hamming 2-NN + Lowe ratio test for 256-bit descriptors (gfx950).
//
// Replaces cv2.FlannBasedMatcher(...).knnMatch(prev_desc, new_desc, k=2) and the ratio filter of
// processor.featureTracking (reference processor.py:132-137) by the exact search FLANN-LSH approximates.
//
// Mapping to the hardware.  The search is compute bound (0.02 B of HBM per descriptor pair), and
// there are four formulations of the distance in this file, all returning identical results (ties -> lowest train
// index); MM_BF_VARIANT selects one, the default is the fastest (times: tools/bench_bf.py, 499 pairs of 4000 x 4000):
//  314  matrix cores, FP4 operands, ONE candidate per lane and train tile (bf_knn2_fp4min_kernel: the vector unit only
//       takes a minimum over the lane's 16 accumulators; 4 waves per SIMD): 0.95 ms = 8.4 T pairs/s       <- default
//       (310: the same at 3 waves per SIMD with the A operand read one tile ahead: no faster)
//  300  matrix cores, FP4 operands, packed per-accumulator streams (bf_knn2_fp4_kernel): 1.15-1.2 ms = 6.6-6.9 T pairs/s
//  200  matrix cores, int8 operands (bf_knn2_mfma_kernel): 1.9 ms = 4.2 T pairs/s
//  114  xor / popcount on the vector unit (bf_knn2_lds_kernel, below): 3.9-4.2 ms = 2.0 T pairs/s; also what small
//       train sets (< 64) and train sets of 65536 or more descriptors take
// The matrix-core kernels are described where they are defined; the vector-unit formulation:
//  * one lane owns one query descriptor (8 VGPRs); a wave covers 64 queries, a workgroup 256;
//  * the workgroup stages 128 train descriptors at a time in LDS (double buffered, the next chunk's global loads fly
//    during the current chunk's compute); every lane reads the same train with two same-address ds_read_b128 (LDS
//    broadcast) so that the eight v_xor_b32 take VGPR operands (2.6 cycles per wave instruction; 4.1 with an SGPR
//    operand, measured: profiles/r01_valu_issue_rates.txt); 8 v_bcnt_u32_b32 accumulate the popcounts;
//  * trains are handled in groups of four: a candidate can only enter the two smallest keys (key = dist << 20 | train
//    index, ties -> lowest train index) if its distance is below the current second-best distance, so the group's
//    minimum is tested and the min / med3 bookkeeping sits behind a wave-uniform, rarely taken branch:
//    18.8 VALU instructions per descriptor pair, 16 of them the xor / popcount floor;
//  * no cross-lane traffic: the bound is VALU integer issue.  Measured with the SQ counters
//    (profiles/r02_bf_pmc.txt): 3.84 cycles per VALU instruction at the 2.1 GHz the chip holds under this load = 91 %
//    of the issue roof of this instruction mix (v_bcnt / v_min issue at quarter rate) -- which is why the distances
//    moved to the matrix cores;
//  * small launches split the train range over blockIdx.y and merge partial top-2 keys in a second kernel;
//  * the SGPR-fed variants (train descriptor by scalar loads, bf_knn2_kernel) are kept for MM_BF_VARIANT tuning runs.
#include "mm_common.h"
#include <cstdlib>
#include <type_traits>

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

// ---- Hamming distances on the matrix cores ------------------------------------------------------------------------------
// With the bits of a descriptor written as 256 int8 values +1 / -1, the dot product of two descriptors is
// (#equal bits) - (#different bits) = 256 - 2 dist: all-pairs matching is a [queries x 256] x [256 x trains] int8 GEMM, exact
// in the int32 accumulators of v_mfma_i32_32x32x32_i8 -- 0.25 cycles per descriptor pair and SIMD at the instruction's
// nominal rate, against the 19 VALU instructions per pair (1.2 cycles per pair and SIMD) of the xor / popcount formulation,
// which sits at 91 % of ITS roof (see the header).  (Measured: the int8 instruction issues every 29 ns per SIMD with the
// whole chip busy, 2.3 POP/s -- tools/dev/mfma_rate.hip -- and this kernel runs at that rate: 4.2 T pairs/s.)  The matching IS compute bound (0.02 B/pair), so this is where it belongs.
//   * a pre-pass expands every train set once ([nt, 256] int8, 8x the packed size, written to the workspace);
//   * a wave keeps 64 queries resident as the B operand (2 column tiles x 8 K-steps x 16 bytes per lane = 64 VGPRs,
//     expanded from the packed descriptors at kernel start); the workgroup (4 waves, 256 queries) streams train tiles of
//     32 descriptors through LDS (double buffered, 272-byte row pitch: conflict-free ds_read_b128) as the A operand;
//   * D[m][n]: lane l holds column (= query) l % 32 and 16 of the 32 rows (= trains), so the two-smallest search needs
//     no cross-lane traffic: accumulators are packed in pairs to 16 bits and turned into keys dist * 128 + tile number
//     with one v_perm_b32 + one v_pk_mad_u16 per two values, and every (lane, packed position) is its own stream with
//     its own two smallest keys (v_pk_max_u16 / v_pk_min_u16 x 2): 2.5 VALU instructions per descriptor pair.  The
//     train index of a stream entry is tile * 32 + the row of its accumulator; streams are folded into 32-bit keys
//     (dist << 16 | train index, ties -> lowest index as before) every 128 tiles and at the end, then the two lanes of a
//     query are merged.
typedef int bf_v4i __attribute__((ext_vector_type(4)));
typedef int bf_v16i __attribute__((ext_vector_type(16)));
constexpr int MF_TT = 32;        // train descriptors per tile
constexpr int MF_PITCH = 272;    // bytes per expanded descriptor row in LDS
constexpr int MF_SEG = 128;      // tiles per 7-bit tile number

__device__ __forceinline__ uint32_t bf_expand4(uint32_t nib) {      // 4 bits -> 4 bytes +1 / -1 (bit k -> byte k)
    const uint32_t x = (nib * 0x00204081u) & 0x01010101u;
    return ((x ^ 0x01010101u) * 0xFFu) | x;
}
__device__ __forceinline__ bf_v4i bf_expand16(uint32_t bits) {
    bf_v4i r;
    r[0] = (int)bf_expand4(bits & 15u);
    r[1] = (int)bf_expand4((bits >> 4) & 15u);
    r[2] = (int)bf_expand4((bits >> 8) & 15u);
    r[3] = (int)bf_expand4((bits >> 12) & 15u);
    return r;
}

// out [n_pairs][ntp][256] int8 (ntp = nt_cap rounded up to a tile); one thread per 16 bits -> 16 bytes
__global__ __launch_bounds__(256) void bf_expand_kernel(const uint8_t *__restrict__ t, int nt_cap, size_t t_stride, int ntp,
                                                        int n_pairs, uint8_t *__restrict__ out) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)n_pairs * ntp * 16;
    if (g >= total) return;
    const int chunk = (int)(g & 15);
    const size_t rowg = g >> 4;
    const int row = (int)(rowg % ntp), pair = (int)(rowg / ntp);
    uint32_t bits = 0;
    if (row < nt_cap) bits = reinterpret_cast<const uint16_t *>(t + (size_t)pair * t_stride + (size_t)row * 32)[chunk];
    reinterpret_cast<bf_v4i *>(out)[g] = bf_expand16(bits);
}

// (a builtin, not asm: this is the first reader of the MFMA results, and the compiler's hazard recogniser has to see
// the dependence to insert the wait states between a matrix instruction and a VALU read of its destination)
__device__ __forceinline__ uint32_t bf_pk_lo16(uint32_t hi_src, uint32_t lo_src) {   // {lo16(hi_src), lo16(lo_src)}
    return __builtin_amdgcn_perm(hi_src, lo_src, 0x05040100u);
}
// two 16-bit lanes per instruction (v_pk_mad_u16, v_pk_max_u16, v_pk_min_u16); vector types rather than asm so that the
// instruction scheduler knows them as VALU work it can place in the shadow of the matrix instructions
typedef unsigned short bf_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf_pk_mad(uint32_t a, uint32_t b, uint32_t c) {
    const bf_u16x2 r = __builtin_bit_cast(bf_u16x2, a) * __builtin_bit_cast(bf_u16x2, b) + __builtin_bit_cast(bf_u16x2, c);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ void bf_pk_top2(uint32_t key, uint32_t &b0, uint32_t &b1) {
    const bf_u16x2 k = __builtin_bit_cast(bf_u16x2, key), x0 = __builtin_bit_cast(bf_u16x2, b0), x1 = __builtin_bit_cast(bf_u16x2, b1);
    const bf_u16x2 hi = __builtin_elementwise_max(x0, k);
    b0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(x0, k));
    b1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(x1, hi));
}

__global__ __launch_bounds__(BF_THREADS) void bf_knn2_mfma_kernel(
    const uint8_t *__restrict__ q, const int32_t *__restrict__ nq_dev, int nq_cap, size_t q_stride,
    const uint8_t *__restrict__ tx, const int32_t *__restrict__ nt_dev, int nt_cap, int ntp,
    int32_t *__restrict__ idx, int32_t *__restrict__ dist) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[2][MF_TT][MF_PITCH];
    const int pair = blockIdx.z;
    const int nq = nq_dev ? min(nq_dev[pair], nq_cap) : nq_cap;
    const int nt = nt_dev ? min(nt_dev[pair], nt_cap) : nt_cap;
    const int qbase = blockIdx.x * BF_THREADS;
    if (qbase >= nq) return;  // workgroup-uniform
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    // resident operand: queries qbase + 64 wv + 32 u + c; this lane supplies the 16 K-values [32 j + 16 h, +16) of step j
    bf_v4i B[2][8];
    const uint8_t *qp = q + (size_t)pair * q_stride;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qi = qbase + 64 * wv + 32 * u + c;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (qi < nq) {
            const uint4 *p = reinterpret_cast<const uint4 *>(qp + (size_t)qi * 32);
            lo = p[0];
            hi = p[1];
        }
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) B[u][j] = bf_expand16((w[j] >> (16 * h)) & 0xFFFFu);
    }
    uint32_t s0[2][8], s1[2][8];      // per (query tile, packed accumulator pair): the two smallest 16-bit keys, two streams each
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int p = 0; p < 8; ++p) s0[u][p] = s1[u][p] = 0xFFFFFFFFu;
    uint32_t g0[2] = {BF_NONE, BF_NONE}, g1[2] = {BF_NONE, BF_NONE};   // folded: dist << 16 | train index
    const int ntiles = (nt + MF_TT - 1) / MF_TT;
    const uint8_t *txp = tx + (size_t)pair * ntp * 256;
    // staging: 512 chunks of 16 bytes per tile, two per thread
    const int e0 = threadIdx.x, e1 = threadIdx.x + 256;
    // Train tiles travel global -> registers -> LDS.  A tile is consumed in ~0.25 us per workgroup sharing the CU and a
    // load from L2 / HBM takes over a microsecond, so THREE tiles are in flight in registers (sets 0..2, used round robin:
    // the loop is unrolled by three so that the set index is static) on top of the two LDS buffers.
    bf_v4i a0, a1, b0_, b1_, c0_, c1_;      // register sets 0, 1, 2
    auto fetch = [&](int tt, auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
        if (tt < ntiles) {      // (workgroup-uniform)
            const bf_v4i *src = reinterpret_cast<const bf_v4i *>(txp + (size_t)tt * MF_TT * 256);
            const bf_v4i v0 = src[e0], v1 = src[e1];
            if constexpr (S == 0) { a0 = v0; a1 = v1; }
            if constexpr (S == 1) { b0_ = v0; b1_ = v1; }
            if constexpr (S == 2) { c0_ = v0; c1_ = v1; }
        }
    };
    auto commit = [&](int buf, auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
        bf_v4i *d0 = reinterpret_cast<bf_v4i *>(&tile[buf][e0 >> 4][(e0 & 15) * 16]);
        bf_v4i *d1 = reinterpret_cast<bf_v4i *>(&tile[buf][e1 >> 4][(e1 & 15) * 16]);
        if constexpr (S == 0) { *d0 = a0; *d1 = a1; }
        if constexpr (S == 1) { *d0 = b0_; *d1 = b1_; }
        if constexpr (S == 2) { *d0 = c0_; *d1 = c1_; }
    };
    auto fold = [&](int seg) __attribute__((always_inline)) {      // streams -> 32-bit keys; rows of packed pair p: 8 (p / 2) + 4 h + 2 (p % 2) and + 1
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int m = 8 * (p >> 1) + 4 * h + 2 * (p & 1);
#pragma unroll
                for (int x = 0; x < 2; ++x) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const uint32_t key16 = ((k == 0 ? s0[u][p] : s1[u][p]) >> (16 * x)) & 0xFFFFu;
                        if (key16 != 0xFFFFu) {
                            const uint32_t d = key16 >> 7, tno = key16 & 127u;
                            top2_insert((d << 16) | (uint32_t)((seg * MF_SEG + (int)tno) * MF_TT + m + x), g0[u], g1[u]);
                        }
                    }
                }
                s0[u][p] = s1[u][p] = 0xFFFFFFFFu;
            }
    };
    // one train tile: 16 MFMA, then 2.5 VALU instructions per descriptor pair.  PARTIAL (the last tile of a train set
    // whose size is no multiple of 32, peeled out of the loop): rows past the last train never win.
    auto tile_step = [&](int tt, auto partial_tag) __attribute__((always_inline)) {
        constexpr bool PARTIAL = decltype(partial_tag)::value;
        const int buf = tt & 1;
        bf_v16i acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
        bf_v4i A[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) A[j] = *reinterpret_cast<const bf_v4i *>(&tile[buf][c][32 * j + 16 * h]);
        // key = dist * 128 + tile number = 16384 - 64 s + tno in 16-bit arithmetic, two accumulators per instruction
        const uint32_t tno = (uint32_t)(tt & (MF_SEG - 1));
        const uint32_t base = (16384u + tno) * 0x00010001u;
        auto keys = [&](const bf_v16i &acc, int u, int p) __attribute__((always_inline)) {
            uint32_t k = bf_pk_mad(bf_pk_lo16((uint32_t)acc[2 * p + 1], (uint32_t)acc[2 * p]), 0xFFC0FFC0u, base);
            if constexpr (PARTIAL) {
                const int m = tt * MF_TT + 8 * (p >> 1) + 4 * h + 2 * (p & 1);
                k |= (m >= nt ? 0x0000FFFFu : 0u) | (m + 1 >= nt ? 0xFFFF0000u : 0u);
            }
            bf_pk_top2(k, s0[u][p], s1[u][p]);
        };
        // the first query tile's 8 MFMA, then the second tile's 8 with the first tile's bookkeeping in their shadow
        // (one matrix instruction, then five vector instructions: the scheduler is told to keep that pattern)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[j], B[0][j], acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {      // (scheduling barriers: keep one matrix instruction per slice of vector work)
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[j], B[1][j], acc1, 0, 0, 0);
            keys(acc0, 0, j);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) keys(acc1, 1, p);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    fetch(0, I0{});
    fetch(1, I1{});
    fetch(2, I2{});
    if (ntiles > 0) commit(0, I0{});
    fetch(3, I0{});
    __syncthreads();
    const int nfull = nt / MF_TT;      // complete tiles
    // iteration tt: tile tt is in LDS buffer tt & 1; tiles tt+1, tt+2, tt+3 are in the register sets (tt+1) % 3, (tt+2) % 3,
    // tt % 3.  After the compute: tile tt+1 goes to the other LDS buffer and its register set takes tile tt+4.
    auto iteration = [&](int tt, auto next_set) __attribute__((always_inline)) {
        tile_step(tt, std::false_type{});
        if ((tt & (MF_SEG - 1)) == MF_SEG - 1) fold(tt / MF_SEG);
        if (tt + 1 < ntiles) commit((tt & 1) ^ 1, next_set);
        fetch(tt + 4, next_set);
        __syncthreads();
    };
    int tt = 0;
    for (; tt + 3 <= nfull; tt += 3) {
        iteration(tt, I1{});
        iteration(tt + 1, I2{});
        iteration(tt + 2, I0{});
    }
    if (tt < nfull) {
        iteration(tt, I1{});
        if (tt + 1 < nfull) iteration(tt + 1, I2{});
    }
    if (nfull < ntiles) tile_step(nfull, std::true_type{});
    if (ntiles > 0) fold((ntiles - 1) / MF_SEG);
    // the two lanes of a query (rows 4 h ..) -> one result
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t o0 = __shfl_xor(g0[u], 32, 64), o1 = __shfl_xor(g1[u], 32, 64);
        top2_insert(o0, g0[u], g1[u]);
        top2_insert(o1, g0[u], g1[u]);
        const int qi = qbase + 64 * wv + 32 * u + c;
        if (h == 0 && qi < nq) {
            const size_t o = ((size_t)pair * nq_cap + qi) * 2;
            idx[o] = g0[u] == BF_NONE ? -1 : (int32_t)(g0[u] & 0xFFFFu);
            idx[o + 1] = g1[u] == BF_NONE ? -1 : (int32_t)(g1[u] & 0xFFFFu);
            dist[o] = g0[u] == BF_NONE ? -1 : (int32_t)(g0[u] >> 16);
            dist[o + 1] = g1[u] == BF_NONE ? -1 : (int32_t)(g1[u] >> 16);
        }
    }
}

// ---- the same on the FP4 path of the matrix cores ------------------------------------------------------------------------
// +1 and -1 are exact in FP4 (E2M1: 0x2 / 0xA), their products and sums up to 256 exact in the f32 accumulators, and
// v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 operands (block scales 1.0) is the fastest matrix instruction of the chip:
// measured 16 ns per instruction and SIMD with every SIMD busy against 29 ns for v_mfma_i32_32x32x32_i8 at half the K
// (tools/dev/mfma_rate.hip) -- 3.6x the int8 rate per descriptor pair.  A descriptor is 256 nibbles = 128 bytes (half the
// int8 form: half the LDS traffic too), a K-step takes 64 bits, a tile four matrix instructions per query tile.  The
// accumulators start at 1.5 * 2^23: the sum lands in the low mantissa bits as a two's-complement integer, so the
// 16-bit key arithmetic of the int8 kernel applies unchanged to the float's bit pattern.
typedef int bf_v8i __attribute__((ext_vector_type(8)));
typedef float bf_v16f __attribute__((ext_vector_type(16)));
constexpr int F4_PITCH = 144;    // bytes per expanded descriptor row in LDS (128 + 16: conflict-free ds_read_b128)

__device__ __forceinline__ uint32_t bf_fp4x8(uint32_t byte) {      // 8 bits -> 8 nibbles (+1: 0x2, -1: 0xA), bit k -> nibble k
    uint32_t x = byte;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return 0xAAAAAAAAu ^ (x << 3);
}
__device__ __forceinline__ bf_v4i bf_fp4x32(uint32_t bits) {
    bf_v4i r;
    r[0] = (int)bf_fp4x8(bits & 255u);
    r[1] = (int)bf_fp4x8((bits >> 8) & 255u);
    r[2] = (int)bf_fp4x8((bits >> 16) & 255u);
    r[3] = (int)bf_fp4x8(bits >> 24);
    return r;
}

// out [n_pairs][ntp][128] bytes; one thread per 32 bits -> 16 bytes
__global__ __launch_bounds__(256) void bf_expand_fp4_kernel(const uint8_t *__restrict__ t, int nt_cap, size_t t_stride, int ntp,
                                                            int n_pairs, uint8_t *__restrict__ out) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)n_pairs * ntp * 8;
    if (g >= total) return;
    const int chunk = (int)(g & 7);
    const size_t rowg = g >> 3;
    const int row = (int)(rowg % ntp), pair = (int)(rowg / ntp);
    uint32_t bits = 0;
    if (row < nt_cap) bits = reinterpret_cast<const uint32_t *>(t + (size_t)pair * t_stride + (size_t)row * 32)[chunk];
    reinterpret_cast<bf_v4i *>(out)[g] = bf_fp4x32(bits);
}

__global__ __launch_bounds__(BF_THREADS, 3) void bf_knn2_fp4_kernel(
    const uint8_t *__restrict__ q, const int32_t *__restrict__ nq_dev, int nq_cap, size_t q_stride,
    const uint8_t *__restrict__ tx, const int32_t *__restrict__ nt_dev, int nt_cap, int ntp,
    int32_t *__restrict__ idx, int32_t *__restrict__ dist) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[2][MF_TT][F4_PITCH];
    const int pair = blockIdx.z;
    const int nq = nq_dev ? min(nq_dev[pair], nq_cap) : nq_cap;
    const int nt = nt_dev ? min(nt_dev[pair], nt_cap) : nt_cap;
    const int qbase = blockIdx.x * BF_THREADS;
    if (qbase >= nq) return;  // workgroup-uniform
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    // resident operand: queries qbase + 64 wv + 32 u + c; this lane supplies the 32 K-values [64 j + 32 h, +32) of step j
    bf_v8i B[2][4];
    const uint8_t *qp = q + (size_t)pair * q_stride;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qi = qbase + 64 * wv + 32 * u + c;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (qi < nq) {
            const uint4 *p = reinterpret_cast<const uint4 *>(qp + (size_t)qi * 32);
            lo = p[0];
            hi = p[1];
        }
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf_v4i e = bf_fp4x32(~(h ? w[2 * j + 1] : w[2 * j]));      // queries enter NEGATED (see tile_step)
            B[u][j] = bf_v8i{e[0], e[1], e[2], e[3], 0, 0, 0, 0};
        }
    }
    uint32_t s0[2][8], s1[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int p = 0; p < 8; ++p) s0[u][p] = s1[u][p] = 0xFFFFFFFFu;
    uint32_t g0[2] = {BF_NONE, BF_NONE}, g1[2] = {BF_NONE, BF_NONE};
    const int ntiles = (nt + MF_TT - 1) / MF_TT;
    const uint8_t *txp = tx + (size_t)pair * ntp * 128;
    // staging: 256 chunks of 16 bytes per tile, one per thread; three tiles in flight in registers (see the int8 kernel)
    const int e0 = threadIdx.x;
    bf_v4i ra, rb, rc;
    auto fetch = [&](int tt, auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
        if (tt < ntiles) {      // (workgroup-uniform)
            const bf_v4i v = reinterpret_cast<const bf_v4i *>(txp + (size_t)tt * MF_TT * 128)[e0];
            if constexpr (S == 0) ra = v;
            if constexpr (S == 1) rb = v;
            if constexpr (S == 2) rc = v;
        }
    };
    auto commit = [&](int buf, auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
        bf_v4i *d = reinterpret_cast<bf_v4i *>(&tile[buf][e0 >> 3][(e0 & 7) * 16]);
        if constexpr (S == 0) *d = ra;
        if constexpr (S == 1) *d = rb;
        if constexpr (S == 2) *d = rc;
    };
    auto fold = [&](int seg) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int m = 8 * (p >> 1) + 4 * h + 2 * (p & 1);
#pragma unroll
                for (int x = 0; x < 2; ++x) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const uint32_t key16 = ((k == 0 ? s0[u][p] : s1[u][p]) >> (16 * x)) & 0xFFFFu;
                        if (key16 != 0xFFFFu) {
                            const uint32_t d = key16 >> 7, tno = key16 & 127u;
                            top2_insert((d << 16) | (uint32_t)((seg * MF_SEG + (int)tno) * MF_TT + m + x), g0[u], g1[u]);
                        }
                    }
                }
                s0[u][p] = s1[u][p] = 0xFFFFFFFFu;
            }
    };
    // One train tile = two phases of 4 matrix instructions; the vector work that goes with a phase's results runs in the
    // shadow of the NEXT phase's matrix instructions (4 MFMA ~ 136 cycles beside 8 key updates = 32 VALU instructions):
    //   phase 1 of tile t: MFMA of query tile 0  |  bookkeeping of query tile 1 of tile t-1 (carried across the barrier)
    //   phase 2 of tile t: MFMA of query tile 1  |  bookkeeping of query tile 0 of tile t
    // The matrix instruction delivers the 16-bit key itself: the queries are negated and the train operand carries the
    // block scale 2^6, so a tile adds -64 s to accumulators that start at 1.5 * 2^23 + 16384 + tile number -- the key
    // dist * 128 + tile number = 16384 - 64 s + tno then sits in the low 16 mantissa bits (all sums are integers below
    // 2^24: exact).  Left for the vector unit: pack two keys (v_perm_b32) and the two-smallest update, 2 instructions
    // per descriptor pair.
    auto keys = [&](const bf_v16f &acc, int u, int p, int dead_from) __attribute__((always_inline)) {
        const float f_hi = acc[2 * p + 1], f_lo = acc[2 * p];      // (values first: a bit cast of the element reference reads element 0)
        uint32_t k = bf_pk_lo16(__float_as_uint(f_hi), __float_as_uint(f_lo));
        if (dead_from >= 0) {      // (compile-time -1 in the loop) partial last tile: rows past the last train never win
            const int m = 8 * (p >> 1) + 4 * h + 2 * (p & 1);
            k |= (m >= dead_from ? 0x0000FFFFu : 0u) | (m + 1 >= dead_from ? 0xFFFF0000u : 0u);
        }
        bf_pk_top2(k, s0[u][p], s1[u][p]);
    };
    bf_v16f acc1_prev;      // query tile 1 of the previous train tile, bookkeeping pending
    bool pending = false;   // (workgroup-uniform)
    auto tile_step = [&](int tt, auto first_tag) __attribute__((always_inline)) {
        constexpr bool HAVE_PREV = !decltype(first_tag)::value;
        const int buf = tt & 1;
        const uint32_t tno = (uint32_t)(tt & (MF_SEG - 1));
        const float init = 12582912.0f + 16384.0f + (float)tno;
        bf_v16f acc0 = {init, init, init, init, init, init, init, init, init, init, init, init, init, init, init, init};
        bf_v16f acc1 = acc0;
        bf_v8i A[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf_v4i a = *reinterpret_cast<const bf_v4i *>(&tile[buf][c][32 * j + 16 * h]);
            A[j] = bf_v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0};
        }
        // cbsz = blgp = 4: both operands FP4; block scales (E8M0): train operand 133 = 2^6, queries 127 = 1.0
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[j], B[0][j], acc0, 4, 4, 0, 133, 0, 127);
            if constexpr (HAVE_PREV) {
                keys(acc1_prev, 1, 2 * j, -1);
                keys(acc1_prev, 1, 2 * j + 1, -1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (HAVE_PREV && ((tt - 1) & (MF_SEG - 1)) == MF_SEG - 1) fold((tt - 1) / MF_SEG);      // the previous tile closed a segment
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[j], B[1][j], acc1, 4, 4, 0, 133, 0, 127);
            keys(acc0, 0, 2 * j, -1);
            keys(acc0, 0, 2 * j + 1, -1);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc1_prev = acc1;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    fetch(0, I0{});
    fetch(1, I1{});
    fetch(2, I2{});
    if (ntiles > 0) commit(0, I0{});
    fetch(3, I0{});
    __syncthreads();
    const int nfull = nt / MF_TT;
    auto iteration = [&](int tt, auto next_set, auto first_tag) __attribute__((always_inline)) {
        tile_step(tt, first_tag);
        if (tt + 1 < ntiles) commit((tt & 1) ^ 1, next_set);
        fetch(tt + 4, next_set);
        __syncthreads();
    };
    int tt = 0;
    if (nfull > 0) {      // peeled: the first tile has no predecessor
        iteration(0, I1{}, std::true_type{});
        pending = true;
        tt = 1;
        if (tt < nfull) { iteration(tt, I2{}, std::false_type{}); ++tt; }
        if (tt < nfull) { iteration(tt, I0{}, std::false_type{}); ++tt; }
    }
    for (; tt + 3 <= nfull; tt += 3) {
        iteration(tt, I1{}, std::false_type{});
        iteration(tt + 1, I2{}, std::false_type{});
        iteration(tt + 2, I0{}, std::false_type{});
    }
    if (tt < nfull) {
        iteration(tt, I1{}, std::false_type{});
        if (tt + 1 < nfull) iteration(tt + 1, I2{}, std::false_type{});
    }
    if (pending) {      // bookkeeping of the last complete tile's second query tile
#pragma unroll
        for (int p = 0; p < 8; ++p) keys(acc1_prev, 1, p, -1);
        if (((nfull - 1) & (MF_SEG - 1)) == MF_SEG - 1) fold((nfull - 1) / MF_SEG);
    }
    if (nfull < ntiles) {      // the partial last tile, unpipelined
        const int buf = nfull & 1;
        const float init = 12582912.0f + 16384.0f + (float)(nfull & (MF_SEG - 1));
        bf_v16f acc0 = {init, init, init, init, init, init, init, init, init, init, init, init, init, init, init, init};
        bf_v16f acc1 = acc0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf_v4i a = *reinterpret_cast<const bf_v4i *>(&tile[buf][c][32 * j + 16 * h]);
            const bf_v8i A = bf_v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0};
            acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B[0][j], acc0, 4, 4, 0, 133, 0, 127);
            acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B[1][j], acc1, 4, 4, 0, 133, 0, 127);
        }
        const int dead_from = nt - nfull * MF_TT;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            keys(acc0, 0, p, dead_from);
            keys(acc1, 1, p, dead_from);
        }
    }
    if (ntiles > 0) fold((ntiles - 1) / MF_SEG);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const uint32_t o0 = __shfl_xor(g0[u], 32, 64), o1 = __shfl_xor(g1[u], 32, 64);
        top2_insert(o0, g0[u], g1[u]);
        top2_insert(o1, g0[u], g1[u]);
        const int qi = qbase + 64 * wv + 32 * u + c;
        if (h == 0 && qi < nq) {
            const size_t o = ((size_t)pair * nq_cap + qi) * 2;
            idx[o] = g0[u] == BF_NONE ? -1 : (int32_t)(g0[u] & 0xFFFFu);
            idx[o + 1] = g1[u] == BF_NONE ? -1 : (int32_t)(g1[u] & 0xFFFFu);
            dist[o] = g0[u] == BF_NONE ? -1 : (int32_t)(g0[u] >> 16);
            dist[o + 1] = g1[u] == BF_NONE ? -1 : (int32_t)(g1[u] >> 16);
        }
    }
}

// ---- FP4 matrix cores, one candidate per (lane, train tile): the bookkeeping cut from 2 to ~0.4 VALU per pair ------------
// In bf_knn2_fp4_kernel the vector unit is the limiter: pack + two-smallest update of every accumulator is 2 instructions
// per descriptor pair, 64 per wave and train tile, plus 32 v_mov to re-initialise the accumulators with the tile number:
// ~400 cycles beside ~270 cycles of matrix instructions, and the two do not overlap well.  Here the 16 accumulators of a
// lane (16 trains x 1 query) only feed a MINIMUM: the row of each accumulator rides in the low bits of the value itself
// (C operand = 1.5 * 2^23 + 2^18 + row, a per-lane constant in registers that is never rewritten; train block scale 2^10:
// acc = 1.5 * 2^23 + 2^11 * dist + row, exact), so the tile's best (dist, row) is an 8-instruction v_min3_u32 tree over
// the raw bit patterns, 3 instructions turn it into a key  dist << 22 | tile * 32 + row, 2 more (v_med3 / v_min) keep the
// lane's two smallest tile minima: 13 instructions per 16 pairs.  The exact two smallest of a query are then
//     best   = the smallest tile minimum (tile t0),
//     second = min(the smallest minimum of the other tiles, the SECOND smallest inside tile t0),
// and the second term is recomputed once per (lane, query) at the end of the kernel with xor / popcount on the 15 other
// rows of that one tile (packed descriptors from global memory): ~300 instructions per lane and query against ~6000 saved.
// Ties go to the lowest train index as everywhere (keys order by (dist, tile, row)).

// The 13 instructions per (query tile, train tile) in four chunks (5 + 3 + 3 + 2) so that each can be placed behind one
// matrix instruction of the OTHER query tile.
struct F4mState {
    uint32_t m[5], last, n0, n1, key;
};
template <bool PARTIAL>
__device__ __forceinline__ void f4m_chunk0(const bf_v16f &acc, F4mState &st, int h, int dead_from) {
    uint32_t v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float f = acc[i];      // (value first: a bit cast of the element reference reads element 0)
        v[i] = __float_as_uint(f);
        if constexpr (PARTIAL) {
            const int m = 8 * (i >> 2) + 4 * h + (i & 3);
            if (m >= dead_from) v[i] = 0xFFFFFFFFu;
        }
    }
    // minimum of 16 as v_min3_u32 x 8 (all values share sign and exponent: unsigned order = numeric order)
#pragma unroll
    for (int g = 0; g < 5; ++g) st.m[g] = min(min(v[3 * g], v[3 * g + 1]), v[3 * g + 2]);
    st.last = v[15];
}
__device__ __forceinline__ void f4m_chunk1(F4mState &st) {
    st.n0 = min(min(st.m[0], st.m[1]), st.m[2]);
    st.n1 = min(min(st.m[3], st.m[4]), st.last);
    st.n0 = min(st.n0, st.n1);
}
template <bool PARTIAL>
__device__ __forceinline__ void f4m_chunk2(F4mState &st, uint32_t tile32) {
    // key = dist << 22 | tile * 32 + row   (mask: mantissa bits 5..21 = dist << 11; bit 22 is the 1.5)
    const uint32_t m = st.n0;
    uint32_t k = ((m & 0x003FFFE0u) << 11) + tile32;
    k = (m & 31u) | k;
    if constexpr (PARTIAL) k = m == 0xFFFFFFFFu ? BF_NONE : k;
    st.key = k;
}
__device__ __forceinline__ void f4m_update(uint32_t k, uint32_t &b0, uint32_t &b1) {      // b0 <= b1
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(b1) : "v"(b0), "v"(b1), "v"(k));      // = min(max(b0, k), b1) for b0 <= b1
    b0 = min(b0, k);
}

#ifdef MM_BF_CLOCK
// diagnostic build only (tools/dev/bf_inkernel_clock.sh): shader clock the kernel ran at = s_memtime ticks per 100 MHz
// s_memrealtime tick around each workgroup's life; the product build executes no stamp
__device__ long long g_bf_clock[2 * 4096];
#endif
// PIPE: the A operand of the next train tile is read from LDS behind the last use of the current one (see below); without
// it a wave reads its operand after the barrier -- fewer registers: four waves per SIMD fit
template <int WAVES, bool PIPE>
__global__ __launch_bounds__(BF_THREADS, WAVES) void bf_knn2_fp4min_kernel(
    const uint8_t *__restrict__ q, const int32_t *__restrict__ nq_dev, int nq_cap, size_t q_stride,
    const uint8_t *__restrict__ tx, const uint8_t *__restrict__ tpk, size_t t_stride, const int32_t *__restrict__ nt_dev,
    int nt_cap, int ntp, int32_t *__restrict__ idx, int32_t *__restrict__ dist, int qtiles, int n_pairs) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[3][MF_TT][F4_PITCH];
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the linear id is read as
    // (slot, xcd) and an XCD walks ITS pairs one after the other, all query tiles of a pair side by side: the expanded
    // train set of a pair (512 KB at 4000 key points) is fetched into one L2 instead of into all eight.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = slot / qtiles;
    const int pair = grp * 8 + xcd;
    if (pair >= n_pairs) return;      // (padding of the last group of eight pairs)
    const int nq = nq_dev ? min(nq_dev[pair], nq_cap) : nq_cap;
    const int nt = nt_dev ? min(nt_dev[pair], nt_cap) : nt_cap;
    const int qbase = (slot - grp * qtiles) * BF_THREADS;
    if (qbase >= nq) return;  // workgroup-uniform
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
#ifdef MM_BF_CLOCK
    const long long clk_t0 = __builtin_readcyclecounter();
    const unsigned long long clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    bf_v4i B4[2][4];
    const uint8_t *qp = q + (size_t)pair * q_stride;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qi = qbase + 64 * wv + 32 * u + c;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (qi < nq) {
            const uint4 *p = reinterpret_cast<const uint4 *>(qp + (size_t)qi * 32);
            lo = p[0];
            hi = p[1];
        }
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf_v4i e = bf_fp4x32(~(h ? w[2 * j + 1] : w[2 * j]));      // queries enter NEGATED
            B4[u][j] = e;
        }
    }
    // C operand: 1.5 * 2^23 + 2^18 + row of accumulator i (rows 8 (i / 4) + 4 h + i % 4): never rewritten
    bf_v16f cinit;
#pragma unroll
    for (int i = 0; i < 16; ++i) cinit[i] = 12582912.0f + 262144.0f + (float)(8 * (i >> 2) + 4 * h + (i & 3));
    uint32_t b0[2] = {BF_NONE, BF_NONE}, b1[2] = {BF_NONE, BF_NONE};      // the two smallest tile minima per query
    const int ntiles = (nt + MF_TT - 1) / MF_TT;
    const uint8_t *txp = tx + (size_t)pair * ntp * 128;
    const int e0 = threadIdx.x;
    bf_v4i ra, rb, rc;
    auto fetch = [&](int tt, auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
        if (tt < ntiles) {      // (workgroup-uniform)
            const bf_v4i v = reinterpret_cast<const bf_v4i *>(txp + (size_t)tt * MF_TT * 128)[e0];
            if constexpr (S == 0) ra = v;
            if constexpr (S == 1) rb = v;
            if constexpr (S == 2) rc = v;
        }
    };
    auto commit = [&](int buf, auto set) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
        bf_v4i *d = reinterpret_cast<bf_v4i *>(&tile[buf][e0 >> 3][(e0 & 7) * 16]);
        if constexpr (S == 0) *d = ra;
        if constexpr (S == 1) *d = rb;
        if constexpr (S == 2) *d = rc;
    };
    // One train tile = two phases of 4 matrix instructions (query tile 0, query tile 1); the 13 vector instructions that
    // go with a phase's results run in the shadow of the NEXT phase's matrix instructions (query tile 1 of tile t-1 beside
    // query tile 0 of tile t: carried across the barrier).  cbsz = blgp = 4: FP4 operands; block scales (E8M0): train
    // operand 137 = 2^10, queries 127 = 1.0.
    // The A operand of tile t + 1 is read from LDS while tile t's matrix instructions run (three LDS buffers: tile t + 2 is
    // being committed meanwhile), so that a wave leaves the barrier with its operands in registers.
    bf_v16f acc1_prev;
    bf_v4i A4[4];      // (four registers per K-step; the instruction's operand type is eight wide, the upper half unused for FP4)
    auto read_a = [&](int buf, int j) __attribute__((always_inline)) {
        A4[j] = *reinterpret_cast<const bf_v4i *>(&tile[buf][c][32 * j + 16 * h]);
    };
#define F4M_A(j) (bf_v8i{A4[j][0], A4[j][1], A4[j][2], A4[j][3], 0, 0, 0, 0})
#define F4M_B(u, j) (bf_v8i{B4[u][j][0], B4[u][j][1], B4[u][j][2], B4[u][j][3], 0, 0, 0, 0})
    // A wave whose 64 queries all lie past the pair's last one (the fourth wave of the 16th query tile at 4000 key points)
    // only helps to stage the train tiles: no matrix instructions, no bookkeeping -- 1/64 of the matrix work at that size.
    const bool wave_active = __builtin_amdgcn_readfirstlane(qbase + 64 * wv) < nq;
    auto tile_step = [&](int tt, auto first_tag, int nxt, bool have_next) __attribute__((always_inline)) {
        constexpr bool HAVE_PREV = !decltype(first_tag)::value;
        if (!wave_active) return;
        if constexpr (!PIPE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) read_a((nxt + 2) % 3, j);
        }
        // query tile 0 of this train tile beside the bookkeeping of query tile 1 of the previous one, then query tile 1
        // beside the bookkeeping of query tile 0: one matrix instruction, one chunk of vector work, and so on
        F4mState st;
        bf_v16f acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F4M_A(0), F4M_B(0, 0), cinit, 4, 4, 0, 137, 0, 127);
        if constexpr (HAVE_PREV) f4m_chunk0<false>(acc1_prev, st, h, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F4M_A(1), F4M_B(0, 1), acc0, 4, 4, 0, 137, 0, 127);
        if constexpr (HAVE_PREV) f4m_chunk1(st);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F4M_A(2), F4M_B(0, 2), acc0, 4, 4, 0, 137, 0, 127);
        if constexpr (HAVE_PREV) f4m_chunk2<false>(st, (uint32_t)(tt - 1) * MF_TT);
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F4M_A(3), F4M_B(0, 3), acc0, 4, 4, 0, 137, 0, 127);
        if constexpr (HAVE_PREV) f4m_update(st.key, b0[1], b1[1]);
        __builtin_amdgcn_sched_barrier(0);
        bf_v16f acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F4M_A(0), F4M_B(1, 0), cinit, 4, 4, 0, 137, 0, 127);
        if (PIPE && have_next) read_a(nxt, 0);
        f4m_chunk0<false>(acc0, st, h, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F4M_A(1), F4M_B(1, 1), acc1, 4, 4, 0, 137, 0, 127);
        if (PIPE && have_next) read_a(nxt, 1);
        f4m_chunk1(st);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F4M_A(2), F4M_B(1, 2), acc1, 4, 4, 0, 137, 0, 127);
        if (PIPE && have_next) read_a(nxt, 2);
        f4m_chunk2<false>(st, (uint32_t)tt * MF_TT);
        __builtin_amdgcn_sched_barrier(0);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F4M_A(3), F4M_B(1, 3), acc1, 4, 4, 0, 137, 0, 127);
        if (PIPE && have_next) read_a(nxt, 3);
        f4m_update(st.key, b0[0], b1[0]);
        __builtin_amdgcn_sched_barrier(0);
        acc1_prev = acc1;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // tile t lives in register set / LDS buffer t % 3
    fetch(0, I0{});
    fetch(1, I1{});
    fetch(2, I2{});
    if (ntiles > 0) commit(0, I0{});
    fetch(3, I0{});
    if (ntiles > 1) commit(1, I1{});
    fetch(4, I1{});
    __syncthreads();
    if constexpr (PIPE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) read_a(0, j);
    }
    const int nfull = nt / MF_TT;
    // iteration tt (tt % 3 == CUR): A(tt) is in registers; read A(tt + 1), compute, commit tile tt + 2, request tile tt + 5
    auto iteration = [&](int tt, auto cur_tag, auto first_tag) __attribute__((always_inline)) {
        constexpr int CUR = decltype(cur_tag)::value, NXT = (CUR + 1) % 3, C2 = (CUR + 2) % 3;
        tile_step(tt, first_tag, NXT, tt + 1 < ntiles);
        if (tt + 2 < ntiles) commit(C2, std::integral_constant<int, C2>{});
        fetch(tt + 5, std::integral_constant<int, C2>{});
        __syncthreads();
    };
    int tt = 0;
    if (nfull > 0) {      // peeled: the first tile has no predecessor
        iteration(0, I0{}, std::true_type{});
        tt = 1;
        if (tt < nfull) { iteration(tt, I1{}, std::false_type{}); ++tt; }
        if (tt < nfull) { iteration(tt, I2{}, std::false_type{}); ++tt; }
    }
    for (; tt + 3 <= nfull; tt += 3) {      // (tt is a multiple of 3 here)
        iteration(tt, I0{}, std::false_type{});
        iteration(tt + 1, I1{}, std::false_type{});
        iteration(tt + 2, I2{}, std::false_type{});
    }
    if (tt < nfull) {
        iteration(tt, I0{}, std::false_type{});
        if (tt + 1 < nfull) iteration(tt + 1, I1{}, std::false_type{});
    }
    auto book = [&](const bf_v16f &acc, uint32_t tile32, int u, auto partial_tag, int dead_from) __attribute__((always_inline)) {
        constexpr bool PARTIAL = decltype(partial_tag)::value;
        F4mState st;
        f4m_chunk0<PARTIAL>(acc, st, h, dead_from);
        f4m_chunk1(st);
        f4m_chunk2<PARTIAL>(st, tile32);
        f4m_update(st.key, b0[u], b1[u]);
    };
    if (!wave_active) return;      // (no barrier follows)
    if (nfull > 0) book(acc1_prev, (uint32_t)(nfull - 1) * MF_TT, 1, std::false_type{}, 0);
    if (nfull < ntiles) {      // the partial last tile, unpipelined: rows past the last train never win
        bf_v16f acc0 = cinit, acc1 = cinit;      // (PIPE: the operands of tile nfull were read during the last iteration)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (!PIPE) read_a(nfull % 3, j);
            acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F4M_A(j), F4M_B(0, j), acc0, 4, 4, 0, 137, 0, 127);
            acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(F4M_A(j), F4M_B(1, j), acc1, 4, 4, 0, 137, 0, 127);
        }
        const int dead_from = nt - nfull * MF_TT;
        book(acc0, (uint32_t)nfull * MF_TT, 0, std::true_type{}, dead_from);
        book(acc1, (uint32_t)nfull * MF_TT, 1, std::true_type{}, dead_from);
    }
    // the second smallest INSIDE the best tile, by xor / popcount on the lane's 15 other rows of that tile
    const uint8_t *tp = tpk + (size_t)pair * t_stride;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qi = qbase + 64 * wv + 32 * u + c;
        {
            // (all loads unconditional on clamped rows and issued in four batches of four: a guarded load per row turns into
            // sixteen dependent trips to memory)
            const uint4 *p = reinterpret_cast<const uint4 *>(qp + (size_t)min(qi, nq - 1) * 32);
            const uint4 lo = p[0], hi = p[1];
            const uint32_t a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const bool have = b0[u] != BF_NONE;
            const uint32_t best_idx = have ? (b0[u] & 0xFFFFu) : 0u, t0 = best_idx & ~31u;
            const uint32_t last = (uint32_t)max(nt - 1, 0);
            uint32_t c2 = BF_NONE;
#pragma unroll
            for (int part = 0; part < 4; ++part) {
                uint4 r0[4], r1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ii = 4 * part + i;
                    const uint32_t ti = t0 + (uint32_t)(8 * (ii >> 2) + 4 * h + (ii & 3));
                    const uint4 *r = reinterpret_cast<const uint4 *>(tp + (size_t)min(ti, last) * 32);
                    r0[i] = r[0];
                    r1[i] = r[1];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ii = 4 * part + i;
                    const uint32_t ti = t0 + (uint32_t)(8 * (ii >> 2) + 4 * h + (ii & 3));
                    const uint32_t key = (ham256(a, r0[i], r1[i]) << 22) | ti;
                    c2 = (have && ti != best_idx && ti < (uint32_t)nt) ? min(c2, key) : c2;
                }
            }
            b1[u] = min(b1[u], c2);
        }
        // the two lanes of a query (rows 4 h ..) -> one result
        const uint32_t o0 = __shfl_xor(b0[u], 32, 64), o1 = __shfl_xor(b1[u], 32, 64);
        top2_insert(o0, b0[u], b1[u]);
        top2_insert(o1, b0[u], b1[u]);
        if (h == 0 && qi < nq) {
            const size_t o = ((size_t)pair * nq_cap + qi) * 2;
            idx[o] = b0[u] == BF_NONE ? -1 : (int32_t)(b0[u] & 0xFFFFu);
            idx[o + 1] = b1[u] == BF_NONE ? -1 : (int32_t)(b1[u] & 0xFFFFu);
            dist[o] = b0[u] == BF_NONE ? -1 : (int32_t)(b0[u] >> 22);
            dist[o + 1] = b1[u] == BF_NONE ? -1 : (int32_t)(b1[u] >> 22);
        }
    }
#ifdef MM_BF_CLOCK
    if (threadIdx.x == 0) {
        const unsigned w = blockIdx.x & 4095u;
        g_bf_clock[2 * w] = __builtin_readcyclecounter() - clk_t0;
        g_bf_clock[2 * w + 1] = (long long)(__builtin_amdgcn_s_memrealtime() - clk_r0);
    }
#endif
}

#undef F4M_B
#undef F4M_A
// queries per lane: 2 amortises the scalar train loads over two descriptor pairs; small launches use 1 to get more waves.
// MM_BF_VARIANT=<qpl><unroll> (e.g. 24, 28, 44, 14) overrides for tuning runs.
int bf_variant(int n_pairs, int nq_cap) {
    const char *e = getenv("MM_BF_VARIANT");
    const int forced = e ? atoi(e) : 0;
    if (forced) return forced;
    return 314;  // 314 / 310: matrix cores, FP4 operands, one candidate per lane and train tile (bf_knn2_fp4min_kernel, 4 / 3
                 // waves per SIMD); 300: FP4 with packed per-accumulator streams (bf_knn2_fp4_kernel); 200: int8; 114: the best
}                // xor / popcount variant (LDS-fed, one query per lane)

// the MFMA formulation needs train indices below 2^16 and pays off from a few train tiles on
bool bf_use_mfma(int n_pairs, int nq_cap, int nt_cap) {
    const int v = bf_variant(n_pairs, nq_cap);
    return (v == 200 || v == 300 || v == 310 || v == 314) && nt_cap >= 64 && nt_cap < 65536;
}
int bf_ntp(int nt_cap) { return (nt_cap + MF_TT - 1) / MF_TT * MF_TT; }

int bf_choose_splits(int n_pairs, int nq_cap, int nt_cap) {
    if (bf_use_mfma(n_pairs, nq_cap, nt_cap)) return 1;
    int var = bf_variant(n_pairs, nq_cap);
    if (var == 200 || var >= 300) var = 114;      // (shapes the matrix-core kernels do not take)
    const int BF_QTILE = BF_THREADS * ((var / 10) % 10);
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

#ifdef MM_BF_CLOCK
extern "C" int mm_debug_bf_clock(long long *host /*[2*4096]*/) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bf_clock), sizeof(g_bf_clock));
}
#endif

extern "C" {

size_t mm_bf_workspace_bytes(int n_pairs, int nq_cap, int nt_cap) {
    if (n_pairs > 0 && nq_cap > 0 && bf_use_mfma(n_pairs, nq_cap, nt_cap))
        return mm_align_up((size_t)n_pairs * bf_ntp(nt_cap) * (bf_variant(n_pairs, nq_cap) >= 300 ? 128 : 256), 256);   // expanded train sets
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
    if (bf_use_mfma(n_pairs, nq_cap, nt_cap)) {
        if (!ws || ws_bytes < mm_bf_workspace_bytes(n_pairs, nq_cap, nt_cap) || ((uintptr_t)ws & 15))
            return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_bf_knn2_batched: workspace too small or misaligned");
        const int ntp = bf_ntp(nt_cap);
        const dim3 grid((nq_cap + BF_THREADS - 1) / BF_THREADS, 1, n_pairs);
        if (bf_variant(n_pairs, nq_cap) >= 300) {
            const size_t chunks = (size_t)n_pairs * ntp * 8;
            MM_LAUNCH(ctx, "bf_expand_kernel", bf_expand_fp4_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, t, nt_cap,
                      t_set_stride, ntp, n_pairs, (uint8_t *)ws);
            const int qtiles = (int)grid.x;
            const dim3 grid_x(8u * (unsigned)((n_pairs + 7) / 8) * grid.x);      // (slot, xcd): see the kernel
            if (bf_variant(n_pairs, nq_cap) == 310) {
                MM_LAUNCH(ctx, "bf_knn2_fp4min_kernel", (bf_knn2_fp4min_kernel<3, true>), grid_x, dim3(BF_THREADS), 0, q, nq, nq_cap, q_set_stride,
                          (const uint8_t *)ws, t, t_set_stride, nt, nt_cap, ntp, idx, dist, qtiles, n_pairs);
                return MM_OK;
            }
            if (bf_variant(n_pairs, nq_cap) == 314) {
                MM_LAUNCH(ctx, "bf_knn2_fp4min_kernel", (bf_knn2_fp4min_kernel<4, false>), grid_x, dim3(BF_THREADS), 0, q, nq, nq_cap, q_set_stride,
                          (const uint8_t *)ws, t, t_set_stride, nt, nt_cap, ntp, idx, dist, qtiles, n_pairs);
                return MM_OK;
            }
            MM_LAUNCH(ctx, "bf_knn2_fp4_kernel", bf_knn2_fp4_kernel, grid, dim3(BF_THREADS), 0, q, nq, nq_cap, q_set_stride,
                      (const uint8_t *)ws, nt, nt_cap, ntp, idx, dist);
            return MM_OK;
        }
        const size_t chunks = (size_t)n_pairs * ntp * 16;
        MM_LAUNCH(ctx, "bf_expand_kernel", bf_expand_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, t, nt_cap,
                  t_set_stride, ntp, n_pairs, (uint8_t *)ws);
        MM_LAUNCH(ctx, "bf_knn2_mfma_kernel", bf_knn2_mfma_kernel, grid, dim3(BF_THREADS), 0, q, nq, nq_cap, q_set_stride,
                  (const uint8_t *)ws, nt, nt_cap, ntp, idx, dist);
        return MM_OK;
    }
    int var = bf_variant(n_pairs, nq_cap);
    if (var == 200 || var >= 300) var = 114;      // (shapes the matrix-core kernels do not take)
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
    MM_LAUNCH(ctx, "bf_knn2_lds_kernel", (bf_knn2_lds_kernel<Q>), grid, dim3(BF_THREADS), 0, q, nq, nq_cap, q_set_stride, t, \
              nt, nt_cap, t_set_stride, s, (uint32_t *)ws, idx, dist)
        case 114: BF_GO_LDS(1); break;
        case 115:
            MM_LAUNCH(ctx, "bf_knn2_lds_kernel", (bf_knn2_lds_kernel<1, 512>), grid, dim3(BF_THREADS), 0, q, nq, nq_cap, q_set_stride,
                      t, nt, nt_cap, t_set_stride, s, (uint32_t *)ws, idx, dist);
            break;
        case 125:
            MM_LAUNCH(ctx, "bf_knn2_lds_kernel", (bf_knn2_lds_kernel<2, 512>), grid, dim3(BF_THREADS), 0, q, nq, nq_cap, q_set_stride,
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
