// SPD solve by blocked Cholesky, f64, with an optional band (gfx950).
//
// This is the dense linear algebra of the trust-region step: the reduced camera system S (6F x 6F) that the
// reference hands to LSMR implicitly (scipy trf.py:480 via bundleAdjuster.py:180-192) is factored here.
// Tracks built from consecutive-keyframe matching only connect cameras at most `track length` apart, so S is block
// banded; `half_bandwidth` (A[i][j] == 0 for i - j > half_bandwidth) limits the work to the band: n*bw^2 instead of
// n^3/3 flops.  half_bandwidth >= n means dense.  Row-major A, lower triangle referenced / overwritten.
//
// Two execution schemes over the same 64 x 64 block kernels:
//   * narrow bands (<= 15 blocks): ONE data-flow scheduled launch factors the matrix and does the forward substitution
//     (chol_band_fused_kernel), one more does the backward substitution (chol_band_bwd_kernel) -- see the comments there;
//   * everything else: right-looking, three launches per block column
//       (1) chol_diag   : L_kk = chol(A_kk) and L_kk^-1
//       (2) chol_panel  : A_ik <- A_ik L_kk^-T          (64x64x64 f64-MFMA GEMM per row block, B operand = L_kk^-1)
//       (3) chol_update : A_ij <- A_ij - A_ik A_jk^T    (64x64x64 f64-MFMA GEMM per block pair i >= j > k)
//     and one launch per block column and direction for the substitutions (every workgroup recomputes the 64-vector
//     L_kk^-1 b_k instead of waiting for a separate launch).
// MFMA fragment maps for f64 16x16x4 (guide section 3): A lane l -> A[l&15][l>>4], B lane l -> B[l>>4][l&15],
// D reg i of lane l -> D[(l>>4) + 4 i][l&15].
#include "mm_common.h"
#include "chol_init.h"
#include <mutex>
#include <atomic>
#include <type_traits>
#include <cstdlib>

// The single-launch factorisation hands tiles from workgroup to workgroup INSIDE one launch with write-through (sc1)
// stores, agent-scope relaxed loads and words that are polled for a sentinel value instead of release / acquire fences.
// That is a property of the CDNA3 / CDNA4 memory system (stores tracked by vmcnt, sc1 stores written through to memory,
// agent-scope loads missing the other XCDs' L2), not of the HIP memory model: refuse to build for anything else, so that a
// new target shows up as a compile error and not as a rare wrong factor.  (MM_CHOL_FUSED=0 selects the launch-per-column
// factorisation, which needs none of this; tests/test_gpu_parity.py runs the parity cases on both.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "chol.hip: the cross-workgroup hand-over of chol_band_fused_kernel is written for gfx942 / gfx950"
#endif

namespace {

constexpr int NB = 64;
constexpr int LDT = 66;  // LDS tile leading dimension (doubles): 66 keeps the 32-lane ds_read_b64 groups conflict-free
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wave index inside the workgroup as a SCALAR (the compiler treats threadIdx.x >> 6 as divergent: every row / column
// offset derived from it would be per-lane arithmetic, and hoisted out of the loops, per-lane registers)
__device__ __forceinline__ int wave_id() {
    int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    asm volatile("" : "+s"(w));
    return w;
}
// Lane / thread index behind an opaque barrier: the fused kernel inlines every block operation into loops over block
// rows, and the optimiser hoists the (cheap) per-lane LDS and global offsets of ALL of them out of those loops -- several
// hundred live registers, spilled to scratch on the dependency chain.  An index the optimiser cannot see through keeps
// the address arithmetic next to its use.
__device__ __forceinline__ int lane_id() {
    int l = (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(l));
    return l;
}
__device__ __forceinline__ int thread_id() {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// ---- (1) diagonal block ------------------------------------------------------------------------------------------------
// Panel pb (columns c0 = 16 pb ..) of the 64 x 64 block in LDS.  update: wave w brings rows 16w..16w+15 of the panel
// up to date from the finished columns with f64 MFMA (K = c0).  factor (ONE wave): the 64 x 16 panel in registers,
// lane = row.  A lone wave is issue bound (an f64 instruction every 8 cycles), so the point is to minimise instructions
// per FMA: fully unrolled, static register indices, 1/sqrt by v_rsq_f64 + two Newton steps (one transcendental and ~8
// FMAs per pivot instead of a sqrt and a division); the reciprocals 1/L_jj are kept in R for the inverse.
template <int PB>
__device__ __forceinline__ void panel16_update(double (*M)[NB + 1]) {
    constexpr int c0 = PB * 16;
    const int lane = lane_id(), w = wave_id();
    if (16 * w + 15 >= c0) {  // rows above the panel hold final entries / structural zeros
        double4_t acc = {0, 0, 0, 0};
        const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < c0 / 4; ++ks) {
            const double av = M[16 * w + lr][4 * ks + lk];
            const double bv = M[c0 + lr][4 * ks + lk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) M[16 * w + (lane >> 4) + 4 * i][c0 + (lane & 15)] -= acc[i];
    }
    __syncthreads();
}

// ---- the panel itself: DPP row broadcasts ---------------------------------------------------------------------------------
// (Round 2 used 2 v_readlane_b32 + 1 v_fma_f64 with an SGPR operand per rank-1 update: see the history of this file.)
// f64 vector instructions issue every 8 cycles from a lone wave, a v_readlane_b32 every 4: 2 v_readlane + 1 v_fma per
// rank-1 update is 16 cycles, and so are TWO v_fmac_f64_dpp -- one for the lane's own row, one for a SHADOW copy of the
// 16 x 16 diagonal block that every 16-lane DPP row keeps (row t of the block in lane t of each DPP row), so that
// row_newbcast:j delivers L[c0 + j][q] to all 64 lanes without touching the scalar unit:
//     a[j]  -= bcast_j(slq) * lq        (own row)          sh[j] -= bcast_j(slq) * slq       (shadow row)
// That alone is a draw; the gain is what falls away around it: no SGPR traffic (no hazard nops, no pivot read-ahead), 13
// instead of 20 instructions of per-column overhead.  Measured (tools/dev/panel_insitu.hip): 1.65 us per 64 x 16 panel
// against 2.3.  (Tried: identity rows in the idle lanes above the panel, which the elimination turns into L_dd^-T for
// free -- 0.3 us for the extra selects plus 0.2 us for the masked LDS stores per panel; the separate 16 x 16 inversion
// below, with the same broadcasts, is 0.5 us and runs on another wave for three of the four panels.)
template <int J>
__device__ __forceinline__ void fmac_bcast(double &acc, double bsrc, double x) {   // acc -= bcast_J(bsrc) * x
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(x), "n"(J));
}
template <int J>
__device__ __forceinline__ double mov_bcast(double src) {   // (s_nop: a DPP read of a VGPR written by the previous VALU needs 2 wait states)
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(J));
    return r;
}
template <int Q, int J>
__device__ __forceinline__ void dpp_update_cols(double (&a)[16], double (&sh)[16], double slq, double lq) {
    if constexpr (J < 16) {
        fmac_bcast<J>(sh[J], slq, slq);
        fmac_bcast<J>(a[J], slq, lq);
        dpp_update_cols<Q, J + 1>(a, sh, slq, lq);
    }
}
// (Round 4, tried and dropped: issuing column Q + 1's pivot chain -- broadcast, v_rsq_f64, refinement, scaling: ten
// dependent instructions -- one by one BETWEEN the independent v_fmac_f64_dpp pairs of column Q's update, every instruction a
// volatile asm statement in hand-written order.  Bit-identical factors, and not a microsecond gained (0.679 ms per
// factor + solve at n = 3000 either way): the panel is bound by the NUMBER of f64 / DPP instructions a lone wave can issue
// (one per ~8 cycles), not by the latency of the pivot chain.  Likewise the third-order refinement below (5 instead of 9
// instructions per pivot) moved nothing measurable.  What would: half the FMAs, i.e. no shadow copy -- which f64 DPP,
// with row_newbcast as its only broadcast, does not offer.)
template <int Q>
__device__ __forceinline__ void dpp_column(double (&a)[16], double (&sh)[16], double &rvec, int t) {
    const double piv = mov_bcast<Q>(sh[Q]);
    // 1 / sqrt(piv): v_rsq_f64 (~2^-26) + ONE third-order step r (1 + e/2 + 3 e^2 / 8), e = 1 - piv r^2: 5 instructions on
    // the pivot's dependency chain instead of the 9 of two Newton steps (the wave issues one f64 instruction per 8 cycles,
    // 64 pivots per block: ~1 us per block column of the chain); the dropped term is 5/16 e^3 < 2^-75
    double r = __builtin_amdgcn_rsq(piv);
    {
        const double t = piv * r;
        const double e = fma(-t, r, 1.0);
        const double q = fma(0.375, e, 0.5);
        r = fma(r * e, q, r);
    }
    rvec = t == Q ? r : rvec;
    sh[Q] *= r;
    a[Q] *= r;
    asm volatile("s_nop 1" : "+v"(sh[Q]), "+v"(a[Q]));
    dpp_update_cols<Q, Q + 1>(a, sh, sh[Q], a[Q]);
}
// The LAST panel has no rows below its diagonal block: the block sits in DPP row 3, where the lane's own row IS the copy a
// broadcast needs -- no shadow, half the FMAs (120 instead of 240; round 4).  Lanes of the rows above hold zeros throughout.
template <int Q, int J>
__device__ __forceinline__ void dpp_update_cols_own(double (&a)[16], double lq) {
    if constexpr (J < 16) {
        fmac_bcast<J>(a[J], lq, lq);
        dpp_update_cols_own<Q, J + 1>(a, lq);
    }
}
template <int Q>
__device__ __forceinline__ void dpp_column_own(double (&a)[16], double &rvec, int t) {
    const double piv = mov_bcast<Q>(a[Q]);
    double r = __builtin_amdgcn_rsq(piv);
    {
        const double tt = piv * r;
        const double e = fma(-tt, r, 1.0);
        const double q = fma(0.375, e, 0.5);
        r = fma(r * e, q, r);
    }
    rvec = t == Q ? r : rvec;
    a[Q] *= r;
    asm volatile("s_nop 1" : "+v"(a[Q]));
    dpp_update_cols_own<Q, Q + 1>(a, a[Q]);
}
__device__ __forceinline__ void panel16_factor_dpp_last(double (*M)[NB + 1], double *R) {
    constexpr int c0 = 48;
    const int lane = lane_id(), t = lane & 15;
    double a[16], rvec = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const double v = M[lane][c0 + q];
        a[q] = lane >= c0 + q ? v : 0.0;
    }
    // (a pivot broadcast in the rows above reads 0 -> rsq = inf, 0 * inf = NaN in DEAD lanes only: they are written back
    // as zeros below, and rvec is taken from the lanes of row 3)
    dpp_column_own<0>(a, rvec, t);
    dpp_column_own<1>(a, rvec, t);
    dpp_column_own<2>(a, rvec, t);
    dpp_column_own<3>(a, rvec, t);
    dpp_column_own<4>(a, rvec, t);
    dpp_column_own<5>(a, rvec, t);
    dpp_column_own<6>(a, rvec, t);
    dpp_column_own<7>(a, rvec, t);
    dpp_column_own<8>(a, rvec, t);
    dpp_column_own<9>(a, rvec, t);
    dpp_column_own<10>(a, rvec, t);
    dpp_column_own<11>(a, rvec, t);
    dpp_column_own<12>(a, rvec, t);
    dpp_column_own<13>(a, rvec, t);
    dpp_column_own<14>(a, rvec, t);
    dpp_column_own<15>(a, rvec, t);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        M[lane][c0 + q] = lane >= c0 + q ? a[q] : 0.0;
    }
    if (lane >= 48) R[lane] = rvec;
}
// Panel PB of the 64 x 64 block in M (columns c0 = 16 PB ..), ONE wave; R receives the reciprocal pivots.
template <int PB>
__device__ __forceinline__ void panel16_factor_dpp(double (*M)[NB + 1], double (*X)[NB + 1], double *R) {
    if constexpr (PB == 3) {
        panel16_factor_dpp_last(M, R);
        return;
    }
    constexpr int c0 = PB * 16;
    const int lane = lane_id(), t = lane & 15;
    double a[16], sh[16], rvec = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const double v = M[lane][c0 + q], d = M[c0 + t][c0 + q];
        a[q] = lane >= c0 + q ? v : 0.0;
        sh[q] = t >= q ? d : 0.0;
    }
    dpp_column<0>(a, sh, rvec, t);
    dpp_column<1>(a, sh, rvec, t);
    dpp_column<2>(a, sh, rvec, t);
    dpp_column<3>(a, sh, rvec, t);
    dpp_column<4>(a, sh, rvec, t);
    dpp_column<5>(a, sh, rvec, t);
    dpp_column<6>(a, sh, rvec, t);
    dpp_column<7>(a, sh, rvec, t);
    dpp_column<8>(a, sh, rvec, t);
    dpp_column<9>(a, sh, rvec, t);
    dpp_column<10>(a, sh, rvec, t);
    dpp_column<11>(a, sh, rvec, t);
    dpp_column<12>(a, sh, rvec, t);
    dpp_column<13>(a, sh, rvec, t);
    dpp_column<14>(a, sh, rvec, t);
    dpp_column<15>(a, sh, rvec, t);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        M[lane][c0 + q] = lane >= c0 + q ? a[q] : 0.0;
    }
    if (lane < 16) R[c0 + lane] = rvec;
}

// first column (1-based, offset by k0) whose diagonal entry of L is not a positive number, 0 if none; call with one
// full wave after the factorisation
__device__ __forceinline__ int block_first_bad(const double (*M)[NB + 1], int k0) {
    const int lane = lane_id();
    const double d = M[lane][lane];
    const unsigned long long m = __ballot(!(d > 0.0 && d < 1.0e300));
    return m ? k0 + (int)__builtin_ctzll(m) + 1 : 0;
}

// L^-1 by 16 x 16 blocks, one block row per call (ONE wave): the diagonal block by forward substitution in registers
// (lane = column, 120 FMAs, reciprocal pivots from R), the blocks left of it on f64 MFMA:
//   X_ij = -X_ii * sum_{k=j}^{i-1} L_ik X_kj
// Block row i only needs L rows <= 16 i + 15 and X rows < 16 i, so it runs on an idle wave while wave 0 factors the
// next panel; only the last row is left for the end.
template <int K, int I>
__device__ __forceinline__ void inv16_rows(double (&x)[16], double lk, double xk) {   // x[i] -= L[i][K] x[K], i > K
    if constexpr (I < 16) {
        fmac_bcast<I>(x[I], lk, xk);      // lane i of the DPP row holds L[i][K] in lk
        inv16_rows<K, I + 1>(x, lk, xk);
    }
}
template <int K>
__device__ __forceinline__ void inv16_step(double (&x)[16], const double (&Lrow)[16], double Rv) {
    x[K] *= mov_bcast<K>(Rv);
    inv16_rows<K, K + 1>(x, Lrow[K], x[K]);
}
// Column `lane & 15` of the inverse of diagonal 16 x 16 block bi, right-looking forward substitution; the block sits in
// registers (lane t of every DPP row keeps row t) and L[i][K] reaches the FMAs as a row broadcast: 120 v_fmac_f64_dpp.
__device__ __forceinline__ void inv_diag16(const double (*M)[NB + 1], double (*X)[NB + 1], const double *R, int bi) {
    const int lane = lane_id(), t = lane & 15;
    const int o = 16 * bi;
    double Lrow[16], x[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) Lrow[q] = M[o + t][o + q];
    const double Rv = R[o + t];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (i == t) ? 1.0 : 0.0;
    inv16_step<0>(x, Lrow, Rv);
    inv16_step<1>(x, Lrow, Rv);
    inv16_step<2>(x, Lrow, Rv);
    inv16_step<3>(x, Lrow, Rv);
    inv16_step<4>(x, Lrow, Rv);
    inv16_step<5>(x, Lrow, Rv);
    inv16_step<6>(x, Lrow, Rv);
    inv16_step<7>(x, Lrow, Rv);
    inv16_step<8>(x, Lrow, Rv);
    inv16_step<9>(x, Lrow, Rv);
    inv16_step<10>(x, Lrow, Rv);
    inv16_step<11>(x, Lrow, Rv);
    inv16_step<12>(x, Lrow, Rv);
    inv16_step<13>(x, Lrow, Rv);
    inv16_step<14>(x, Lrow, Rv);
    inv16_step<15>(x, Lrow, Rv);
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) X[o + i][o + lane] = x[i];   // zero above the diagonal by construction
    }
    wave_lds_sync();
}

__device__ __forceinline__ void inv_offdiag16(const double (*M)[NB + 1], double (*X)[NB + 1], double (*Tw)[17], int bi,
                                              int bj) {
    const int lane = lane_id();
    const int lr = lane & 15, lk = lane >> 4;
    double4_t acc = {0, 0, 0, 0};
    for (int bk = bj; bk < bi; ++bk) {
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            const double av = M[16 * bi + lr][16 * bk + 4 * ss + lk];   // L_ik
            const double bv = X[16 * bk + 4 * ss + lk][16 * bj + lr];   // X_kj
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) Tw[lk + 4 * i][lr] = acc[i];
    wave_lds_sync();
    double4_t acc2 = {0, 0, 0, 0};
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
        const double av = X[16 * bi + lr][16 * bi + 4 * ss + lk];       // X_ii
        const double bv = Tw[4 * ss + lk][lr];
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc2, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) X[16 * bi + lk + 4 * i][16 * bj + lr] = -acc2[i];
    wave_lds_sync();
}

#ifdef MM_CHOL_TRACE
__device__ unsigned long long g_chol_trace[128][32];
#define MM_TRACE(r, e) do { if (side < 2 && threadIdx.x == 0 && (r) < 64) g_chol_trace[64 * side + (r)][e] = wall_clock64(); } while (0)
#define MM_TRACE_ROW(r, e) do { if (threadIdx.x == 0 && (r) >= 0 && (r) < 128) g_chol_trace[r][e] = wall_clock64(); } while (0)
#else
#define MM_TRACE(r, e) do { } while (0)
#define MM_TRACE_ROW(r, e) do { } while (0)
#endif

// Delay injection (causal profiling of the chain: tools/dev/chol_delay.sh builds one variant per site): ~1 us of s_sleep at
// ONE site; if the factorisation takes 28 x 1 us longer the site is on the critical path, if not it has slack.
#ifdef MM_CHOL_DELAY_SITE
#define MM_DELAY(site) do { if (MM_CHOL_DELAY_SITE == (site)) __builtin_amdgcn_s_sleep(40); } while (0)
#define MM_DELAY_IF(site, cond) do { if (MM_CHOL_DELAY_SITE == (site) && (cond)) __builtin_amdgcn_s_sleep(40); } while (0)
#else
#define MM_DELAY(site) do { } while (0)
#define MM_DELAY_IF(site, cond) do { } while (0)
#endif


// L (in place, lower triangle of M), the reciprocal pivots R and the four 16 x 16 diagonal blocks of X = L^-1 (the
// rest of X zeroed); 256 threads.  The diagonal inverses ride on an idle wave while wave 0 factors the next panel.
// `pub` streams the block out while it is being factored (the fused kernel): pub.l(k), called by wave 2 as soon as
// column panel k is final, hands over the 16 x 16 blocks of L below diagonal block k; pub.x(k), called by the wave
// that inverted diagonal block k, hands over X_kk.  Consumers work panel by panel behind the factorisation.
struct NoPub {
    __device__ void l(int) const {}
    __device__ void x(int) const {}
    __device__ void bulk(int) const {}
    __device__ void drain() const {}
    __device__ void raise_sub_flag() const {}
};
// Round 4: the factorisation is ONE wave's chain (wave 0: panel, its own rank-16k update of the next panel's live rows,
// panel, ...), the other three waves trail it -- diagonal inverses, streaming the finished blocks out -- released panel by
// panel through a counter in LDS instead of workgroup barriers.  The phase trace of round 3 showed every panel phase with
// helpers taking 2.6-2.9 us against 1.9 us for the first one, which wave 0 runs alone: six barriers per block made the
// chain wait for the helpers' global stores.  (The update of a 16-row tile is the same MFMA sequence whichever wave runs
// it: results unchanged bit for bit.)
template <int PB>
__device__ __forceinline__ void panel16_update_own(double (*M)[NB + 1]) {      // wave 0 alone; no barrier
    constexpr int c0 = PB * 16;
    const int lane = lane_id();
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int w = PB; w < 4; ++w) {
        double4_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < c0 / 4; ++ks) {
            const double av = M[16 * w + lr][4 * ks + lk];
            const double bv = M[c0 + lr][4 * ks + lk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) M[16 * w + (lane >> 4) + 4 * i][c0 + (lane & 15)] -= acc[i];
    }
    wave_lds_sync();
}
__device__ __forceinline__ void lds_counter_wait(volatile int *c, int target) {
    while (*c < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void lds_counter_set(volatile int *c, int value) {      // by one wave, after its LDS writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane_id() == 0) *c = value;
}
template <class Pub = NoPub>
__device__ __forceinline__ void factor_block_lds(double (*M)[NB + 1], double (*X)[NB + 1], double *R, int k0, int &bad,
                                                 Pub pub = Pub(), int trace_row = -1) {
    __shared__ int s_panels, s_xzero;      // panels finished by wave 0; helper waves that have cleared their share of X
    const int w = wave_id();
    if (thread_id() == 0) {
        s_panels = 0;
        s_xzero = 0;
    }
    __syncthreads();
    volatile int *panels = &s_panels, *xzero = &s_xzero;
    if (w == 0) {
        MM_DELAY(7);
        panel16_factor_dpp<0>(M, X, R);
        wave_lds_sync();
        lds_counter_set(panels, 1);
        MM_TRACE_ROW(trace_row, 10);
        MM_DELAY(15);
        panel16_update_own<1>(M);
        MM_TRACE_ROW(trace_row, 11);
        panel16_factor_dpp<1>(M, X, R);
        wave_lds_sync();
        lds_counter_set(panels, 2);
        MM_TRACE_ROW(trace_row, 12);
        panel16_update_own<2>(M);
        panel16_factor_dpp<2>(M, X, R);
        wave_lds_sync();
        lds_counter_set(panels, 3);
        MM_TRACE_ROW(trace_row, 13);
        MM_DELAY(16);
        panel16_update_own<3>(M);
        MM_TRACE_ROW(trace_row, 14);
        panel16_factor_dpp<3>(M, X, R);
        wave_lds_sync();
        lds_counter_wait(xzero, 3);      // (long true: X is cleared while the first panel is factored)
        MM_DELAY(1);
        inv_diag16(M, X, R, 3);          // the one inverse on the chain: half a microsecond
        pub.x(3);
    } else {
        for (int e = thread_id() - 64; e < NB * NB; e += 192) X[e / NB][e % NB] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane_id() == 0) atomicAdd(&s_xzero, 1);
        if (w == 1) {
            lds_counter_wait(xzero, 3);
            for (int k = 0; k < 3; ++k) {
                lds_counter_wait(panels, k + 1);
                inv_diag16(M, X, R, k);
                pub.x(k);
            }
        } else if (w == 2) {
            for (int k = 0; k < 3; ++k) {
                lds_counter_wait(panels, k + 1);
                MM_DELAY(5);
                pub.l(k);
            }
        } else {
            pub.bulk(0);      // (the copy of L_{r,r-1} it streams out was final before the factorisation started)
            pub.bulk(1);
            pub.bulk(2);
            pub.drain();
            pub.raise_sub_flag();
        }
    }
    __syncthreads();
    if (w == 1) bad = block_first_bad(M, k0);
    __syncthreads();
}

// the blocks of X below the diagonal, level by level (block (w + d, w) on wave w), after factor_block_lds
__device__ __forceinline__ void inverse_offdiag_lds(const double (*M)[NB + 1], double (*X)[NB + 1], double (*T)[16][17]) {
    const int w = wave_id();
#pragma unroll
    for (int d = 1; d < 4; ++d) {
        if (w + d < 4) inv_offdiag16(M, X, T[w], w + d, w);
        __syncthreads();
    }
}

// smallest failing column wins (several diagonal blocks may report in the fused kernel)
__device__ __forceinline__ void report_bad(int32_t *info, int bad) {
    int old = __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (old == 0 || bad < old) {
        const int prev = atomicCAS(info, old, bad);
        if (prev == old) break;
        old = prev;
    }
}

__global__ __launch_bounds__(256) void chol_diag_kernel(double *__restrict__ A, int n, int k0, double *__restrict__ Linv,
                                                        int32_t *__restrict__ info) {
    __shared__ double M[NB][NB + 1];
    __shared__ double X[NB][NB + 1];
    __shared__ double T[4][16][17];
    __shared__ double R[NB];
    const int lane = lane_id(), w = wave_id();
    const int nb = min(NB, n - k0);
    {   // wave w loads rows 16w..16w+15, lane = column: coalesced rows, 16 loads in flight per lane
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int r = 16 * w + q;
            v[q] = (r < nb && lane < nb && lane <= r) ? A[(size_t)(k0 + r) * n + k0 + lane] : 0.0;
            if (r >= nb && r == lane) v[q] = 1.0;  // identity padding of a partial last block
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) M[16 * w + q][lane] = v[q];
    }
    __syncthreads();
    int bad = 0;
    factor_block_lds(M, X, R, k0, bad);
    if (bad && threadIdx.x == 64) report_bad(info, bad);
    inverse_offdiag_lds(M, X, T);
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e / NB, c = e % NB;
        if (r < nb && c < nb && c <= r) A[(size_t)(k0 + r) * n + k0 + c] = M[r][c];
        Linv[e] = X[r][c];
    }
}

// 64x64 tile product  acc += As (64 x 64, rows) * Bs^T  on f64 MFMA; wave w owns the 32x32 quadrant (w>>1, w&1).
__device__ __forceinline__ void tile_gemm_nt(const double (*As)[LDT], const double (*Bs)[LDT], double4_t (&acc)[2][2]) {
    const int lane = lane_id(), w = wave_id();
    const int r0 = (w >> 1) * 32, c0 = (w & 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll 4
    for (int ks = 0; ks < NB / 4; ++ks) {
        const int k = ks * 4 + lk;
        double a0 = As[r0 + lr][k], a1 = As[r0 + 16 + lr][k];
        double b0 = Bs[c0 + lr][k], b1 = Bs[c0 + 16 + lr][k];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}

__device__ __forceinline__ void load_tile(double (*T)[LDT], const double *__restrict__ src, int ld, int rows, int cols) {
    // 64 x 64 doubles as 16-byte pieces, 8 per thread; all eight global loads are issued before the first LDS write
    // (a rolled loop with a guarded load per trip serialises eight memory latencies); zero fill outside (rows, cols)
    double2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = thread_id() + 256 * q;
        const int r = e / (NB / 2), c = (e % (NB / 2)) * 2;
        double2 t = make_double2(0.0, 0.0);
        if (r < rows) {
            if (c + 1 < cols) {
                t = *reinterpret_cast<const double2 *>(src + (size_t)r * ld + c);
            } else if (c < cols) {
                t.x = src[(size_t)r * ld + c];
            }
        }
        v[q] = t;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = thread_id() + 256 * q;
        const int r = e / (NB / 2), c = (e % (NB / 2)) * 2;
        T[r][c] = v[q].x;
        T[r][c + 1] = v[q].y;
    }
}

// ---- (2) panel: A_ik <- A_ik * Linv^T --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_panel_kernel(double *__restrict__ A, int n, int k0,
                                                         const double *__restrict__ Linv) {
    __shared__ double As[NB][LDT], Bs[NB][LDT];
    const int i0 = k0 + NB + blockIdx.x * NB;
    const int rows = min(NB, n - i0);
    const int kc = min(NB, n - k0);
    double *tile = A + (size_t)i0 * n + k0;
    load_tile(As, tile, n, rows, kc);
    load_tile(Bs, Linv, NB, NB, NB);
    __syncthreads();
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0, 0, 0, 0};
    tile_gemm_nt(As, Bs, acc);
    const int lane = lane_id(), w = wave_id();
    const int r0 = (w >> 1) * 32, c0 = (w & 1) * 32;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = r0 + a * 16 + (lane >> 4) + 4 * i, c = c0 + b * 16 + (lane & 15);
                if (r < rows && c < kc) tile[(size_t)r * n + c] = acc[a][b][i];
            }
}

// ---- (3) trailing update: A_ij -= A_ik A_jk^T for block pairs i >= j > k (inside the band) ------------------------------
__global__ __launch_bounds__(256) void chol_update_kernel(double *__restrict__ A, int n, int k0) {
    __shared__ double As[NB][LDT], Bs[NB][LDT];
    // decode the lower-triangular pair index
    const int t = blockIdx.x;
    int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while (bi * (bi + 1) / 2 > t) --bi;
    const int bj = t - bi * (bi + 1) / 2;
    const int i0 = k0 + NB + bi * NB, j0 = k0 + NB + bj * NB;
    const int rows = min(NB, n - i0), cols = min(NB, n - j0);
    const int kc = min(NB, n - k0);
    load_tile(As, A + (size_t)i0 * n + k0, n, rows, kc);
    load_tile(Bs, A + (size_t)j0 * n + k0, n, cols, kc);
    __syncthreads();
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0, 0, 0, 0};
    tile_gemm_nt(As, Bs, acc);
    const int lane = lane_id(), w = wave_id();
    const int r0 = (w >> 1) * 32, c0 = (w & 1) * 32;
    double *tile = A + (size_t)i0 * n + j0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = r0 + a * 16 + (lane >> 4) + 4 * i, c = c0 + b * 16 + (lane & 15);
                if (r < rows && c < cols && (bi != bj || c <= r)) tile[(size_t)r * n + c] -= acc[a][b][i];
            }
}

// ---- fused banded factorisation: ONE launch ----------------------------------------------------------------------------
// The launch-per-step scheme above pays three dependent kernels per block column (~37 us) although a column holds a
// few MFLOP.  For a narrow band the whole factorisation runs as one grid of resident workgroups that hand blocks to
// each other through flags in global memory (left-looking per block, data-flow scheduled):
//   * block (r, c) of L has ONE owner workgroup, which keeps  sum_k L_rk L_ck^T  in its MFMA accumulators while the
//     columns k = r - bwb .. c - 1 become available, then finishes the block (x L_cc^-T, or the diagonal factorisation)
//     and publishes it: no block is ever read-modified-written by two workgroups;
//   * the owner of the diagonal block (r, r) also owns (r, r - 1): the critical chain
//     L_cc^-1 -> L_{c+1,c} -> S_{c+1,c+1} -> L_{c+1,c+1}^-1 stays inside one workgroup (LDS, no flag hop);
//   * block (c + d, c) is alive for bwb - d + 1 columns, so that many workgroups per offset d (round robin over c) keep
//     every live block resident: (bwb + 1) + bwb (bwb - 1) / 2 workgroups per side (46 for bwb = 9).
// Every workgroup walks its blocks in increasing column order and waits only for blocks of smaller (column, offset)
// rank, so the unfinished block of smallest rank always has a running owner: no deadlock while all workgroups are
// resident (the host only takes this path for grids far below one workgroup per CU).  Spins are bounded anyway: a
// workgroup that gives up raises the abort flag, everybody leaves and the call reports MM_ERR_HIP.
//
// TWO-ENDED ("twisted") elimination.  The factorisation is a chain of nblk dependent block columns (~18 us each) and
// nothing else: 47 columns at n = 3000.  With both triangles of the band available the elimination runs from BOTH ends
// of the matrix at once: side 0 eliminates the block columns 0 .. a-1 downwards (T), side 1 the columns nblk-1 ..
// a+m downwards in its own, fully reversed coordinates (Bt: index i <-> 64 nblk - 1 - i, so "lower triangle" of side 1
// is the upper triangle of A and its factor lands there), and the m >= bwb middle columns (M), which separate the two
// ends, are eliminated last by side 0 with the contributions of both: chain a + m instead of nblk (28 instead of 47).
// This is the Cholesky factorisation of P A P^T for the permutation [T, reverse(Bt), M]; every entry of the factor is
// stored at the position of the entry of A it replaces.  A second set of workgroups plays side 1; it never waits for
// side 0, so the deadlock argument carries over.  n is padded to a multiple of 64 VIRTUALLY (identity rows that exist
// only in the index arithmetic): side 1 then shares the block grid of side 0 and its first block has `pad` leading
// identity rows -- the same trick as the partial last block of the one-ended scheme.
constexpr long SPIN_LIMIT = 1L << 23;
constexpr int FUSED_MAX_BWB = 15;  // 16 + 105 = 121 resident workgroups per side at most
constexpr size_t FUSED_LDS_BYTES = (size_t)(4 * NB * LDT + 4 * 16 * 17 + 3 * NB) * sizeof(double);   // 146 KB of the CU's 160

struct TwGeom {
    int n, nblk, bwb;
    int a, m, b;  // block columns eliminated by side 0 first (T), last (M), and by side 1 (Bt); b == 0: one-ended
    int pad;      // 64 nblk - n virtual identity rows behind the matrix
};

// Element (row, col) of a 64 x 64 tile in either coordinate system: p + row sr + col sc, valid inside [r_lo, r_hi) x
// [c_lo, c_hi) (outside: structural zero / identity padding, never dereferenced).
struct TileRef {
    double *p;
    int ld, sgn;   // element (row, col) sits sgn * (row * ld + col) doubles from p (32-bit offsets: the host checks 64 nblk n < 2^31)
    int r_lo, r_hi, c_lo, c_hi;
    __device__ __forceinline__ double *at(int row, int col) const { return p + (long)(sgn * (row * ld + col)); }
    __device__ __forceinline__ bool rv(int row) const { return row >= r_lo && row < r_hi; }
    __device__ __forceinline__ bool cv(int col) const { return col >= c_lo && col < c_hi; }
};

__device__ __forceinline__ TileRef tile_ref(double *A, const TwGeom &g, int side, int rb, int cb) {
    TileRef t;
    const long ld = g.n;
    if (side == 0) {
        t.p = A + (long)rb * NB * ld + (long)cb * NB;
        t.ld = g.n;
        t.sgn = 1;
        t.r_lo = 0;
        t.c_lo = 0;
        t.r_hi = max(0, min(NB, g.n - rb * NB));
        t.c_hi = max(0, min(NB, g.n - cb * NB));
    } else {
        const long N1 = (long)NB * g.nblk - 1;
        t.p = A + (N1 - (long)rb * NB) * ld + (N1 - (long)cb * NB);
        t.ld = g.n;
        t.sgn = -1;
        t.r_lo = max(0, g.pad - rb * NB);
        t.c_lo = max(0, g.pad - cb * NB);
        t.r_hi = NB;
        t.c_hi = NB;
    }
    return t;
}

// position of element i of block `blk` of a vector (right-hand side, y, x) in natural order; valid iff 0 <= . < n
__device__ __forceinline__ long vec_index(const TwGeom &g, int side, int blk, int i) {
    return side == 0 ? (long)NB * blk + i : (long)NB * g.nblk - 1 - (long)NB * blk - i;
}

// Coherence between the workgroups (they sit on different XCDs, each with its own L2): every shared block is written
// and read with agent-scope relaxed atomics (write-through stores, cache-bypassing loads: `global_* ... sc1`), the
// writer drains its stores (s_waitcnt) before raising the flag: no cache maintenance.  (Plain accesses bracketed by
// agent-scope release / acquire fences measured +4 % per factorisation.)
template <int MODE>
__device__ __forceinline__ double ld_shared(const double *p) {
    if (MODE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <int MODE>
__device__ __forceinline__ void st_shared(double *p, double v) {
    if (MODE == 2)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

// Two doubles in ONE 16-byte write-through store (p 16-byte aligned).  An 8-byte sc1 store is one fabric write per LANE:
// publishing a 64 x 16 panel that way took longer than factoring it (guide, visibility table: "scalar sc1 stores").
typedef double double2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_shared16(double *p, double v0, double v1) {
    const double2_t v = {v0, v1};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// 64 x 64 tile of a published block -> LDS (zero outside the valid range); 16 coalesced 8-byte loads per thread, all
// in flight
template <int MODE>
__device__ __forceinline__ void load_tile_shared(double (*T)[LDT], const TileRef &t) {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = thread_id() + 256 * q;
        const int r = e / NB, c = e % NB;
        v[q] = (t.rv(r) && t.cv(c)) ? ld_shared<MODE>(t.at(r, c)) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = thread_id() + 256 * q;
        T[e / NB][e % NB] = v[q];
    }
}

// LDS tile -> its place in A, 16-byte write-through stores (8 per thread); reversed coordinates: pairs the other way round
__device__ __forceinline__ void store_tile_shared16(const double (*T)[LDT], const TileRef &t) {
    const int tid = thread_id();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 256 * q, row = e >> 5, col = 2 * (e & 31);
        if (t.rv(row) && t.cv(col)) {
            if (t.sgn > 0)
                st_shared16(t.at(row, col), T[row][col], T[row][col + 1]);
            else
                st_shared16(t.at(row, col + 1), T[row][col + 1], T[row][col]);
        }
    }
}

// the same in two halves: request into registers (several tiles may be in flight), commit later (tile_commit)
template <int MODE>
__device__ __forceinline__ void tile_prefetch_shared(double (&pre)[16], const TileRef &t) {
    const int tid = thread_id();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * q;
        const int r = e / NB, c = e % NB;
        pre[q] = (t.rv(r) && t.cv(c)) ? ld_shared<MODE>(t.at(r, c)) : 0.0;
    }
}

__device__ __forceinline__ void tile_commit(double (*T)[LDT], const double (&pre)[16]) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = thread_id() + 256 * q;
        T[e / NB][e % NB] = pre[q];
    }
}

__device__ __forceinline__ bool spin_until_set(const int32_t *flag, int32_t *abort_flag) {
    for (long it = 0; it < SPIN_LIMIT; ++it) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
        if ((it & 255) == 255 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// all 256 threads; returns false (uniformly) when the grid is being abandoned
template <int MODE>
__device__ __forceinline__ bool wg_wait(const int32_t *f1, const int32_t *f2, int32_t *abort_flag, int *s_ok) {
    if (threadIdx.x == 0) *s_ok = spin_until_set(f1, abort_flag) && (f2 == nullptr || spin_until_set(f2, abort_flag));
    __syncthreads();
    const bool ok = *s_ok != 0;
    __syncthreads();
    if (MODE == 1)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop stale cache lines before reading the published block
    else
        asm volatile("" ::: "memory");
    return ok;
}

template <int MODE>
__device__ __forceinline__ void wg_publish(int32_t *flag) {
    if (MODE == 1)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's write-through stores have landed
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void zero_acc(double4_t (&acc)[2][2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0, 0, 0, 0};
}

// element (row, col) of accumulator register acc[a][b][i] (tile_gemm_nt layout)
#define MM_ACC_FOREACH(body)                                                                      \
    {                                                                                             \
        const int lane_ = lane_id(), w_ = wave_id();                                       \
        const int r0_ = (w_ >> 1) * 32, c0_ = (w_ & 1) * 32;                                       \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b) \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                       \
            const int row = r0_ + a * 16 + (lane_ >> 4) + 4 * i, col = c0_ + b * 16 + (lane_ & 15); \
            body                                                                                  \
        }                                                                                         \
    }

// out[row] (+)= sign * sum_k Tm[row][k] v[k] for a 64 x 64 tile in LDS (leading dimension LD); 256 threads, 4 per row
template <int LD>
__device__ __forceinline__ double tile_matvec(const double (*Tm)[LD], const double *v) {
    const int tid_ = thread_id(), r = tid_ >> 2, part = tid_ & 3;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += Tm[r][part * 16 + q] * v[part * 16 + q];
    s += __shfl_down(s, 2, 4);
    s += __shfl_down(s, 1, 4);
    return s;  // valid in the threads with (threadIdx.x & 3) == 0, row threadIdx.x >> 2
}

// finish an off-diagonal block: solve P L_cc^T = V (V = A0 - acc, staged in As), written to `tile` (global) and left
// in As.  The producer streams L_cc out while it factors it (factor_block_lds): after column panel k the 16 x 16 blocks
// L_jk (j > k) below diagonal block k, and X_kk = L_kk^-1 once an idle wave has inverted it.  The solve runs by
// 16-column blocks behind the producer,
//   stage k:  P_k = W_k X_kk^T,   W_j -= P_k L_jk^T  (j > k),
// wave w on rows 16w..16w+15 with wave-local LDS traffic and no workgroup barrier: each wave waits for the two stage
// counters itself and fetches the pieces it needs (all four waves store identical values into Bs / Xd).  When the last
// stage arrives, four MFMAs and a 2 KB load are all that is left; the owner of the diagonal block (r, r) also adds
// P_k P_k^T to its accumulator stage by stage, so the rank-64 update is off the chain as well.
template <int MODE>
__device__ __forceinline__ bool wave_wait_ge(const int32_t *flag, int want, int32_t *abort_flag) {
    for (long it = 0; it < SPIN_LIMIT; ++it) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) {
            if (MODE == 1)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            else
                asm volatile("" ::: "memory");
            return true;
        }
        if ((it & 255) == 255 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// s = sum_{k in [K0, K1)} P_k L_jk^T for this wave's 16 rows, subtracted from block j of As
template <int J, int K0, int K1>
__device__ __forceinline__ void trsm_update(double (*As)[LDT], const double (*Bs)[LDT], int row0, int lr, int lk) {
    if constexpr (K1 > K0) {
        double4_t s = {0, 0, 0, 0};
#pragma unroll
        for (int k = K0; k < K1; ++k)
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                const double av = As[row0 + lr][16 * k + 4 * ss + lk];   // P_k
                const double bv = Bs[16 * J + lr][16 * k + 4 * ss + lk];  // L_jk
                s = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, s, 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) As[row0 + lk + 4 * i][16 * J + lr] -= s[i];
        wave_lds_sync();
    }
}

// P_j = W_j X_jj^T: block j of As in place, and to global memory
template <int MODE, int J>
__device__ __forceinline__ void trsm_finish(double (*As)[LDT], const double (*Xd)[16][17], const TileRef &t, int row0,
                                            int lr, int lk, double *spub_blk) {
    double4_t pj = {0, 0, 0, 0};
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
        const double av = As[row0 + lr][16 * J + 4 * ss + lk];  // W = V_j - sum
        const double bv = Xd[J][lr][4 * ss + lk];               // X_jj^T
        pj = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, pj, 0, 0, 0);
    }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < 4; ++i) As[row0 + lk + 4 * i][16 * J + lr] = pj[i];
    wave_lds_sync();
    // to global memory from the LDS copy: two neighbouring columns per lane = one 16-byte write-through store (the
    // accumulator layout has them in neighbouring LANES).  Reversed coordinates (side 1): the pair sits the other way round.
    const int lane = lr + 16 * lk;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = lane + 64 * q, row = row0 + (e >> 3), col = 16 * J + 2 * (e & 7);
        const double v0 = As[row][col], v1 = As[row][col + 1];
        // Blocks that sit on the critical chain (block (r, r - 1) of a row head, the offset-2 block below it) go into the
        // polled hand-over buffer, in this side's own orientation, padding rows included: the next row head multiplies with
        // them panel by panel.  A write-through store takes microseconds to land and they queue behind each other, so
        // NOTHING else is stored here for such a block: its copy in A follows after the solve (store_tile_shared16).
        if (spub_blk) {
            st_shared16(spub_blk + row * NB + col, v0, v1);
        } else if (t.rv(row) && t.cv(col)) {
            if (t.sgn > 0)
                st_shared16(t.at(row, col), v0, v1);
            else
                st_shared16(t.at(row, col + 1), v1, v0);
        }
    }
}


// The streamed solve, stage by stage (this wave's 16 rows).
// No flag: the producer's pieces land in buffers that start out as a NaN sentinel (chol_init_kernel) -- X_kk in its place
// inside L_cc^-1, the blocks L_jk in the hand-over buffer `lpub` -- and every lane polls the very values it needs, so a
// hand-over costs one trip to memory instead of store-acknowledge + flag + load (the backward kernel's trick).
constexpr unsigned long long STAGE_SENTINEL = ~0ull;
constexpr int LPUB_BLOCK = 6 * 256;      // doubles per diagonal block: (1,0) (2,0) (3,0) | (2,1) (3,1) | (3,2), 16 x 16 row-major
__device__ __forceinline__ constexpr int lpub_first(int k) { return k == 0 ? 0 : (k == 1 ? 3 : 5); }
// doubles of the polled hand-over region (what chol_init fills with the sentinel): per block row the pieces of the diagonal
// block, the blocks (r, r - 1) and (r, r - 2), and -- per side -- the blocks of the offsets 3 .. bwb (round 4: every
// off-diagonal block reaches the owner of its right-hand neighbour panel by panel, not through a flag and its copy in A)
__host__ __device__ __forceinline__ size_t chol_nlpub(int nblk, int bwb) {
    return (size_t)nblk * (LPUB_BLOCK + (size_t)(2 + 2 * (bwb > 2 ? bwb - 2 : 0)) * NB * NB);
}

// Poll one word per lane (lanes may watch different words; `watch` = lanes that count) until none is the sentinel, with FOUR
// polls in flight: a poll that has to return before the next one is issued samples the word once per trip to memory
// (1-2 us under load), and half of that on average is pure detection latency on the critical chain.  Bounded; false: abandoned.
__device__ __forceinline__ bool poll_words_pipelined(const double *p, unsigned long long watch, int32_t *abort_flag) {
    auto ld = [&]() { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto there = [&](double v) {
        return !(__builtin_amdgcn_ballot_w64((unsigned long long)__double_as_longlong(v) == STAGE_SENTINEL) & watch);
    };
#ifndef MM_CHOL_POLL_DEPTH
#define MM_CHOL_POLL_DEPTH 4
#endif
    double c0 = ld();
    double c1 = 0.0, c2 = 0.0, c3 = 0.0;
    if (MM_CHOL_POLL_DEPTH >= 2) {
        __builtin_amdgcn_s_sleep(2);
        c1 = ld();
    }
    if (MM_CHOL_POLL_DEPTH >= 3) {
        __builtin_amdgcn_s_sleep(2);
        c2 = ld();
    }
    if (MM_CHOL_POLL_DEPTH >= 4) {
        __builtin_amdgcn_s_sleep(2);
        c3 = ld();
    }
    for (long it = 0; it < SPIN_LIMIT; ++it) {
        if (there(c0)) return true;
        c0 = ld();
        if (MM_CHOL_POLL_DEPTH >= 2) {
            if (there(c1)) return true;
            c1 = ld();
        }
        if (MM_CHOL_POLL_DEPTH >= 3) {
            if (there(c2)) return true;
            c2 = ld();
        }
        if (MM_CHOL_POLL_DEPTH >= 4) {
            if (there(c3)) return true;
            c3 = ld();
        }
        if ((it & 63) == 63 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    }
    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// registers of one stage: the blocks L_jK (j > K) this wave multiplies with, and X_KK
template <int K>
struct StageRegs {
    static constexpr int NL = K < 3 ? 4 * (3 - K) : 0;
    double lv[NL > 0 ? NL : 1], xv[4];
};
// one attempt: issue the loads of stage K (every lane the very values it will use)
template <int K>
__device__ __forceinline__ void stage_issue(StageRegs<K> &g, const double *lpub_c, const double *Linv_c, int lane) {
#pragma unroll
    for (int q = 0; q < StageRegs<K>::NL; ++q)
        g.lv[q] = __hip_atomic_load(lpub_c + 256 * lpub_first(K) + lane + 64 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q;
        g.xv[q] = __hip_atomic_load(Linv_c + (size_t)(16 * K + (e >> 4)) * NB + 16 * K + (e & 15), __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int K>
__device__ __forceinline__ bool stage_complete(const StageRegs<K> &g) {      // wave-uniform
    bool missing = false;
#pragma unroll
    for (int q = 0; q < StageRegs<K>::NL; ++q) missing |= (unsigned long long)__double_as_longlong(g.lv[q]) == STAGE_SENTINEL;
#pragma unroll
    for (int q = 0; q < 4; ++q) missing |= (unsigned long long)__double_as_longlong(g.xv[q]) == STAGE_SENTINEL;
    return !__builtin_amdgcn_ballot_w64(missing);
}
// Wait until stage K is there (bounded); false: abandoned.  Polling the whole stage (16 loads per lane, four waves per
// consumer, ten consumers per diagonal block) floods the memory queues the producer's stores travel through: while
// something is missing only TWO words are polled -- the last element each of the producer's two publishing waves
// stores for this stage (the corner of block (3, K), X_KK[15][15]) -- and the stage is loaded and checked once they
// are there (the check stays: stores of one wave may land in any order).
template <int K>
__device__ __forceinline__ bool stage_wait(StageRegs<K> &g, const double *lpub_c, const double *Linv_c, int lane,
                                           int32_t *abort_flag) {
    if (stage_complete<K>(g)) return true;
    // two words tell that the stage is (about to be) there: the last element each of the producer's two publishing waves
    // stores for it -- the corner of block (3, K) and X_KK[15][15]; even lanes watch the one, odd lanes the other
    const double *canary_x = Linv_c + (size_t)(16 * K + 15) * NB + 16 * K + 15;
    const double *canary_l = K < 3 ? lpub_c + 256 * (lpub_first(K) + (2 - K)) + 255 : canary_x;
    for (int attempt = 0; attempt < 64; ++attempt) {
        if (!poll_words_pipelined((lane & 1) ? canary_l : canary_x, 3ull, abort_flag)) return false;
        stage_issue<K>(g, lpub_c, Linv_c, lane);
        if (stage_complete<K>(g)) return true;      // (almost always: the check stays because stores of one wave may land in any order)
    }
    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}
// The row head's LAST stage sits on the chain  factorisation of L_{r-1,r-1} -> this solve -> factorisation of L_rr: it may
// poll the stage's own words (four per lane) instead of a canary followed by the load -- one trip
// to memory less per block column (only this one consumer per diagonal block: see stage_wait on flooding).
#ifndef MM_CHOL_DIRECT_LAST
#define MM_CHOL_DIRECT_LAST 2
#endif
template <int K>
__device__ __forceinline__ bool stage_wait_direct(StageRegs<K> &g, const double *lpub_c, const double *Linv_c, int lane,
                                                  int32_t *abort_flag) {
    // ONE attempt in flight: two (0.566 ms per factor + solve) or more flood the queues the producer's stores travel through
    // and are slower than this (0.560)
    for (long it = 0; it < SPIN_LIMIT; ++it) {
        if (stage_complete<K>(g)) return true;
        stage_issue<K>(g, lpub_c, Linv_c, lane);
        if ((it & 63) == 63 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    }
    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}
// stage K of the solve on this wave's 16 rows, its operands in registers
template <int MODE, int K>
__device__ __forceinline__ void stage_compute(const StageRegs<K> &g, double (*As)[LDT], double (*Bs)[LDT], double (*Xd)[16][17],
                                              const TileRef &t, double *spub_blk, int lane, int row0) {
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int q = 0; q < StageRegs<K>::NL; ++q) {
        const int e = lane + 64 * q;
        Bs[16 * (K + 1) + (e >> 4)][16 * K + (e & 15)] = g.lv[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q;
        Xd[K][e >> 4][e & 15] = g.xv[q];
    }
    wave_lds_sync();
    trsm_finish<MODE, K>(As, Xd, t, row0, lr, lk, spub_blk);
    if constexpr (K < 1) trsm_update<1, K, K + 1>(As, Bs, row0, lr, lk);
    if constexpr (K < 2) trsm_update<2, K, K + 1>(As, Bs, row0, lr, lk);
    if constexpr (K < 3) trsm_update<3, K, K + 1>(As, Bs, row0, lr, lk);
}

// acc += P_K P_K^T (columns 16K .. 16K+15 of As, all 64 rows: call after a workgroup barrier), tile_gemm_nt layout
template <int K>
__device__ __forceinline__ void syrk_slice(const double (*As)[LDT], double4_t (&acc)[2][2]) {
    const int lane = lane_id(), w = wave_id();
    const int r0 = (w >> 1) * 32, c0 = (w & 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int ks = 4 * K; ks < 4 * K + 4; ++ks) {
        const int k = ks * 4 + lk;
        double a0 = As[r0 + lr][k], a1 = As[r0 + 16 + lr][k];
        double b0 = As[c0 + lr][k], b1 = As[c0 + 16 + lr][k];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// The whole streamed solve of one off-diagonal block; V must be staged in As behind a workgroup barrier.  The loads of
// stage K + 1 are in flight while stage K computes (one attempt; whatever is still the sentinel is polled afterwards).
// SYRK (the owner of the diagonal block of the same row): also acc += P P^T -- the first three 16-column slices while the
// last stage is awaited, so that only 16 MFMAs of it follow the last stage.  spub_blk: see trsm_finish.  Returns false
// (uniformly) if a wait was abandoned.
template <int MODE, bool SYRK>
__device__ __forceinline__ bool finish_off_block_streamed(double (*As)[LDT], double (*Bs)[LDT], double (*Xd)[16][17],
                                                          const double *lpub_c, const double *Linv_c, int32_t *abort_flag,
                                                          const TileRef &t, double4_t (&acc)[2][2], double *spub_blk,
                                                          int trace_row = -1, int delay_tag = 0) {
    const int lane = lane_id(), row0 = 16 * wave_id();
    StageRegs<0> g0;
    StageRegs<1> g1;
    StageRegs<2> g2;
    StageRegs<3> g3;
    // (MM_CHOL_DIRECT_LAST: 1 = the row head polls the words of its LAST stage directly, 2 = of every stage -- its early
    // loads usually come back as the sentinel, and canary + reload is two trips to memory per stage on a consumer that delay
    // injection shows running BEHIND the producer's panels)
    auto wait_stage = [&](auto &gk, auto ktag) __attribute__((always_inline)) -> bool {
        constexpr int K = decltype(ktag)::value;
        if constexpr (SYRK && (MM_CHOL_DIRECT_LAST >= 2 || (MM_CHOL_DIRECT_LAST == 1 && K == 3)))
            return stage_wait_direct<K>(gk, lpub_c, Linv_c, lane, abort_flag);
        else
            return stage_wait<K>(gk, lpub_c, Linv_c, lane, abort_flag);
    };
    stage_issue<0>(g0, lpub_c, Linv_c, lane);
    bool ok = wait_stage(g0, std::integral_constant<int, 0>{});
    stage_issue<1>(g1, lpub_c, Linv_c, lane);
    if constexpr (SYRK) MM_DELAY(12);
    stage_compute<MODE, 0>(g0, As, Bs, Xd, t, spub_blk, lane, row0);
    ok = wait_stage(g1, std::integral_constant<int, 1>{}) && ok;
    stage_issue<2>(g2, lpub_c, Linv_c, lane);
    if constexpr (SYRK) MM_DELAY(13);
    stage_compute<MODE, 1>(g1, As, Bs, Xd, t, spub_blk, lane, row0);
    ok = wait_stage(g2, std::integral_constant<int, 2>{}) && ok;
    stage_issue<3>(g3, lpub_c, Linv_c, lane);
    if constexpr (SYRK) MM_DELAY(14);
    stage_compute<MODE, 2>(g2, As, Bs, Xd, t, spub_blk, lane, row0);
    if constexpr (SYRK) {
        __syncthreads();      // P_0 .. P_2 of all four waves
        syrk_slice<0>(As, acc);
        syrk_slice<1>(As, acc);
        syrk_slice<2>(As, acc);
    }
    ok = wait_stage(g3, std::integral_constant<int, 3>{}) && ok;
    MM_TRACE_ROW(trace_row, 2);
    if constexpr (SYRK) MM_DELAY(2);
    else MM_DELAY(4);
    MM_DELAY_IF(41, (delay_tag & 3) == 1);      // d = 2 owners
    MM_DELAY_IF(42, (delay_tag & 3) == 2);      // d >= 3 owners
    MM_DELAY_IF(43, (delay_tag & 3) && (delay_tag & 4));      // d >= 2, block of M x M
    MM_DELAY_IF(44, (delay_tag & 3) && !(delay_tag & 4));     // d >= 2, outside M x M
#ifdef MM_CHOL_DELAY_SITE
    MM_DELAY_IF(MM_CHOL_DELAY_SITE, MM_CHOL_DELAY_SITE >= 52 && MM_CHOL_DELAY_SITE < 70 && (delay_tag >> 4) == MM_CHOL_DELAY_SITE - 50);      // offset d = site - 50
#endif
    stage_compute<MODE, 3>(g3, As, Bs, Xd, t, spub_blk, lane, row0);
    MM_TRACE_ROW(trace_row, 3);
    const bool all_ok = !__syncthreads_or(!ok);
    if constexpr (SYRK) syrk_slice<3>(As, acc);
    return all_ok;
}

// Polled loads from the sub-diagonal hand-over buffer (block (r, r - 1) of row head r, written 16 columns at a time by
// trsm_finish): a whole 64 x 64 tile, or one 16-column panel of it, into an LDS tile; 256 threads, bounded.  Returns false
// (uniformly) if the wait was abandoned.
__device__ __forceinline__ bool load_tile_polled(double (*T)[LDT], const double *blk, int32_t *abort_flag) {
    const int tid = thread_id();
    double v[16];
    bool ok = false;
    for (long it = 0; it < SPIN_LIMIT; ++it) {
        bool missing = false;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            v[q] = __hip_atomic_load(blk + tid + 256 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            missing |= (unsigned long long)__double_as_longlong(v[q]) == STAGE_SENTINEL;
        }
        if (!__builtin_amdgcn_ballot_w64(missing)) {
            ok = true;
            break;
        }
        if ((it & 63) == 63 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = tid + 256 * q;
        T[e / NB][e % NB] = v[q];
    }
    return !__syncthreads_or(!ok);
}
// One 16-column panel of TWO blocks at once (blk_b may be null), pipelined: the loads of panel K + 1 are in flight while the
// products with panel K run.  issue: one attempt; wait: if something is still the sentinel, wave 0 polls the four (eight)
// words each writing wave stores LAST for this panel (row 16 w + 15, column 16 K + 15) while the others sit at a
// barrier -- polling whole panels with 256 threads floods the queues the producers' stores travel through -- then the
// panel is loaded again and checked (the check stays: stores of one wave may land in any order); commit: to LDS.
struct PanelPair {
    double va[4], vb[4];
};
template <int K>
__device__ __forceinline__ void panels_issue(PanelPair &g, const double *blk_a, const double *blk_b, int /*tid*/) {
    const int tid = thread_id();      // (opaque: keeps the offsets next to the loads, see lane_id())
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, off = (e >> 4) * NB + 16 * K + (e & 15);
        g.va[q] = __hip_atomic_load(blk_a + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g.vb[q] = blk_b ? __hip_atomic_load(blk_b + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
}
template <int K>
__device__ __forceinline__ bool panels_wait(PanelPair &g, const double *blk_a, const double *blk_b, int tid, int32_t *abort_flag,
                                            int trace_row = -1) {
#ifdef MM_CHOL_TRACE
    bool seen_a = false, seen_b = false;
#endif
    for (long round = 0; round < SPIN_LIMIT; ++round) {
#ifdef MM_CHOL_TRACE
        if (K == 3 && trace_row >= 0) {
            bool ma = false, mb = false;
            for (int q = 0; q < 4; ++q) {
                ma |= (unsigned long long)__double_as_longlong(g.va[q]) == STAGE_SENTINEL;
                mb |= (unsigned long long)__double_as_longlong(g.vb[q]) == STAGE_SENTINEL;
            }
            if (!__syncthreads_or(ma) && !seen_a) { seen_a = true; MM_TRACE_ROW(trace_row, 22); }
            if (!__syncthreads_or(mb) && !seen_b) { seen_b = true; MM_TRACE_ROW(trace_row, 23); }
        }
#endif
        bool missing = false;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            missing |= (unsigned long long)__double_as_longlong(g.va[q]) == STAGE_SENTINEL ||
                       (unsigned long long)__double_as_longlong(g.vb[q]) == STAGE_SENTINEL;
        if (!__syncthreads_or(missing)) return true;
        bool dead = false;
        if (wave_id() == 0) {
            const int lane = lane_id();
            const double *cp = ((lane & 4) && blk_b ? blk_b : blk_a) + (size_t)(16 * (lane & 3) + 15) * NB + 16 * K + 15;
            dead = !poll_words_pipelined(cp, blk_b ? 0xFFull : 0x0Full, abort_flag);
        }
        if (__syncthreads_or(dead)) break;
        panels_issue<K>(g, blk_a, blk_b, tid);
    }
    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}
template <int K>
__device__ __forceinline__ void panels_commit(const PanelPair &g, double (*Ta)[LDT], double (*Tb)[LDT], bool have_b, int /*tid*/) {
    const int tid = thread_id();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q;
        Ta[e >> 4][16 * K + (e & 15)] = g.va[q];
        if (have_b) Tb[e >> 4][16 * K + (e & 15)] = g.vb[q];
    }
    __syncthreads();
}
// acc += As[:, 16K .. 16K+15] * Bs[:, 16K .. 16K+15]^T (one 16-column slice of tile_gemm_nt)
template <int K>
__device__ __forceinline__ void gemm_slice(const double (*As)[LDT], const double (*Bs)[LDT], double4_t (&acc)[2][2]) {
    const int lane = lane_id(), w = wave_id();
    const int r0 = (w >> 1) * 32, c0 = (w & 1) * 32;
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int ks = 4 * K; ks < 4 * K + 4; ++ks) {
        const int k = ks * 4 + lk;
        double a0 = As[r0 + lr][k], a1 = As[r0 + 16 + lr][k];
        double b0 = Bs[c0 + lr][k], b1 = Bs[c0 + 16 + lr][k];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// producer side of the streamed hand-over (factor_block_lds calls l(k) on wave 2, x(k) on the wave that inverted
// diagonal block k): write-through stores and nothing else -- the consumers poll the values
template <int MODE>
struct StagePub {
    const double (*M)[NB + 1];
    const double (*X)[NB + 1];
    TileRef dt;
    double *Lr, *lpub;
    int32_t *sub_flag;      // flag of block (r, r - 1), raised once every wave's stores of it have landed (may be null)
    const double (*Ss)[LDT];   // L_{r,r-1} in LDS, st its place in A
    TileRef st;
    // wave 3 (idle throughout) beside the first three panels, a third each: the copy of L_{r,r-1} in A (what the
    // pre-accumulators and the later kernels read), 16-byte write-through stores, eleven per lane and panel
    __device__ __forceinline__ void bulk(int part) const {
        if (!sub_flag) return;
        const int lane = lane_id();
        for (int q = 11 * part; q < 11 * part + 11 && q < 32; ++q) {
            const int e = lane + 64 * q, row = e >> 5, col = 2 * (e & 31);
            if (st.rv(row) && st.cv(col)) {
                if (st.sgn > 0)
                    st_shared16(st.at(row, col), Ss[row][col], Ss[row][col + 1]);
                else
                    st_shared16(st.at(row, col + 1), Ss[row][col + 1], Ss[row][col]);
            }
        }
    }
    __device__ __forceinline__ void drain() const {      // every wave, two panels later: its stores are long acknowledged
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __device__ __forceinline__ void raise_sub_flag() const {      // one wave, one barrier after drain()
        if (sub_flag && lane_id() == 0) __hip_atomic_store(sub_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void l(int k) const {   // blocks (j, k), j > k: 16 (3 - k) rows x 16 columns, 16 bytes per lane
        const int lane = lane_id();
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int e = 2 * (lane + 64 * q), rr = 16 * (k + 1) + (e >> 4), cc = 16 * k + (e & 15);
            if (q < 2 * (3 - k)) st_shared16(lpub + 256 * lpub_first(k) + e, M[rr][cc], M[rr][cc + 1]);
        }
    }
    __device__ __forceinline__ void x(int k) const {
        const int lane = lane_id();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = 2 * (lane + 64 * q), rr = 16 * k + (e >> 4), cc = 16 * k + (e & 15);
            st_shared16(Lr + rr * NB + cc, X[rr][cc], X[rr][cc + 1]);
        }
    }
};

#define MM_FUSED_ABANDON                       \
    {                                          \
        if (threadIdx.x == 0) info[0] = -1;    \
        return;                                \
    }

// flags (int32): [0] abort, then per side: flag[nblk][W] block (r, d) published (d >= 1; the diagonal blocks are handed
// over piecewise through polled data, see trsm_stage) | yflag[nblk] | cflag[nblk][W] forward contribution of block (r, d)
// written | [nblk] unused
__device__ __forceinline__ size_t tw_side_flags(int nblk, int W) { return 2 * (size_t)nblk * W + 2 * (size_t)nblk; }

// Forward substitution L y = b rides along (b_fwd != nullptr): the owner of block (r, c) multiplies it with y_c as
// soon as that exists and hands the 64-vector to the owner of the diagonal block r, which adds the contributions in
// a fixed order (deterministic), applies L_rr^-1 and publishes y_r.  The y chain trails the factorisation by a hop or
// two.  NOT for free, as this comment claimed until round 4: y_c leaves its row head ~12 us after L_cc, and an owner that
// waits for it between two blocks is late for the next one -- with MM_CHOL_NO_FWD=1 the kernel took 506 instead of 531 us.
// The owners of the two farthest offsets therefore take the product after their next block (defer_fwd below); what is
// left of the difference is the tail of the y chain behind the last column.  (Two-ended: the rows of M receive
// contributions from both sides.)

template <int MODE>
__device__ __forceinline__ void chol_band_fused_body(double *A, TwGeom g, double *Linv,
                                                              int32_t *__restrict__ flags, int32_t *__restrict__ info,
                                                              const double *b_fwd, double *y,
                                                              double *contrib, double *lpub, double *spub,
                                                              const int32_t *slab_ready, int cams_per_slab, int n_cams, const unsigned bx) {
    // these waves form a latency chain; when the reduced system is being built by a concurrent launch they share their
    // SIMDs with its waves, so ask the instruction arbiter to prefer them
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ double smem[];
    double (*As)[LDT] = reinterpret_cast<double (*)[LDT]>(smem);
    double (*Bs)[LDT] = reinterpret_cast<double (*)[LDT]>(smem + NB * LDT);
    double (*M)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(smem);             // aliases As
    double (*X)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(smem + NB * LDT);  // aliases Bs
    double (*T)[16][17] = reinterpret_cast<double (*)[16][17]>(smem + 2 * NB * LDT);
    double *R = smem + 2 * NB * LDT + 4 * 16 * 17;
    double *ys = R + NB, *rhs = R + 2 * NB;
    // a row head parks its own two tiles of A here (fetched while it waits for the last column; 64 registers otherwise)
    double (*Ds)[LDT] = reinterpret_cast<double (*)[LDT]>(smem + 2 * NB * LDT + 4 * 16 * 17 + 3 * NB);
    double (*Ss)[LDT] = reinterpret_cast<double (*)[LDT]>(smem + 3 * NB * LDT + 4 * 16 * 17 + 3 * NB);
    __shared__ int s_ok;
    const int bwb = g.bwb, W = bwb + 1, nblk = g.nblk, n = g.n;
    const int G = W + bwb * (bwb - 1) / 2;  // workgroups per side
    const int side = (int)bx / G, lid = (int)bx % G;  // side 2: the pre-accumulators of M x M
    const bool tw = g.b > 0;
    const int nrows = side == 0 ? (tw ? g.a + g.m : nblk) : g.b + g.m;  // block rows this side touches
    const int ncols = side == 0 ? nrows : g.b;                         // block columns it eliminates
    int32_t *abort_flag = flags;
    int32_t *fb = flags + 1 + (size_t)(side & 1) * tw_side_flags(nblk, W);  // this side's flags
    auto fb0 = [&]() { return flags + 1; };
    auto fb1 = [&]() { return flags + 1 + tw_side_flags(nblk, W); };
    auto flag = [&](int32_t *base, int r, int d) { return base + (size_t)r * W + d; };
    auto yflag = [&](int32_t *base, int r) { return base + (size_t)nblk * W + r; };
    auto cflag = [&](int32_t *base, int r, int d) { return base + (size_t)nblk * W + nblk + (size_t)r * W + d; };
    auto pflag = [&](int i, int jj) { return flags + 1 + 2 * tw_side_flags(nblk, W) + (size_t)i * g.m + jj; };
    auto nat = [&](int s, int blk) { return s == 0 ? blk : nblk - 1 - blk; };  // block index in natural order
    auto cslot = [&](int s, int r, int d) { return contrib + (((size_t)s * nblk + r) * W + d) * NB; };
    auto linv = [&](int s, int blk) { return Linv + (size_t)nat(s, blk) * NB * NB; };
    auto lpubp = [&](int s, int blk) { return lpub + (size_t)nat(s, blk) * LPUB_BLOCK; };
    // polled copies of the blocks (blk, blk - 1) and (blk, blk - 2) of the rows that HAVE a row head on their side (blk <
    // ncols: natural indices are disjoint between the sides then; the rows of M in side 1's numbering have none, their
    // blocks go through A and flags only)
    auto spubp = [&](int s, int blk) { return spub + (size_t)nat(s, blk) * NB * NB; };
    auto spub2p = [&](int s, int blk) { return spub + ((size_t)nblk + nat(s, blk)) * NB * NB; };
    // block (r, r - dd), dd = 3 .. bwb, of side s: per side (the rows of M exist on both), in that side's orientation
    auto opubp = [&](int s, int r_, int dd) { return spub + (2 * (size_t)nblk + ((size_t)s * nblk + r_) * (bwb - 2) + (dd - 3)) * NB * NB; };
    // Entries of A (and of the right-hand side) may still be under construction by a concurrent launch on another
    // stream (the reduced camera system, built in camera slabs).  The entries (i, j) and (j, i), i >= j, are written
    // together with camera row i / 6: a tile is complete when the slabs of its LARGER natural block index are
    // (natural block q covers the cameras 64q/6 .. (64q+63)/6, i.e. at most two slabs).
    auto rows_ready = [&](int q) -> bool {
        if (!slab_ready) return true;
        const int s_lo = (NB * q / 6) / cams_per_slab, s_hi = min(n_cams - 1, (NB * q + NB - 1) / 6) / cams_per_slab;
        return wg_wait<MODE>(slab_ready + s_lo, s_hi != s_lo ? slab_ready + s_hi : nullptr, abort_flag, &s_ok);
    };
    double4_t acc[2][2];
    if (side == 2) {
        // Pre-accumulator of one block (r, c) of M x M.  What the columns of T and of the other end contribute to it
        // is known long before the elimination reaches M, but its owner on side 0 walks its blocks in column order and
        // would only start adding 2 bwb tile products when it gets there -- twice the work per block of the steady
        // state, paid on the chain.  This workgroup adds them up as the columns appear and replaces the block of A by
        // A_rc - sum; side 0 then eliminates M as a plain continuation of the band.
        int i = 0, q = lid;
        while (q > i) {
            q -= i + 1;
            ++i;
        }
        const int jj = q;  // block (i, jj) of M, jj <= i
        if (i >= g.m) return;
        const int r = g.a + i, c = g.a + jj;
        const TileRef t = tile_ref(A, g, 0, r, c);
        zero_acc(acc);
        // its own entries: requested now, parked in Ds before they are needed (not when the matrix is still being produced)
        double own[16];
        const bool early = !slab_ready;
        if (early) tile_prefetch_shared<MODE>(own, t);
        // the other end's blocks sit at their natural positions (upper triangle): read with side 0's addressing
        const int rr = nblk - 1 - r, rc = nblk - 1 - c;  // the two rows in side 1's numbering (rr <= rc)
        bool ok_ = true;
        auto product0 = [&](int k) __attribute__((always_inline)) {      // with column k of T
            if (r == c && k == r - 1) return;  // (the head of row r adds L_{r,r-1} L_{r,r-1}^T itself, from its LDS copy)
            if (!wg_wait<MODE>(flag(fb0(), r, r - k), r != c ? flag(fb0(), c, c - k) : nullptr, abort_flag, &s_ok)) {
                ok_ = false;
                return;
            }
            load_tile_shared<MODE>(As, tile_ref(A, g, 0, r, k));
            if (r != c) load_tile_shared<MODE>(Bs, tile_ref(A, g, 0, c, k));
            __syncthreads();
            if (r != c)
                tile_gemm_nt(As, Bs, acc);
            else
                tile_gemm_nt(As, As, acc);
        };
        auto product1 = [&](int kt) __attribute__((always_inline)) {     // with column kt of the other end
            if (!wg_wait<MODE>(flag(fb1(), rr, rr - kt), r != c ? flag(fb1(), rc, rc - kt) : nullptr, abort_flag, &s_ok)) {
                ok_ = false;
                return;
            }
            load_tile_shared<MODE>(As, tile_ref(A, g, 0, r, nblk - 1 - kt));
            if (r != c) load_tile_shared<MODE>(Bs, tile_ref(A, g, 0, c, nblk - 1 - kt));
            __syncthreads();
            if (r != c)
                tile_gemm_nt(As, Bs, acc);
            else
                tile_gemm_nt(As, As, acc);
        };
        // Every block of M x M takes a product with the LAST column of either end, and those two appear at about the same
        // time, last of all: everything older goes first (both ends), so that two products are left when they are there --
        // walking one end to its last column first left up to bwb products of the other end behind it, and the first rows
        // of M waited for them (the phase trace showed row a + 1 starting 12 us late).
        for (int k = max(0, r - bwb); k < g.a - 1 && ok_; ++k) product0(k);
        for (int kt = max(0, rc - bwb); kt < g.b - 1 && ok_; ++kt) product1(kt);
        if (ok_ && g.b - 1 >= max(0, rc - bwb)) product1(g.b - 1);
        if (ok_ && g.a - 1 >= max(0, r - bwb)) product0(g.a - 1);
        if (!ok_) MM_FUSED_ABANDON;
        if (!rows_ready(r)) MM_FUSED_ABANDON;
        // A_rc - sum goes back through LDS: 16-byte write-through stores (an 8-byte one is a fabric write per lane, and the
        // first rows of M wait for exactly this publication)
        if (early) tile_commit(Ds, own);
        __syncthreads();
        if (early) {
            MM_ACC_FOREACH(As[row][col] = Ds[row][col] - acc[a][b][i];)
        } else {
            MM_ACC_FOREACH(As[row][col] = ((t.rv(row) && t.cv(col)) ? ld_shared<MODE>(t.at(row, col)) : 0.0) - acc[a][b][i];)
        }
        __syncthreads();
        MM_DELAY(9);
        store_tile_shared16(As, t);
        wg_publish<MODE>(pflag(i, jj));
        return;
    }
    // role of this workgroup: offset d (0 = row head: blocks (r, r-1) and (r, r)), first column / row j, period
    int d = 0, j = lid, period = W;
    if (lid >= W) {
        int b = lid - W;
        d = 2;
        while (b >= bwb - d + 1) {
            b -= bwb - d + 1;
            ++d;
        }
        j = b;
        period = bwb - d + 1;
    }
    if (d >= 2) {
        // L_rc y_c for the forward substitution.  y_c leaves the head of row c at the END of its row, ~12 us after L_cc: an
        // owner that waits for it before it turns to its next block is late for that block's solve when its blocks follow
        // each other column by column (offset bwb: ONE workgroup, offset bwb - 1: two) -- and these are the blocks the whole
        // anti-diagonal of owners below waits for (measured by delay injection, tools/dev/chol_delay.sh: the row heads
        // have slack, the owners of the far offsets do not).  Those owners take the product of block (r, c) with y_c AFTER
        // their next block is out, from its copy in A; the head of row r needs it bwb - 1 columns later at the earliest.
#ifndef MM_CHOL_DEFER_MAXP
#define MM_CHOL_DEFER_MAXP 2
#endif
        const bool defer_fwd = period <= MM_CHOL_DEFER_MAXP && d >= period + 2;
        int pend_c = -1;
        auto fwd_part = [&](int rr, int cc, bool reload) -> bool {
            if (!wg_wait<MODE>(yflag(fb, cc), nullptr, abort_flag, &s_ok)) return false;
            if (reload) load_tile_shared<MODE>(As, tile_ref(A, g, side, rr, cc));      // (the barrier of wg_wait covers the last use of As)
            if (threadIdx.x < NB) {
                const long vi = vec_index(g, side, cc, threadIdx.x);
                ys[threadIdx.x] = (vi >= 0 && vi < n) ? ld_shared<MODE>(y + vi) : 0.0;
            }
            __syncthreads();
            const double tv = tile_matvec<LDT>(As, ys);
            // a row of M is finished by side 0: hand the vector over in ITS element order
            const int row = threadIdx.x >> 2, at = (side == 1 && rr >= ncols) ? NB - 1 - row : row;
            if ((threadIdx.x & 3) == 0) st_shared<MODE>(cslot(side, rr, d) + at, tv);
            wg_publish<MODE>(cflag(fb, rr, d));
            return true;
        };
        // The block's own entries of A: requested a whole block ahead (into registers, before the previous block's solve) and
        // parked in Ds, so that forming V = A_rc - sum costs LDS reads on the owners' chain, not a trip to memory.  Not
        // when the matrix is still being produced by a concurrent launch (slab gating), nor for the blocks of M x M, whose
        // entries come out of a pre-accumulator at the last moment.
        double own[16];
        auto own_early = [&](int c_) { return !slab_ready && !(tw && side == 0 && c_ >= g.a) && c_ + d < nrows && c_ < ncols; };
        if (own_early(j)) tile_prefetch_shared<MODE>(own, tile_ref(A, g, side, j + d, j));
        for (int c = j; c + d < nrows && c < ncols; c += period) {
            const int r = c + d;
            const TileRef t = tile_ref(A, g, side, r, c);
            zero_acc(acc);
            // a block of M x M: the columns of T and of the other end have been taken care of by its pre-accumulator
            const bool in_m = tw && side == 0 && c >= g.a;
            const bool early = own_early(c);
            for (int k = max(in_m ? g.a : 0, r - bwb); k < c; ++k) {
                if (k == c - 1) {
                    // The newest column.  Its two blocks -- (r, c - 1), solved by the owner of the next offset behind the
                    // factorisation of L_{c-1,c-1}, and L_{c,c-1} of row head c -- are taken 16 columns at a time from the
                    // polled hand-over buffers as they appear (what the row heads do with their last column): a flag, a
                    // drain of 32 KB of write-through stores and a tile load less on the chain  block (r, c - 1) -> block
                    // (r, c) -> ... -> row head r, which delay injection showed to pace the factorisation.
                    const double *pa_ = opubp(side, r, d + 1), *pb_ = spubp(side, c);
                    const int tid = thread_id();
                    __syncthreads();      // the previous column's products have read As / Bs
                    PanelPair p0, p1;
                    panels_issue<0>(p0, pa_, pb_, tid);
                    bool col_ok = panels_wait<0>(p0, pa_, pb_, tid, abort_flag);
                    panels_issue<1>(p1, pa_, pb_, tid);
                    panels_commit<0>(p0, As, Bs, true, tid);
                    gemm_slice<0>(As, Bs, acc);
                    col_ok = panels_wait<1>(p1, pa_, pb_, tid, abort_flag) && col_ok;
                    panels_issue<2>(p0, pa_, pb_, tid);
                    panels_commit<1>(p1, As, Bs, true, tid);
                    gemm_slice<1>(As, Bs, acc);
                    col_ok = panels_wait<2>(p0, pa_, pb_, tid, abort_flag) && col_ok;
                    panels_issue<3>(p1, pa_, pb_, tid);
                    panels_commit<2>(p0, As, Bs, true, tid);
                    gemm_slice<2>(As, Bs, acc);
                    col_ok = panels_wait<3>(p1, pa_, pb_, tid, abort_flag) && col_ok;
                    panels_commit<3>(p1, As, Bs, true, tid);
                    gemm_slice<3>(As, Bs, acc);
                    if (!col_ok) MM_FUSED_ABANDON;
                    continue;
                }
                if (!wg_wait<MODE>(flag(fb, r, r - k), c - k > 1 ? flag(fb, c, c - k) : nullptr, abort_flag, &s_ok)) MM_FUSED_ABANDON;
                load_tile_shared<MODE>(As, tile_ref(A, g, side, r, k));
                if (c - k > 1) {
                    load_tile_shared<MODE>(Bs, tile_ref(A, g, side, c, k));
                    __syncthreads();
                } else if (!load_tile_polled(Bs, spubp(side, c), abort_flag)) {
                    MM_FUSED_ABANDON;
                }
                tile_gemm_nt(As, Bs, acc);
            }
            // the block's own entries are fetched as late as possible (their latency hides behind the wait for L_cc):
            // when the matrix is still being produced by a concurrent launch, this owner needs them only now
            if (in_m) {
                if (!wg_wait<MODE>(pflag(r - g.a, c - g.a), nullptr, abort_flag, &s_ok)) MM_FUSED_ABANDON;
            } else if (!rows_ready(side == 0 ? r : nat(1, c))) {
                MM_FUSED_ABANDON;
            }
            if (early) tile_commit(Ds, own);
            __syncthreads();   // the last tile product has read As / Bs
            if (early) {
                MM_ACC_FOREACH(As[row][col] = Ds[row][col] - acc[a][b][i];)
            } else {
                MM_ACC_FOREACH(As[row][col] = ((t.rv(row) && t.cv(col)) ? ld_shared<MODE>(t.at(row, col)) : 0.0) - acc[a][b][i];)
            }
            __syncthreads();
            if (own_early(c + period)) tile_prefetch_shared<MODE>(own, tile_ref(A, g, side, c + period + d, c + period));
            // (d = 2: the block also goes out panel by panel for the head of its row, whose last column it is)
            MM_DELAY(8);      // (d >= 2 owner, products done, before its solve)
            if (!finish_off_block_streamed<MODE, false>(As, Bs, T, lpubp(side, c), linv(side, c), abort_flag, t, acc,
                                                        d == 2 ? (r < ncols ? spub2p(side, r) : nullptr) : opubp(side, r, d), -1,
                                                        (d == 2 ? 1 : 2) | (in_m ? 4 : 0) | (d << 4)))
                MM_FUSED_ABANDON;
            if (d > 2 || r < ncols) store_tile_shared16(As, t);      // (its copy in A: see trsm_finish)
            wg_publish<MODE>(flag(fb, r, d));
            MM_DELAY(11);      // (d >= 2 owner, block published)
            if (b_fwd) {
                if (pend_c >= 0) {      // the previous block's product: its y has long been published
                    if (!fwd_part(pend_c + d, pend_c, true)) MM_FUSED_ABANDON;
                    pend_c = -1;
                }
                if (defer_fwd)
                    pend_c = c;
                else if (!fwd_part(r, c, false))
                    MM_FUSED_ABANDON;
            }
        }
        if (pend_c >= 0 && !fwd_part(pend_c + d, pend_c, true)) MM_FUSED_ABANDON;
        return;
    }
    double4_t acc1[2][2];
    for (int r = j; r < nrows; r += period) {
        const bool diag_here = r < ncols;          // side 1 past its last column: only the block (b, b - 1) is left
        if (!diag_here && r != ncols) continue;
        const bool has_sub = r >= 1 && bwb >= 1;
        if (!diag_here && !has_sub) continue;
        const TileRef dt = tile_ref(A, g, side, r, r);
        const TileRef st = tile_ref(A, g, side, r, has_sub ? r - 1 : r);  // block (r, r - 1)
        MM_TRACE(r, 0);
        zero_acc(acc);
        zero_acc(acc1);
        // rows of M (two-ended, side 0): the diagonal block -- and the block left of it when that is in M too -- come
        // pre-accumulated over the columns of T and of the other end; only the columns of M are left to add
        const bool diag_pre = tw && side == 0 && r >= g.a, sub_pre = diag_pre && r - 1 >= g.a;
        // The row's own two tiles of A (Ds: diagonal tile, Ss: the tile left of it) are parked in LDS before the products
        // start -- nothing depends on them until the streamed solve, and two tile loads at the end of the row would sit on
        // the critical chain  L_{r-1,r-2} -> product with it -> streamed solve behind the factorisation of L_{r-1,r-1}.
        // (The first two rows of M wait for their pre-accumulators, which finish late: they fetch before the last column.)
        // The FIRST row of M is special: its block left of the diagonal, (a, a - 1), is an ordinary block of T, while its
        // diagonal block waits for a pre-accumulator that needs the other end's last column -- the latest thing in the whole
        // factorisation.  Its solve must not queue behind that wait (the pre-accumulators of column a wait for L_{a,a-1} in
        // turn, and the next rows for them): the diagonal tile is fetched after the streamed solve, where it is first used.
        const bool defer_diag = diag_pre && r == g.a && has_sub;
        bool fetched = false, fetch_ok = true;
        auto fetch_own = [&]() __attribute__((always_inline)) {
            fetched = true;
            if (diag_pre && !defer_diag &&
                !wg_wait<MODE>(pflag(r - g.a, r - g.a), sub_pre ? pflag(r - g.a, r - 1 - g.a) : nullptr, abort_flag, &s_ok))
                fetch_ok = false;
            if (fetch_ok && diag_here && !rows_ready(nat(side, r))) fetch_ok = false;
            if (fetch_ok && side == 1 && has_sub && !rows_ready(nat(1, r - 1))) fetch_ok = false;
            if (!fetch_ok) return;
            if (diag_here && !defer_diag) load_tile_shared<MODE>(Ds, dt);      // (the upper triangle is masked where the tile is used)
            if (has_sub) load_tile_shared<MODE>(Ss, st);
        };
        if (!diag_pre || r >= g.a + 2 || defer_diag) {
            fetch_own();
            if (!fetch_ok) MM_FUSED_ABANDON;
        }
        for (int k = max(0, r - bwb); k + 1 < r; ++k) {
            const bool do_diag = diag_here && !(diag_pre && k < g.a), do_sub = !(sub_pre && k < g.a);
            if (k + 2 == r && !fetched && !diag_here) {      // (the streamed last column below fetches AFTER its products)
                fetch_own();
                if (!fetch_ok) MM_FUSED_ABANDON;
            }
            if (!do_diag && !do_sub) continue;
            if (k + 2 == r && diag_here) {
                // The last column.  Both blocks of it -- the row's own L_{r,r-2} (owner: offset 2) and L_{r-1,r-2} (the row
                // head above) -- are being solved for right now, 16 columns at a time behind the factorisation of
                // L_{r-2,r-2}: their products are taken panel by panel as the panels appear in the polled buffers, and
                // this workgroup is ready for ITS streamed solve two microseconds after that factorisation ends.
                const double *own = spub2p(side, r), *above = do_sub ? spubp(side, r - 1) : nullptr;
                __syncthreads();      // the previous column's products have read As / Bs
                MM_TRACE(r, 8);
                MM_DELAY(6);
                // (one straight-line body per combination of the two products: accumulators updated under run-time
                // conditions end up shuffled between register files and scratch)
                bool col_ok = true;
                auto last_column = [&](auto diag_tag, auto sub_tag) __attribute__((always_inline)) {
                    constexpr bool DIAG = decltype(diag_tag)::value, SUB = decltype(sub_tag)::value;
                    const int tid = thread_id();
                    PanelPair p0, p1;
                    panels_issue<0>(p0, own, above, tid);
                    col_ok = panels_wait<0>(p0, own, above, tid, abort_flag);
                    MM_TRACE(r, 15);
                    panels_issue<1>(p1, own, above, tid);
                    panels_commit<0>(p0, As, Bs, SUB, tid);
                    if constexpr (DIAG) gemm_slice<0>(As, As, acc);
                    if constexpr (SUB) gemm_slice<0>(As, Bs, acc1);
                    col_ok = panels_wait<1>(p1, own, above, tid, abort_flag) && col_ok;
                    panels_issue<2>(p0, own, above, tid);
                    panels_commit<1>(p1, As, Bs, SUB, tid);
                    if constexpr (DIAG) gemm_slice<1>(As, As, acc);
                    if constexpr (SUB) gemm_slice<1>(As, Bs, acc1);
                    col_ok = panels_wait<2>(p0, own, above, tid, abort_flag) && col_ok;
                    panels_issue<3>(p1, own, above, tid);
                    panels_commit<2>(p0, As, Bs, SUB, tid);
                    if constexpr (DIAG) gemm_slice<2>(As, As, acc);
                    if constexpr (SUB) gemm_slice<2>(As, Bs, acc1);
                    col_ok = panels_wait<3>(p1, own, above, tid, abort_flag, side == 0 ? r : -1) && col_ok;
                    MM_TRACE(r, 5);
                    panels_commit<3>(p1, As, Bs, SUB, tid);
                    if constexpr (DIAG) gemm_slice<3>(As, As, acc);
                    if constexpr (SUB) gemm_slice<3>(As, Bs, acc1);
                };
                if (do_diag && do_sub)
                    last_column(std::true_type{}, std::true_type{});
                else if (do_diag)
                    last_column(std::true_type{}, std::false_type{});
                else
                    last_column(std::false_type{}, std::true_type{});
                if (!col_ok) MM_FUSED_ABANDON;
                if (!fetched) {      // the first two rows of M: their pre-accumulators finish last of all
                    fetch_own();
                    if (!fetch_ok) MM_FUSED_ABANDON;
                }
                continue;
            }
            // both blocks of the column are waited for and fetched together (one trip to memory, one barrier)
            if (!wg_wait<MODE>(flag(fb, r, r - k), do_sub ? flag(fb, r - 1, r - 1 - k) : nullptr, abort_flag, &s_ok)) MM_FUSED_ABANDON;
            {
                double pa[16], pb[16];
                tile_prefetch_shared<MODE>(pa, tile_ref(A, g, side, r, k));
                if (do_sub) tile_prefetch_shared<MODE>(pb, tile_ref(A, g, side, r - 1, k));
                tile_commit(As, pa);
                if (do_sub) tile_commit(Bs, pb);
            }
            __syncthreads();
            if (do_diag) tile_gemm_nt(As, As, acc);
            if (do_sub) tile_gemm_nt(As, Bs, acc1);
        }
        if (!fetched) {
            fetch_own();
            if (!fetch_ok) MM_FUSED_ABANDON;
        }
        if (has_sub) {
            MM_TRACE(r, 1);
            __syncthreads();   // the last tile product has read As / Bs
            // The solve runs IN the parked tile Ss (V = A_{r,r-1} - sum, in place): L_{r,r-1} then outlives the diagonal
            // block's factorisation, which takes As / Bs -- it is copied to A beside that factorisation by the idle waves
            // (StagePub::bulk) and multiplies y_{r-1} at the end of the row without being read back.
            MM_DELAY(17);
            MM_ACC_FOREACH(Ss[row][col] -= acc1[a][b][i];)
            __syncthreads();
            // the solve streams behind the factorisation of L_{r-1,r-1}; acc += L_{r,r-1} L_{r,r-1}^T rides along
            const bool ok = diag_here ? finish_off_block_streamed<MODE, true>(Ss, Bs, T, lpubp(side, r - 1), linv(side, r - 1), abort_flag,
                                                                              st, acc, spubp(side, r), side == 0 ? r : -1)
                                      : finish_off_block_streamed<MODE, false>(Ss, Bs, T, lpubp(side, r - 1), linv(side, r - 1),
                                                                               abort_flag, st, acc, nullptr);
            if (!ok) MM_FUSED_ABANDON;
            MM_TRACE(r, 4);
            // The block is out panel by panel in the polled buffer for those who sit on the chain (the next row head, the
            // owners of the blocks below); its copy in A and the FLAG that goes with it -- for the pre-accumulators --
            // follow beside the factorisation below.  One exception: the first row of M.  Its block (a, a - 1) is the last
            // thing the pre-accumulators of the column (., a) wait for, and the next row heads wait for THEM: published at
            // once (a few microseconds on the chain, once per factorisation).
            if (diag_pre && r == g.a) {
                store_tile_shared16(Ss, st);
                wg_publish<MODE>(flag(fb, r, 1));
            }
        }
        if (!diag_here) {  // side 1, block (b, b - 1): its row belongs to M, side 0 finishes it
            wg_publish<MODE>(flag(fb, r, 1));
            if (b_fwd) {
                if (!wg_wait<MODE>(yflag(fb, r - 1), nullptr, abort_flag, &s_ok)) MM_FUSED_ABANDON;
                if (threadIdx.x < NB) {
                    const long vi = vec_index(g, side, r - 1, threadIdx.x);
                    ys[threadIdx.x] = (vi >= 0 && vi < n) ? ld_shared<MODE>(y + vi) : 0.0;
                }
                __syncthreads();
                const double tv = tile_matvec<LDT>(Ss, ys);
                if ((threadIdx.x & 3) == 0) st_shared<MODE>(cslot(side, r, 1) + (NB - 1 - (threadIdx.x >> 2)), tv);
                wg_publish<MODE>(cflag(fb, r, 1));
            }
            __syncthreads();
            continue;
        }
        if (defer_diag) {      // (see above: the first row of M takes its pre-accumulated diagonal tile only now)
            if (!wg_wait<MODE>(pflag(0, 0), nullptr, abort_flag, &s_ok)) MM_FUSED_ABANDON;
            load_tile_shared<MODE>(Ds, dt);
        }
        __syncthreads();  // As / Bs are reused as M / X from here
        MM_DELAY(3);
        if (dt.r_lo == 0 && dt.c_lo == 0 && dt.r_hi == NB && dt.c_hi == NB) {
            // a whole tile (all but the first / last block row): the waves' 32 x 32 quadrants are entirely below the diagonal
            // (wave 2), entirely above it (wave 1: zeros) or on it -- no range tests, a third of the instructions on the chain
            const int wq = wave_id();
            if (wq == 2) {
                MM_ACC_FOREACH(M[row][col] = Ds[row][col] - acc[a][b][i];)
            } else if (wq == 1) {
                MM_ACC_FOREACH(M[row][col] = 0.0;)
            } else {
                MM_ACC_FOREACH(M[row][col] = col <= row ? Ds[row][col] - acc[a][b][i] : 0.0;)
            }
        } else {
            MM_ACC_FOREACH(M[row][col] = (dt.rv(row) && dt.cv(col) && col <= row) ? Ds[row][col] - acc[a][b][i]
                                                                                 : ((!dt.rv(row) && row == col) ? 1.0 : 0.0);)
        }
        __syncthreads();
        int bad = 0;
        double *Lr = linv(side, r);
        MM_TRACE(r, 6);
        // the block streams out while it is factored: column panel k's sub-diagonal blocks, then X_kk (see trsm_stage)
        factor_block_lds(M, X, R, 0, bad, StagePub<MODE>{M, X, dt, Lr, lpubp(side, r), has_sub && !(diag_pre && r == g.a) ? flag(fb, r, 1) : nullptr, Ss, st},
                         side == 0 ? r : -1);
        MM_TRACE(r, 7);
        if (bad && threadIdx.x == 64) {  // `bad` = 1-based position inside the block; report the natural column
            const long col = vec_index(g, side, r, bad - 1);
            report_bad(info, (int)(col >= 0 && col < n ? col + 1 : n));
        }
        for (int e = thread_id(); e < NB * NB; e += 256) {  // L_rr itself: only the later kernels read it (plain stores)
            const int rr = e / NB, cc = e % NB;
            if (dt.rv(rr) && dt.cv(cc) && cc <= rr) *dt.at(rr, cc) = M[rr][cc];
        }
        // the rest of L_rr^-1 (for the substitution kernels) is nobody's critical path
        inverse_offdiag_lds(M, X, T);
        for (int e = threadIdx.x; e < NB * NB; e += 256) {
            const int rr = e / NB, cc = e % NB;
            if ((rr >> 4) != (cc >> 4)) Lr[e] = X[rr][cc];
        }
        MM_DELAY(10);      // (row head, end of row)
        if (b_fwd) {  // y_r = L_rr^-1 (b_r - sum_d L_{r,r-d} y_{r-d})
            if (threadIdx.x < NB) {
                const long vi = vec_index(g, side, r, threadIdx.x);
                rhs[threadIdx.x] = (vi >= 0 && vi < n) ? ld_shared<MODE>(b_fwd + vi) : 0.0;
            }
            if (has_sub) {  // this workgroup owns (r, r-1): still in Ss
                if (!wg_wait<MODE>(yflag(fb, r - 1), nullptr, abort_flag, &s_ok)) MM_FUSED_ABANDON;
                if (threadIdx.x < NB) {
                    const long vi = vec_index(g, side, r - 1, threadIdx.x);
                    ys[threadIdx.x] = (vi >= 0 && vi < n) ? ld_shared<MODE>(y + vi) : 0.0;
                }
                __syncthreads();
                const double tv = tile_matvec<LDT>(Ss, ys);
                if ((threadIdx.x & 3) == 0) rhs[threadIdx.x >> 2] -= tv;
            }
            for (int dd = 2; dd <= bwb && dd <= r; ++dd) {
                if (!wg_wait<MODE>(cflag(fb, r, dd), nullptr, abort_flag, &s_ok)) MM_FUSED_ABANDON;
                if (threadIdx.x < NB) rhs[threadIdx.x] -= ld_shared<MODE>(cslot(side, r, dd) + threadIdx.x);
            }
            if (tw && side == 0 && r >= g.a) {  // a row of M: what the other end's columns contribute (fixed order)
                const int rr = nblk - 1 - r;
                for (int dd = max(1, rr - g.b + 1); dd <= bwb && dd <= rr; ++dd) {
                    if (!wg_wait<MODE>(cflag(fb1(), rr, dd), nullptr, abort_flag, &s_ok)) MM_FUSED_ABANDON;
                    if (threadIdx.x < NB) rhs[threadIdx.x] -= ld_shared<MODE>(cslot(1, rr, dd) + threadIdx.x);
                }
            }
            __syncthreads();
            const double yr = tile_matvec<NB + 1>(X, rhs);
            if ((threadIdx.x & 3) == 0) {
                const long vi = vec_index(g, side, r, threadIdx.x >> 2);
                if (vi >= 0 && vi < n) st_shared<MODE>(y + vi, yr);
            }
            wg_publish<MODE>(yflag(fb, r));
        }
        MM_TRACE(r, 9);
        __syncthreads();  // M / X are overwritten by the next row's tiles
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void chol_band_fused_kernel(double *A, TwGeom g, double *Linv,
                                                              int32_t *__restrict__ flags, int32_t *__restrict__ info,
                                                              const double *b_fwd, double *y,
                                                              double *contrib, double *lpub, double *spub,
                                                              const int32_t *slab_ready, int cams_per_slab, int n_cams) {
    chol_band_fused_body<MODE>(A, g, Linv, flags, info, b_fwd, y, contrib, lpub, spub, slab_ready, cams_per_slab, n_cams, blockIdx.x);
}
__device__ __forceinline__ TwGeom batch_geom(const mm_batch_prob &bp) {
    return TwGeom{bp.chol_n, bp.chol_nblk, bp.chol_bwb, bp.chol_a, bp.chol_m, bp.chol_b, bp.chol_pad};
}
// batched (mm_ba_trf_batched): blockIdx.y picks the problem.  Workgroups are dispatched in order of their linear index, x
// fastest: all workgroups of the problems listed earlier are resident (or done) before a later problem's, and a workgroup
// only ever waits for workgroups of its own problem -- so the batch makes progress with more workgroups than compute
// units, problem by problem.  (The spins stay bounded all the same: info = -1 sends the caller to the single-problem path.)
__global__ __launch_bounds__(256) void chol_band_fused_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    const mm_batch_prob &bp = tab[list[blockIdx.y]];
    if (blockIdx.x >= bp.g_chol) return;
    chol_band_fused_body<2>(bp.S, batch_geom(bp), bp.chol_Linv, bp.chol_flags, bp.info, bp.v, bp.chol_ytmp, bp.chol_contrib, bp.chol_lpub,
                            bp.chol_spub, nullptr, 1, 0, blockIdx.x);
}


#ifdef MM_CHOL_TRACE
extern "C" int mm_debug_chol_trace(unsigned long long *host /*[128*32]*/) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_chol_trace), sizeof(g_chol_trace));
}
#endif

// ---- backward substitution L^T x = y for a narrow band: ONE launch ------------------------------------------------------
// Per side, workgroup 0 owns the diagonal and the first sub-diagonal: x_k = L_kk^-T (y_k - sum_d L_{k+d,k}^T x_{k+d}),
// with the d = 1 term computed locally right after x_{k+1}.  Workgroup d (2 <= d <= bwb) owns the blocks (k + d, k) and
// sends its 64-vector L_{k+d,k}^T x_{k+d} to workgroup 0, which adds the terms in a fixed order.  No flags: x and the
// contribution buffer start out as a NaN sentinel (host memset 0xFF) and every reader polls the very 8 bytes it needs
// (written by one write-through store), so a hand-over costs one memory latency instead of store-ack + flag + load.
// Every tile a workgroup needs next is already in flight (registers) while it waits.
// (Round 4, tried and dropped: the chain's workgroup also owning the SECOND sub-diagonal -- on the theory that the chain
// waits for the d = 2 term's two trips through memory -- made a step 1.6 us LONGER (121 -> 166 us per solve at n = 3000):
// the chain is bound by its own tile commits, matrix-vector products and barriers, not by the hand-over.)
// Two-ended: side 0 walks M then T upwards; side 1 starts from the x of M (polled like any other) and walks its own
// columns in its reversed numbering -- the two chains of a + m and b steps run side by side.
constexpr unsigned long long BWD_SENTINEL = ~0ull;
constexpr size_t BWD_LDS_BYTES = (size_t)(4 * NB * LDT + 4 * NB + 2 * NB) * sizeof(double);      // four tiles: two steps' L_kk^-1 and L_{k,k-1}

__device__ __forceinline__ void tile_prefetch(double (&pre)[16], const TileRef &t) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = threadIdx.x + 256 * q;
        const int r = e / NB, c = e % NB;
        pre[q] = (t.rv(r) && t.cv(c)) ? *t.at(r, c) : 0.0;
    }
}
__device__ __forceinline__ void tile_prefetch_dense(double (&pre)[16], const double *__restrict__ src) {
#pragma unroll
    for (int q = 0; q < 16; ++q) pre[q] = src[threadIdx.x + 256 * q];
}
// out[c] = sum_r T[r][c] v[r] for c = threadIdx.x & 63 (valid in the first 64 threads after the call); 256 threads
__device__ __forceinline__ double tile_matvec_t(const double (*T)[LDT], const double *v, double (*part)[NB]) {
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += T[g * 16 + q][c] * v[g * 16 + q];
    part[g][c] = s;
    __syncthreads();
    return (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}
// The same product straight from the registers a tile was prefetched into (tile_prefetch / tile_prefetch_dense: element
// threadIdx.x + 256 q = row (threadIdx.x >> 6) + 4 q, column threadIdx.x & 63): thread (g, c) adds the rows g, g + 4, ... of
// column c -- the tile never goes through LDS (16 stores + 16 loads per thread and a barrier less per product; round 4).
__device__ __forceinline__ double reg_matvec_t(const double (&pre)[16], const double *v, double (*part)[NB]) {
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += pre[q] * v[g + 4 * q];
    part[g][c] = s;
    __syncthreads();
    return (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}
// poll one double until it is no longer the sentinel (bounded); *failed is set if the wait is abandoned
__device__ __forceinline__ double poll_value(const double *p, int32_t *abort_flag, int &failed) {
    for (long it = 0; it < SPIN_LIMIT; ++it) {
        const double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned long long)__double_as_longlong(v) != BWD_SENTINEL) return v;
        if ((it & 255) == 255 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    failed = 1;
    return 0.0;
}

__device__ __forceinline__ void chol_band_bwd_body(double *A, TwGeom g, const double *__restrict__ Linv,
                                                            const double *__restrict__ y, double *x, double *contrib,
                                                            int32_t *__restrict__ abort_flag, int32_t *__restrict__ info, const unsigned bx) {
    extern __shared__ double smem[];
    double (*T1)[LDT] = reinterpret_cast<double (*)[LDT]>(smem + NB * LDT);      // (side 1's prologue only)
    double (*part)[NB] = reinterpret_cast<double (*)[NB]>(smem + 2 * NB * LDT);
    double *vec = smem + 2 * NB * LDT + 4 * NB, *vec2 = vec + NB;
    const int bwb = g.bwb, W = bwb + 1, nblk = g.nblk, n = g.n;
    const int GB = bwb >= 2 ? bwb : 1;  // workgroups per side
    const int side = (int)bx / GB, lid = (int)bx % GB;
    const bool tw = g.b > 0;
    const int nrows = side == 0 ? (tw ? g.a + g.m : nblk) : g.b + g.m;
    const int ncols = side == 0 ? nrows : g.b;
    auto nat = [&](int blk) { return side == 0 ? blk : nblk - 1 - blk; };
    auto cslot = [&](int k, int d) { return contrib + (((size_t)side * nblk + k) * W + d) * NB; };
    // x is POLLED in the d = 0 slots of side 0 (no step uses them; natural index vi -> block vi / 64), which start out as the
    // sentinel with the rest of the buffer; `x` itself -- the caller's right-hand side, consumed by the forward
    // substitution -- only receives plain copies, so nothing has to fill it between the two kernels
    auto xpoll = [&](long vi) { return contrib + (size_t)(vi >> 6) * W * NB + (vi & 63); };
    // element threadIdx.x of vector block blk (natural storage); false: a virtual padding row
    auto vpos = [&](int blk, long &vi) -> bool {
        vi = vec_index(g, side, blk, threadIdx.x);
        return vi >= 0 && vi < n;
    };
    int failed = 0;
#ifndef MM_CHOL_BWD_CHAIN_WAVE
#define MM_CHOL_BWD_CHAIN_WAVE 1
#endif
    if (lid == 0 && MM_CHOL_BWD_CHAIN_WAVE) {
        // The chain  x_k = L_kk^-T (y_k - L_{k+1,k}^T x_{k+1} - contributions)  on ONE wave (round 4).  Delay injection showed
        // the chain workgroup itself pacing this kernel (its helpers have slack): four waves sharing each product paid five
        // workgroup barriers and as many LDS round trips per step (3.1 us).  Here wave 0 walks the chain alone -- lane c owns
        // element c of every vector, a product is 64 multiply-adds per lane with the tile and the vector read from LDS, wave
        // barriers only -- and waves 1 .. 3 keep the tiles of the next step coming into the other half of a double buffer.
        double (*TA)[NB][LDT] = reinterpret_cast<double (*)[NB][LDT]>(smem);                      // TA[b]: L_kk^-1 of a step with parity b
        double (*TB)[NB][LDT] = reinterpret_cast<double (*)[NB][LDT]>(smem + 2 * NB * LDT);       // TB[b]: L_{k,k-1}
        double *cvec = smem + 4 * NB * LDT, *cvec2 = cvec + NB;
        __shared__ int s_loaded[3], s_done, s_dead;      // steps each loader wave has delivered | steps the chain has left
        const int ktop = ncols - 1;  // first block this chain solves for
        if (ktop < 0) return;
        double local = 0.0;  // L_{k+1,k}^T x_{k+1}, element lane (wave 0)
        if (threadIdx.x == 0) {
            s_loaded[0] = s_loaded[1] = s_loaded[2] = 0;
            s_done = 0;
            s_dead = 0;
        }
        if (side == 1 && ktop + 1 < nrows && bwb >= 1) {
            // the row above the chain's first block belongs to M: x of that row comes from side 0, the block
            // (ktop + 1, ktop) is this side's -> `local` for the first step (all four waves, once)
            double (*T1_)[LDT] = TB[1];
            double (*part_)[NB] = reinterpret_cast<double (*)[NB]>(&TA[1][0][0]);
            double pt[16];
            tile_prefetch(pt, tile_ref(A, g, side, ktop + 1, ktop));
            tile_commit(T1_, pt);
            if (threadIdx.x < NB) {
                long vi;
                cvec2[threadIdx.x] = vpos(ktop + 1, vi) ? poll_value(xpoll(vi), abort_flag, failed) : 0.0;
            }
            if (__syncthreads_or(failed)) MM_FUSED_ABANDON;
            local = tile_matvec_t(T1_, cvec2, part_);      // (valid in the first 64 threads = wave 0)
        }
        __syncthreads();
        const int wv = wave_id();
        volatile int *loaded = s_loaded, *done = &s_done, *deadp = &s_dead;
        if (wv != 0) {
            // loaders: the two tiles of step j (k = ktop - j) into buffer j & 1, once the chain has left step j - 2
            const int lt = (int)threadIdx.x - 64;
            // (all loads of a step in flight at once: three dependent batches per step made the loaders the chain's pace.  Tried:
            // requesting step j + 1 before waiting for the buffer, registers as a third buffer -- 0.604 instead of 0.566 ms.)
            constexpr int NLD = (NB * NB + 191) / 192;
            for (int j = 0; j <= ktop; ++j) {
                const int k = ktop - j, b = j & 1;
                while (*done < j - 1 && !*deadp) __builtin_amdgcn_s_sleep(1);
                if (*deadp) return;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                MM_DELAY(24);      // (backward loaders, buffer free)
                const double *li = Linv + (size_t)nat(k) * NB * NB;
                const TileRef tr = tile_ref(A, g, side, k, k > 0 ? k - 1 : 0);
                double va[NLD], vb[NLD];
#pragma unroll
                for (int q = 0; q < NLD; ++q) {
                    const int e = lt + 192 * q, r_ = e / NB, c_ = e % NB;
                    va[q] = e < NB * NB ? li[e] : 0.0;
                    vb[q] = (k > 0 && e < NB * NB && tr.rv(r_) && tr.cv(c_)) ? *tr.at(r_, c_) : 0.0;
                }
#pragma unroll
                for (int q = 0; q < NLD; ++q) {
                    const int e = lt + 192 * q;
                    if (e < NB * NB) {
                        TA[b][e / NB][e % NB] = va[q];
                        TB[b][e / NB][e % NB] = vb[q];
                    }
                }
                lds_counter_set(loaded + (wv - 1), j + 1);
            }
            return;
        }
        // ---- wave 0: the chain ----
        const int lane = lane_id();
        double cv[FUSED_MAX_BWB + 1], yk = 0.0;
        auto vposl = [&](int blk, long &vi) -> bool {
            vi = vec_index(g, side, blk, lane);
            return vi >= 0 && vi < n;
        };
        auto request = [&](int k) {
            if (k < 0) return;
            long vi;
            yk = vposl(k, vi) ? y[vi] : 0.0;
#pragma unroll
            for (int dd = 2; dd <= FUSED_MAX_BWB; ++dd)
                if (dd <= bwb && k + dd < nrows) cv[dd] = __hip_atomic_load(cslot(k, dd) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // out[lane] = sum_r T[r][lane] v[r]: four independent partial sums (the FMAs of one chain would wait for each other)
        auto col_product = [&](const double (*T)[LDT], const double *v) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
            for (int r_ = 0; r_ < NB; r_ += 4) {
                s0 = fma(T[r_][lane], v[r_], s0);
                s1 = fma(T[r_ + 1][lane], v[r_ + 1], s1);
                s2 = fma(T[r_ + 2][lane], v[r_ + 2], s2);
                s3 = fma(T[r_ + 3][lane], v[r_ + 3], s3);
            }
            return (s0 + s1) + (s2 + s3);
        };
        request(ktop);
        for (int j = 0; j <= ktop; ++j) {
            const int k = ktop - j, b = j & 1;
            MM_DELAY(21);      // (backward chain, top of a step)
            double rhs = yk - local;
#pragma unroll
            for (int dd = 2; dd <= FUSED_MAX_BWB; ++dd)  // fixed summation order
                if (dd <= bwb && k + dd < nrows) {
                    if ((unsigned long long)__double_as_longlong(cv[dd]) == BWD_SENTINEL)
                        cv[dd] = poll_value(cslot(k, dd) + lane, abort_flag, failed);
                    rhs -= cv[dd];
                }
            if (__builtin_amdgcn_ballot_w64(failed != 0)) {
                if (lane == 0) *deadp = 1;
                MM_FUSED_ABANDON;
            }
            cvec[lane] = rhs;
            lds_counter_wait(loaded, j + 1);
            lds_counter_wait(loaded + 1, j + 1);
            lds_counter_wait(loaded + 2, j + 1);
            wave_lds_sync();
            MM_DELAY(25);      // (backward chain, tiles there, before the first product)
            const double xk = col_product(TA[b], cvec);
            long vi;
            const bool okv = vposl(k, vi);
            if (okv) {
                st_shared<2>(xpoll(vi), xk);
                x[vi] = xk;
            }
            cvec2[lane] = okv ? xk : 0.0;
            request(k - 1);
            MM_DELAY(22);      // (backward chain, x_k is out)
            wave_lds_sync();
            if (k > 0) local = col_product(TB[b], cvec2);
            lds_counter_set(done, j + 1);
        }
        return;
    }
    // tiles are fetched two steps ahead (a step is shorter than a trip to memory): two register slots, loop unrolled by 2
    if (lid == 0) {
        double p0a[16], p1a[16], p0b[16], p1b[16];
        auto fetch = [&](int k, double (&p0)[16], double (&p1)[16]) {  // tiles of step k: L_kk^-1 and L_{k,k-1}
            if (k < 0) return;
            tile_prefetch_dense(p0, Linv + (size_t)nat(k) * NB * NB);
            if (k > 0) tile_prefetch(p1, tile_ref(A, g, side, k, k - 1));
        };
        double local = 0.0;  // L_{k+1,k}^T x_{k+1}, element threadIdx.x (first 64 threads)
        bool dead = false;
        // y_k and the contributions of step k are requested during step k + 1 (right after x_{k+1} went out), so their
        // memory latency overlaps the local matrix-vector work; what is still the sentinel then gets polled
        double cv[FUSED_MAX_BWB + 1], yk = 0.0;
        auto request = [&](int k) {
            if (k < 0 || threadIdx.x >= NB) return;
            long vi;
            yk = vpos(k, vi) ? y[vi] : 0.0;
#pragma unroll
            for (int dd = 2; dd <= FUSED_MAX_BWB; ++dd)
                if (dd <= bwb && k + dd < nrows)
                    cv[dd] = __hip_atomic_load(cslot(k, dd) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto step = [&](int k, double (&p0)[16], double (&p1)[16]) {
            MM_DELAY(21);      // (backward chain, top of a step)
            if (threadIdx.x < NB) {
                double rhs = yk - local;
#pragma unroll
                for (int dd = 2; dd <= FUSED_MAX_BWB; ++dd)  // fixed summation order
                    if (dd <= bwb && k + dd < nrows) {
                        if ((unsigned long long)__double_as_longlong(cv[dd]) == BWD_SENTINEL)
                            cv[dd] = poll_value(cslot(k, dd) + threadIdx.x, abort_flag, failed);
                        rhs -= cv[dd];
                    }
                vec[threadIdx.x] = rhs;
            }
            if (__syncthreads_or(failed)) {
                dead = true;
                return;
            }
            const double xk = reg_matvec_t(p0, vec, part);
            if (threadIdx.x < NB) {
                long vi;
                const bool ok = vpos(k, vi);
                if (ok) {
                    st_shared<2>(xpoll(vi), xk);
                    x[vi] = xk;
                }
                vec2[threadIdx.x] = ok ? xk : 0.0;
            }
            request(k - 1);
            MM_DELAY(22);      // (backward chain, x_k is out)
            __syncthreads();
            if (k > 0) local = reg_matvec_t(p1, vec2, part);
            fetch(k - 2, p0, p1);      // (both register sets of this step are spent)
            __syncthreads();  // vec / vec2 / part are rewritten by the next step
        };
        const int ktop = ncols - 1;  // first block this chain solves for
        if (ktop < 0) return;
        if (side == 1 && ktop + 1 < nrows && bwb >= 1) {
            // the row above the chain's first block belongs to M: x of that row comes from side 0, the block
            // (ktop + 1, ktop) is this side's -> `local` for the first step
            double pt[16];
            tile_prefetch(pt, tile_ref(A, g, side, ktop + 1, ktop));
            tile_commit(T1, pt);
            if (threadIdx.x < NB) {
                long vi;
                vec2[threadIdx.x] = vpos(ktop + 1, vi) ? poll_value(xpoll(vi), abort_flag, failed) : 0.0;
            }
            if (__syncthreads_or(failed)) MM_FUSED_ABANDON;
            local = tile_matvec_t(T1, vec2, part);
            __syncthreads();
        }
        request(ktop);
        fetch(ktop, p0a, p1a);
        fetch(ktop - 1, p0b, p1b);
        for (int k = ktop; k >= 0 && !dead; k -= 2) {
            step(k, p0a, p1a);
            if (k - 1 >= 0 && !dead) step(k - 1, p0b, p1b);
        }
        if (dead) MM_FUSED_ABANDON;
        return;
    }
    const int d = lid + 1;  // 2 .. bwb
    const int kfirst = min(nrows - 1 - d, ncols - 1);
    if (kfirst < 0) return;
    double pa[16], pb[16];
    auto fetch = [&](int k, double (&pp)[16]) {
        if (k >= 0) tile_prefetch(pp, tile_ref(A, g, side, k + d, k));
    };
    bool dead = false;
    auto step = [&](int k, double (&pp)[16]) {
        if (threadIdx.x < NB) {
            long vi;
            vec[threadIdx.x] = vpos(k + d, vi) ? poll_value(xpoll(vi), abort_flag, failed) : 0.0;
        }
        if (__syncthreads_or(failed)) {
            dead = true;
            return;
        }
        MM_DELAY(23);      // (backward helpers, x_{k+d} seen)
        const double t = reg_matvec_t(pp, vec, part);
        if (threadIdx.x < NB) st_shared<2>(cslot(k, d) + threadIdx.x, t);
        fetch(k - 2, pp);
        __syncthreads();
    };
    fetch(kfirst, pa);
    fetch(kfirst - 1, pb);
    for (int k = kfirst; k >= 0 && !dead; k -= 2) {
        step(k, pa);
        if (k - 1 >= 0 && !dead) step(k - 1, pb);
    }
    if (dead) MM_FUSED_ABANDON;
}
__global__ __launch_bounds__(256) void chol_band_bwd_kernel(double *A, TwGeom g, const double *__restrict__ Linv,
                                                            const double *__restrict__ y, double *x, double *contrib,
                                                            int32_t *__restrict__ abort_flag, int32_t *__restrict__ info) {
    chol_band_bwd_body(A, g, Linv, y, x, contrib, abort_flag, info, blockIdx.x);
}
__global__ __launch_bounds__(256) void chol_band_bwd_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    const mm_batch_prob &bp = tab[list[blockIdx.y]];
    if (blockIdx.x >= bp.g_bwd) return;
    chol_band_bwd_body(bp.S, batch_geom(bp), bp.chol_Linv, bp.chol_ytmp, bp.v, bp.chol_contrib_bwd, bp.chol_flags, bp.info, blockIdx.x);
}


// one launch instead of a handful of fills per solve: info = 0, flags = 0, sentinels into the backward kernel's
// contribution buffer (which also holds the words x is polled on), the hand-over buffer of the streamed blocks and the
// diagonal 16 x 16 blocks of L^-1 (what the consumers of the factorisation poll on).  The body lives in chol_init.h:
// schur_prepare_kernel runs it too when mm_ba_trf hands it the whole reduced solve (one launch less per solve).
__global__ __launch_bounds__(256) void chol_init_kernel(mm_chol_init_args a) { mm_chol_init_body(a, blockIdx.x, gridDim.x); }
constexpr unsigned CHOL_INIT_GRID = 128;
__global__ __launch_bounds__(256) void chol_init_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    const mm_batch_prob &bp = tab[list[blockIdx.y]];
    const mm_chol_init_args a = {bp.info, bp.chol_flags, (size_t)bp.chol_nflags, (unsigned long long *)bp.chol_contrib_bwd, (size_t)bp.chol_nsent,
                                 (unsigned long long *)bp.chol_lpub, (size_t)bp.chol_nlpub, (unsigned long long *)bp.chol_Linv, (size_t)bp.chol_nblk};
    mm_chol_init_body(a, blockIdx.x, CHOL_INIT_GRID);
}
// copy the band of the lower triangle into the upper triangle for the columns side 1 eliminates: (j, i) <- (i, j) for
// i >= row0, 0 < i - j <= hb (callers of the two-ended path that only filled the lower triangle)
__global__ __launch_bounds__(256) void chol_mirror_kernel(double *A, int n, int row0, int hb) {
    const int i = row0 + blockIdx.x;
    for (int o = 1 + threadIdx.x; o <= hb; o += 256) {
        const int j = i - o;
        if (j >= 0) A[(size_t)j * n + i] = A[(size_t)i * n + j];
    }
}

// ---- solves -----------------------------------------------------------------------------------------------------------
// z = Linv * v (transpose = 0) or Linv^T * v into LDS vector `out`; 256 threads, 4 per row.
__device__ __forceinline__ void block_gemv64(const double *__restrict__ Linv, const double *vin, double *out,
                                             int transpose) {
    const int r = threadIdx.x >> 2, part = threadIdx.x & 3;
    double s = 0.0;
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        const int k = part * 16 + q;
        s += (transpose ? Linv[k * NB + r] : Linv[r * NB + k]) * vin[k];
    }
    s += __shfl_down(s, 2, 4);
    s += __shfl_down(s, 1, 4);
    if (part == 0) out[r] = s;
}

// forward step k: y_k = Linv_kk b_k (every workgroup, redundantly); workgroup 0 publishes y_k; rows below (inside the
// band) get b_i -= L_ik y_k, 32 rows per workgroup, 8 lanes per row.
__global__ __launch_bounds__(256) void fwd_step_kernel(const double *__restrict__ A, const double *__restrict__ Linv,
                                                       double *__restrict__ b, double *__restrict__ y, int n, int k0,
                                                       int row_end) {
    __shared__ double vin[NB], yk[NB];
    const int kc = min(NB, n - k0);
    if (threadIdx.x < NB) vin[threadIdx.x] = threadIdx.x < kc ? b[k0 + threadIdx.x] : 0.0;
    __syncthreads();
    block_gemv64(Linv, vin, yk, 0);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < kc) y[k0 + threadIdx.x] = yk[threadIdx.x];
    const int row = k0 + NB + blockIdx.x * 32 + (threadIdx.x >> 3);
    const int part = threadIdx.x & 7;
    double s = 0.0;
    if (row < row_end) {
        const double *Lr = A + (size_t)row * n + k0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = part * 8 + q;
            if (k < kc) s += Lr[k] * yk[k];
        }
    }
    s += __shfl_down(s, 4, 8);
    s += __shfl_down(s, 2, 8);
    s += __shfl_down(s, 1, 8);
    if (row < row_end && part == 0) b[row] -= s;
}

// backward step k: x_k = Linv_kk^T y_k (redundantly); workgroup 0 publishes x_k into `x`; columns left of the block
// (inside the band) get y_j -= L_kj^T x_k, one column per thread.
__global__ __launch_bounds__(256) void bwd_step_kernel(const double *__restrict__ A, const double *__restrict__ Linv,
                                                       double *__restrict__ y, double *__restrict__ x, int n, int k0,
                                                       int col_begin) {
    __shared__ double vin[NB], xk[NB];
    const int kr = min(NB, n - k0);
    if (threadIdx.x < NB) vin[threadIdx.x] = threadIdx.x < kr ? y[k0 + threadIdx.x] : 0.0;
    __syncthreads();
    block_gemv64(Linv, vin, xk, 1);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < kr) x[k0 + threadIdx.x] = xk[threadIdx.x];
    // 64 columns per workgroup, 4 row groups of 16: sixteen independent loads per thread instead of a 64-long chain
    __shared__ double part[4][NB];
    const int col = col_begin + blockIdx.x * NB + (threadIdx.x & 63);
    const int rg = threadIdx.x >> 6;
    double s = 0.0;
    if (col < k0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int r = rg * 16 + q;
            if (r < kr) s += A[(size_t)(k0 + r) * n + col] * xk[r];
        }
    }
    part[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && col < k0) y[col] -= (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

}  // namespace

extern "C" {

size_t mm_chol_workspace_bytes(int n) {
    size_t nblk = (size_t)(n + NB - 1) / NB;
    return mm_align_up(nblk * NB * NB * sizeof(double), 256) + mm_align_up((size_t)(n + NB) * sizeof(double), 256) +
           mm_align_up((2 * (2 * nblk * (FUSED_MAX_BWB + 1) + 2 * nblk) + 64 + FUSED_MAX_BWB * FUSED_MAX_BWB) * sizeof(int32_t), 256) +
           2 * mm_align_up(2 * nblk * (FUSED_MAX_BWB + 1) * NB * sizeof(double), 256) +  // forward + backward contributions
           mm_align_up(nblk * (LPUB_BLOCK + (size_t)(2 + 2 * (FUSED_MAX_BWB - 2)) * NB * NB) * sizeof(double), 256);   // hand-over buffers of the streamed blocks
}

int mm_chol_solve(mm_ctx *ctx, double *A, int n, double *b, int nrhs, int half_bandwidth, int32_t *info, void *ws,
                  size_t ws_bytes) {
    return mm_chol_solve_gated(ctx, A, n, b, nrhs, half_bandwidth, info, ws, ws_bytes, nullptr, 1, 0, 0);
}

int mm_chol_solve_sym(mm_ctx *ctx, double *A, int n, double *b, int half_bandwidth, int both_triangles, int32_t *info,
                      void *ws, size_t ws_bytes) {
    return mm_chol_solve_gated(ctx, A, n, b, 1, half_bandwidth, info, ws, ws_bytes, nullptr, 1, 0, both_triangles ? 2 : 1);
}

}  // extern "C"

// Co-residency budget of the single-launch factorisation: its workgroups spin on each other's results, so every one of
// them must hold a compute unit (512 registers per lane: one workgroup per CU) for the grid to make progress.  One grid is
// far below the CU count, but several contexts may solve at once (the sliding-window schedule runs 8 streams): each
// launch reserves its grid here, a launch that would push the sum past the device's CUs takes the launch-per-column path
// instead, and a context returns its share at its next host synchronisation.  (Conservative: a reservation outlives its
// kernel until the owner synchronises.  Per process and device 0..15; other processes on the GPU are not seen -- their
// case is covered by the bounded spins + mm_ba_trf's retry on the per-column path.)
// A reservation is also returned without a host synchronisation of its owner: every single-launch solve records an
// event behind its last kernel, and a context whose event has completed gives its share back at its own next
// reservation -- or when ANOTHER context finds the budget exhausted and sweeps the registry (the Python-sequenced
// driver and direct mm_chol_solve* callers synchronise through torch and never reach mm_ctx_sync).
static std::mutex g_budget_mu;
static int g_fused_reserved[16];
static std::vector<mm_ctx *> g_budget_holders;      // contexts with fused_wgs > 0

static void budget_drop_locked(mm_ctx *ctx) {
    if (ctx->fused_wgs > 0) {
        g_fused_reserved[ctx->device & 15] -= ctx->fused_wgs;
        ctx->fused_wgs = 0;
        for (size_t i = 0; i < g_budget_holders.size(); ++i)
            if (g_budget_holders[i] == ctx) {
                g_budget_holders[i] = g_budget_holders.back();
                g_budget_holders.pop_back();
                break;
            }
    }
    ctx->fused_ev_pending = false;
}

void mm_chol_release_budget(mm_ctx *ctx) {
    if (!ctx) return;
    std::lock_guard<std::mutex> lk(g_budget_mu);
    budget_drop_locked(ctx);
}

// the solve whose grid was reserved has been enqueued completely: remember where it ends on the stream
static void chol_budget_mark(mm_ctx *ctx) {
    if (ctx->fused_no_event) return;
    if (!ctx->fused_ev && hipEventCreateWithFlags(&ctx->fused_ev, hipEventDisableTiming) != hipSuccess) {
        ctx->fused_ev = nullptr;
        return;
    }
    if (hipEventRecord(ctx->fused_ev, ctx->stream) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_budget_mu);
    ctx->fused_ev_pending = ctx->fused_wgs > 0;
}

static bool chol_reserve_budget(mm_ctx *ctx, int grid) {
    std::lock_guard<std::mutex> lk(g_budget_mu);
    int &total = g_fused_reserved[ctx->device & 15];
    // this context's earlier launch is stream-ordered before the new one: its share is replaced, not added to
    const int own = ctx->fused_wgs;
    if (ctx->cu_count > 0 && total - own + grid > ctx->cu_count) {
        // exhausted: collect the shares of contexts whose solves have finished meanwhile (never this context's own --
        // its earlier kernel may still be running and keeps its share until the stream has passed it)
        for (size_t i = 0; i < g_budget_holders.size();) {
            mm_ctx *o = g_budget_holders[i];
            if (o != ctx && o->device == ctx->device && o->fused_ev_pending && hipEventQuery(o->fused_ev) == hipSuccess)
                budget_drop_locked(o);      // (swaps the last holder into slot i)
            else
                ++i;
        }
        if (total - own + grid > ctx->cu_count) return false;
    }
    total += grid - own;
    if (own == 0 && grid > 0) g_budget_holders.push_back(ctx);
    ctx->fused_wgs = grid;
    ctx->fused_ev_pending = false;
    return true;
}

static int chol_fused_mode() {
    static const int mode = [] {
        const char *e = getenv("MM_CHOL_FUSED");
        return e ? atoi(e) : 2;
    }();
    return mode;
}

static bool chol_twisted_enabled() {
    static const bool on = [] {
        const char *e = getenv("MM_CHOL_TWISTED");
        return !e || atoi(e) != 0;
    }();
    return on;
}

bool mm_chol_fused_eligible(int n, int half_bandwidth) {
    const int nblk = (n + NB - 1) / NB;
    long bwb_l = ((long)half_bandwidth + NB - 1) / NB;
    const int bwb = bwb_l > nblk ? nblk : (int)bwb_l;
    return chol_fused_mode() > 0 && nblk >= 2 && bwb >= 1 && bwb <= FUSED_MAX_BWB && !(n & 1);
}

// mm_chol_solve with the rows of A / b gated by slab flags (slab_ready == nullptr: everything is there already).
// sym_mode 0: only the lower triangle is valid and it must come back as the plain Cholesky factor (mm_chol_solve);
// 1: only the lower triangle is valid, the layout of the factor is free; 2: both triangles of the band are valid.
// With 1 / 2 (and one right-hand side) a narrow band is eliminated from both ends at once.
int mm_chol_solve_gated(mm_ctx *ctx, double *A, int n, double *b, int nrhs, int half_bandwidth, int32_t *info, void *ws,
                        size_t ws_bytes, const int32_t *slab_ready, int cams_per_slab, int n_cams, int sym_mode) {
    if (!ctx) return MM_ERR_ARG;
    if (n == 0) return MM_OK;
    if (!A || !info || n < 0 || nrhs < 0 || (nrhs > 0 && !b) || half_bandwidth < 0)
        return mm_fail(ctx, MM_ERR_ARG, "mm_chol_solve: bad argument");
    if (!ws || ws_bytes < mm_chol_workspace_bytes(n)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_chol_solve: workspace too small");
    if (((uintptr_t)A & 15) || (n & 1)) return mm_fail(ctx, MM_ERR_ARG, "mm_chol_solve: A must be 16-byte aligned and n even");
    const int nblk = (n + NB - 1) / NB;
    double *Linv = (double *)ws;
    double *ytmp = (double *)((char *)ws + mm_align_up((size_t)nblk * NB * NB * sizeof(double), 256));
    // block (bi, bj) can be non-zero iff 64 (bi - bj) - 63 <= half_bandwidth
    long bwb_l = ((long)half_bandwidth + NB - 1) / NB;
    const int bwb = bwb_l > nblk ? nblk : (int)bwb_l;
    bool fwd_done = false;  // forward substitution of right-hand side 0 already done by the fused kernel
    const int fused_mode = chol_fused_mode();
    if (slab_ready && !mm_chol_fused_eligible(n, half_bandwidth))
        return mm_fail(ctx, MM_ERR_ARG, "mm_chol_solve_gated: gating needs the single-launch factorisation");
    bool fused = fused_mode > 0 && nblk >= 2 && bwb >= 1 && bwb <= FUSED_MAX_BWB && (long)NB * nblk * n < (1L << 31);
    if (fused && ctx->chol_avoid_fused && !slab_ready) fused = false;      // (after an abandoned attempt: mm_ba_trf)
    TwGeom g = {n, nblk, bwb, nblk, 0, 0, nblk * NB - n};
    if (fused && sym_mode > 0 && nrhs == 1 && chol_twisted_enabled() && nblk - bwb >= 4) {
        g.m = bwb;                       // the separator: no coupling across 64 m + 1 > half_bandwidth
        g.a = (nblk - g.m + 1) / 2;
        g.b = nblk - g.m - g.a;
    }
    const int G_side = (bwb + 1) + bwb * (bwb - 1) / 2;
    // two-ended: a third group of workgroups, one per block of M x M (fewer than G: m = bwb), pre-accumulates
    const int fused_grid = g.b > 0 ? 2 * G_side + g.m * (g.m + 1) / 2 : G_side;
    if (fused && !chol_reserve_budget(ctx, fused_grid)) {
        if (slab_ready) return mm_fail(ctx, MM_ERR_HIP, "mm_chol_solve_gated: no room for the single-launch factorisation");
        if (ctx->chol_strict_budget) {      // sharded solve: say so instead of diverging from the other ranks' path
            ctx->chol_last_path = -1;
            MM_HIP(ctx, hipMemsetAsync(info, 0xFF, sizeof(int32_t), ctx->stream));      // info = -1
            return MM_OK;
        }
        fused = false;
        g = TwGeom{n, nblk, bwb, nblk, 0, 0, nblk * NB - n};
    }
    ctx->chol_last_path = fused ? 1 : 0;
    struct InitTokenReset {      // a pre-initialisation is good for exactly one solve
        mm_ctx *c;
        ~InitTokenReset() { c->chol_init_done = nullptr; }
    } token_reset{ctx};
    int32_t *flags = (int32_t *)((char *)ytmp + mm_align_up((size_t)(n + NB) * sizeof(double), 256));
    double *contrib = (double *)((char *)flags + mm_align_up((2 * (2 * (size_t)nblk * (FUSED_MAX_BWB + 1) + 2 * nblk) + 64 + FUSED_MAX_BWB * FUSED_MAX_BWB) * sizeof(int32_t), 256));
    const int sides = g.b > 0 ? 2 : 1;
    double *contrib_bwd = (double *)((char *)contrib + mm_align_up(2 * (size_t)nblk * (FUSED_MAX_BWB + 1) * NB * sizeof(double), 256));
    double *lpub = (double *)((char *)contrib_bwd + mm_align_up(2 * (size_t)nblk * (FUSED_MAX_BWB + 1) * NB * sizeof(double), 256));
    if (!fused) MM_HIP(ctx, hipMemsetAsync(info, 0, sizeof(int32_t), ctx->stream));
    if (fused) {
        if (!ctx->attr_chol_fused) {
            MM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(chol_band_fused_kernel<2>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS_BYTES));
            ctx->attr_chol_fused = true;
        }
        if (g.b > 0 && sym_mode == 1) {  // side 1 reads (and overwrites) the upper triangle of its columns: whole tiles,
            const long reach = (long)NB * bwb + NB - 1;   // i.e. also the structural zeros between the band and the tile edge
            MM_LAUNCH(ctx, "chol_mirror_kernel", chol_mirror_kernel, dim3(n - (g.a + g.m) * NB), dim3(256), 0, A, n,
                      (g.a + g.m) * NB, (int)(reach < n - 1 ? reach : n - 1));
        }
        const size_t nflags = 1 + 2 * (2 * (size_t)nblk * (bwb + 1) + 2 * nblk) + (size_t)g.m * g.m;
        static const bool no_fwd = getenv("MM_CHOL_NO_FWD") && atoi(getenv("MM_CHOL_NO_FWD")) > 0;      // (diagnostic: the forward substitution by the launch-per-column kernels)
        const double *b_fwd = nrhs >= 1 && !no_fwd ? b : nullptr;  // the first right-hand side rides along
        double *spub = lpub + (size_t)nblk * LPUB_BLOCK;
        const mm_chol_init_args ia = {info, flags, nflags, (unsigned long long *)contrib_bwd, (size_t)sides * nblk * (bwb + 1) * NB,
                                      (unsigned long long *)lpub, chol_nlpub(nblk, bwb), (unsigned long long *)Linv,
                                      (size_t)nblk};
        if (ctx->chol_init_done == ws && ctx->chol_init_sides == sides) {
            // (the caller's own kernel ran mm_chol_init_body with mm_chol_init_plan's arguments for this workspace)
        } else {
            MM_LAUNCH(ctx, "chol_init_kernel", chol_init_kernel, dim3(128), dim3(256), 0, ia);
        }
        if (ctx->debug_abandon > 0) {      // test hook (mm_ctx_control): behave as if a workgroup had given up waiting
            --ctx->debug_abandon;
            const int32_t one = 1;
            MM_HIP(ctx, hipMemcpyAsync(flags, &one, sizeof(one), hipMemcpyHostToDevice, ctx->stream));
        }
        MM_LAUNCH(ctx, "chol_band_fused_kernel", chol_band_fused_kernel<2>, dim3(fused_grid), dim3(256), FUSED_LDS_BYTES, A, g, Linv,
                  flags, info, b_fwd, ytmp, contrib, lpub, spub, slab_ready, cams_per_slab, n_cams);
        fwd_done = b_fwd != nullptr;
    } else {
        for (int k = 0; k < nblk; ++k) {
            const int k0 = k * NB;
            double *Lk = Linv + (size_t)k * NB * NB;
            MM_LAUNCH(ctx, "chol_diag_kernel", chol_diag_kernel, dim3(1), dim3(256), 0, A, n, k0, Lk, info);
            int m = nblk - k - 1;
            if (m > bwb) m = bwb;
            if (m > 0) {
                MM_LAUNCH(ctx, "chol_panel_kernel", chol_panel_kernel, dim3(m), dim3(256), 0, A, n, k0, (const double *)Lk);
                MM_LAUNCH(ctx, "chol_update_kernel", chol_update_kernel, dim3(m * (m + 1) / 2), dim3(256), 0, A, n, k0);
            }
        }
    }
    for (int c = 0; c < nrhs; ++c) {
        double *bc = b + (size_t)c * n;
        for (int k = 0; k < nblk && !(c == 0 && fwd_done); ++k) {  // L y = b
            const int k0 = k * NB;
            long re = (long)k0 + NB + (long)bwb * NB;
            const int row_end = re > n ? n : (int)re;
            const int below = row_end - (k0 + NB);
            const int grid = below > 0 ? (below + 31) / 32 : 1;
            MM_LAUNCH(ctx, "fwd_step_kernel", fwd_step_kernel, dim3(grid), dim3(256), 0, (const double *)A,
                      (const double *)(Linv + (size_t)k * NB * NB), bc, ytmp, n, k0, row_end);
        }
        if (fused) {  // L^T x = y in one launch
            if (!ctx->attr_chol_bwd) {
                MM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(chol_band_bwd_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS_BYTES));
                ctx->attr_chol_bwd = true;
            }
            // the contribution buffer (and with it the words x is polled on) starts as the NaN sentinel (for the first
            // right-hand side chol_init_kernel prepared it; an abort flag left by the factorisation makes the kernel
            // leave at once -- info is -1 then anyway)
            if (c > 0) {
                MM_HIP(ctx, hipMemsetAsync(flags, 0, sizeof(int32_t), ctx->stream));
                MM_HIP(ctx, hipMemsetAsync(contrib_bwd, 0xFF, (size_t)sides * nblk * (bwb + 1) * NB * sizeof(double), ctx->stream));
            }
            MM_LAUNCH(ctx, "chol_band_bwd_kernel", chol_band_bwd_kernel, dim3(sides * (bwb >= 2 ? bwb : 1)), dim3(256),
                      BWD_LDS_BYTES, A, g, (const double *)Linv, (const double *)ytmp, bc, contrib_bwd, flags, info);
            continue;
        }
        for (int k = nblk - 1; k >= 0; --k) {  // L^T x = y
            const int k0 = k * NB;
            long cb = (long)k0 - (long)bwb * NB;
            const int col_begin = cb < 0 ? 0 : (int)cb;
            const int left = k0 - col_begin;
            const int grid = left > 0 ? (left + NB - 1) / NB : 1;
            MM_LAUNCH(ctx, "bwd_step_kernel", bwd_step_kernel, dim3(grid), dim3(256), 0, (const double *)A,
                      (const double *)(Linv + (size_t)k * NB * NB), ytmp, bc, n, k0, col_begin);
        }
    }
    if (fused) chol_budget_mark(ctx);
    return MM_OK;
}

// see mm_common.h: the same geometry / layout lines as mm_chol_solve_gated(sym_mode 2, one right-hand side, not gated)
bool mm_chol_init_plan(mm_ctx *ctx, int n, int half_bandwidth, int32_t *info, void *ws, size_t ws_bytes, mm_chol_init_args *out, int *sides_out, int *bwb_out) {
    if (!ctx || !ws || !info || !out || n <= 0 || (n & 1) || half_bandwidth < 0 || ws_bytes < mm_chol_workspace_bytes(n)) return false;
    const int nblk = (n + NB - 1) / NB;
    long bwb_l = ((long)half_bandwidth + NB - 1) / NB;
    const int bwb = bwb_l > nblk ? nblk : (int)bwb_l;
    const bool fused = chol_fused_mode() > 0 && nblk >= 2 && bwb >= 1 && bwb <= FUSED_MAX_BWB && (long)NB * nblk * n < (1L << 31) &&
                       !ctx->chol_avoid_fused;
    if (!fused) return false;
    TwGeom g = {n, nblk, bwb, nblk, 0, 0, nblk * NB - n};
    if (chol_twisted_enabled() && nblk - bwb >= 4) {
        g.m = bwb;
        g.a = (nblk - g.m + 1) / 2;
        g.b = nblk - g.m - g.a;
    }
    const int sides = g.b > 0 ? 2 : 1;
    double *Linv = (double *)ws;
    double *ytmp = (double *)((char *)ws + mm_align_up((size_t)nblk * NB * NB * sizeof(double), 256));
    int32_t *flags = (int32_t *)((char *)ytmp + mm_align_up((size_t)(n + NB) * sizeof(double), 256));
    double *contrib = (double *)((char *)flags + mm_align_up((2 * (2 * (size_t)nblk * (FUSED_MAX_BWB + 1) + 2 * nblk) + 64 + FUSED_MAX_BWB * FUSED_MAX_BWB) * sizeof(int32_t), 256));
    double *contrib_bwd = (double *)((char *)contrib + mm_align_up(2 * (size_t)nblk * (FUSED_MAX_BWB + 1) * NB * sizeof(double), 256));
    double *lpub = (double *)((char *)contrib_bwd + mm_align_up(2 * (size_t)nblk * (FUSED_MAX_BWB + 1) * NB * sizeof(double), 256));
    *out = mm_chol_init_args{info, flags, 1 + 2 * (2 * (size_t)nblk * (bwb + 1) + 2 * nblk) + (size_t)g.m * g.m,
                             (unsigned long long *)contrib_bwd, (size_t)sides * nblk * (bwb + 1) * NB, (unsigned long long *)lpub,
                             chol_nlpub(nblk, bwb), (unsigned long long *)Linv, (size_t)nblk};
    if (sides_out) *sides_out = sides;
    if (bwb_out) *bwb_out = bwb;
    return true;
}
void mm_chol_init_done(mm_ctx *ctx, const void *ws, int sides) {
    ctx->chol_init_done = ws;
    ctx->chol_init_sides = sides;
}

// ---- batched factorisation + substitutions (mm_ba_trf_batched, trf.hip) --------------------------------------------------------
// The geometry and workspace layout mm_chol_solve_sym(both triangles) would use for this problem alone; fails (the caller
// then solves the batch one problem at a time) when that would not be the single-launch path.
int mm_batch_chol_setup(mm_ctx *ctx, mm_batch_prob *bp, void *ws, size_t ws_bytes) {
    const int n = (int)bp->nc, hb = bp->half_bw;
    if (!ws || ws_bytes < mm_chol_workspace_bytes(n) || (n & 1)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_ba_trf_batched: Cholesky workspace");
    const int nblk = (n + NB - 1) / NB;
    long bwb_l = ((long)hb + NB - 1) / NB;
    const int bwb = bwb_l > nblk ? nblk : (int)bwb_l;
    const bool fused = chol_fused_mode() > 0 && nblk >= 2 && bwb >= 1 && bwb <= FUSED_MAX_BWB && (long)NB * nblk * n < (1L << 31) &&
                       !ctx->chol_avoid_fused;
    if (!fused) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_trf_batched: a reduced system outside the single-launch factorisation");
    TwGeom g = {n, nblk, bwb, nblk, 0, 0, nblk * NB - n};
    if (chol_twisted_enabled() && nblk - bwb >= 4) {
        g.m = bwb;
        g.a = (nblk - g.m + 1) / 2;
        g.b = nblk - g.m - g.a;
    }
    const int G_side = (bwb + 1) + bwb * (bwb - 1) / 2;
    const int sides = g.b > 0 ? 2 : 1;
    bp->g_chol = (uint32_t)(g.b > 0 ? 2 * G_side + g.m * (g.m + 1) / 2 : G_side);
    bp->g_bwd = (uint32_t)(sides * (bwb >= 2 ? bwb : 1));
    bp->chol_n = g.n; bp->chol_nblk = g.nblk; bp->chol_bwb = g.bwb; bp->chol_a = g.a; bp->chol_m = g.m; bp->chol_b = g.b; bp->chol_pad = g.pad;
    double *Linv = (double *)ws;
    double *ytmp = (double *)((char *)ws + mm_align_up((size_t)nblk * NB * NB * sizeof(double), 256));
    int32_t *flags = (int32_t *)((char *)ytmp + mm_align_up((size_t)(n + NB) * sizeof(double), 256));
    double *contrib = (double *)((char *)flags + mm_align_up((2 * (2 * (size_t)nblk * (FUSED_MAX_BWB + 1) + 2 * nblk) + 64 + FUSED_MAX_BWB * FUSED_MAX_BWB) * sizeof(int32_t), 256));
    double *contrib_bwd = (double *)((char *)contrib + mm_align_up(2 * (size_t)nblk * (FUSED_MAX_BWB + 1) * NB * sizeof(double), 256));
    double *lpub = (double *)((char *)contrib_bwd + mm_align_up(2 * (size_t)nblk * (FUSED_MAX_BWB + 1) * NB * sizeof(double), 256));
    bp->chol_Linv = Linv;
    bp->chol_ytmp = ytmp;
    bp->chol_flags = flags;
    bp->chol_contrib = contrib;
    bp->chol_contrib_bwd = contrib_bwd;
    bp->chol_lpub = lpub;
    bp->chol_spub = lpub + (size_t)nblk * LPUB_BLOCK;
    bp->chol_nflags = 1 + 2 * (2 * (size_t)nblk * (bwb + 1) + 2 * nblk) + (size_t)g.m * g.m;
    bp->chol_nsent = (size_t)sides * nblk * (bwb + 1) * NB;
    bp->chol_nlpub = chol_nlpub(nblk, bwb);
    return MM_OK;
}
int mm_batch_chol(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g_chol, unsigned max_g_bwd) {
    if (n_list <= 0) return MM_OK;
    if (!ctx->attr_chol_batch) {
        MM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(chol_band_fused_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)FUSED_LDS_BYTES));
        MM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(chol_band_bwd_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)BWD_LDS_BYTES));
        ctx->attr_chol_batch = true;
    }
    const unsigned nl = (unsigned)n_list;
    MM_LAUNCH(ctx, "chol_init_kernel", chol_init_batch_kernel, dim3(CHOL_INIT_GRID, nl), dim3(256), 0, tab, list);
    MM_LAUNCH(ctx, "chol_band_fused_kernel", chol_band_fused_batch_kernel, dim3(max_g_chol, nl), dim3(256), FUSED_LDS_BYTES, tab, list);
    MM_LAUNCH(ctx, "chol_band_bwd_kernel", chol_band_bwd_batch_kernel, dim3(max_g_bwd, nl), dim3(256), BWD_LDS_BYTES, tab, list);
    ctx->chol_last_path = 1;
    return MM_OK;
}
