// Reduced camera system (Schur complement of the point blocks), gfx950, f64.
//
//   S = blockdiag(Bd) - sum_p E_p Cd_p^-1 E_p^T ,   v = gc - sum_p E_p Cd_p^-1 gp_p ,   E_o = Jc_o^T Jp_o (6x3)
//
// One workgroup owns one block ROW of S (camera i) x one window of SR_WIN column blocks.  It walks the observations of
// camera i (CSR by camera); for each (point p) it forms Y = E_o Cd_p^-1 and then visits the other observations of p
// (CSR by point, contiguous) whose camera lies in the window, re-evaluates their E and accumulates -Y E_2^T into a
// 6 x (6*SR_WIN) accumulator held in LDS (f64 LDS atomics, ds_add_f64).  The finished row strip is written with plain
// coalesced stores: S needs no zero fill, no global atomics and both triangles come out filled.  Every (o, o2) pair is
// evaluated from both of its rows — 2x the flops of a symmetric scheme, but the flops are free here: the kernel is bound
// by the LDS atomic rate, and the global-atomic version it replaces ran 16x slower (profiles/r01_*).
#include "ba_eval.h"
#include "chol_init.h"

namespace {

constexpr int SR_WIN = 256;  // cameras per column window: 6 x 1536 doubles = 73,728 B of LDS -> 2 workgroups per CU

// Cinv = Cd^-1 (3x3 symmetric, upper triangle storage xx,xy,xz,yy,yz,zz)
__global__ __launch_bounds__(256) void point_inverse_kernel(int P, const double *__restrict__ Cd,
                                                            double *__restrict__ Cinv) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const double *c = Cd + (size_t)p * 6;
    const double a = c[0], b = c[1], d = c[2], e = c[3], f = c[4], g = c[5];
    const double m00 = e * g - f * f, m01 = d * f - b * g, m02 = b * f - d * e;
    const double det = a * m00 + b * m01 + d * m02;
    const double id = 1.0 / det;
    double *o = Cinv + (size_t)p * 6;
    o[0] = m00 * id;
    o[1] = m01 * id;
    o[2] = m02 * id;
    o[3] = (a * g - d * d) * id;
    o[4] = (b * d - a * f) * id;
    o[5] = (a * e - b * b) * id;
}

__global__ __launch_bounds__(256) void schur_rows_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                         const double *__restrict__ pts, const double *__restrict__ Bd,
                                                         const double *__restrict__ Cinv, const double *__restrict__ gc,
                                                         const double *__restrict__ gp, double *__restrict__ S,
                                                         double *__restrict__ v) {
    __shared__ double acc[6][SR_WIN * 6];
    __shared__ double Ks[9];
    __shared__ double cs[6];
    __shared__ double sm[4];
    const int i = blockIdx.x;
    const int w0 = blockIdx.y * SR_WIN;
    const int wn = min(SR_WIN, pb.F - w0);
    const size_t n = (size_t)pb.F * 6;
    for (int e = threadIdx.x; e < 6 * SR_WIN * 6; e += 256) (&acc[0][0])[e] = 0.0;
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    if (threadIdx.x < 6) cs[threadIdx.x] = cams[(size_t)i * 6 + threadIdx.x];
    __syncthreads();
    double vg[6] = {0, 0, 0, 0, 0, 0};
    for (int e = pb.cam_ptr[i] + threadIdx.x; e < pb.cam_ptr[i + 1]; e += 256) {
        const int o = pb.cam_obs[e];
        const int p = pb.pi[o];
        const double *Xp = pts + (size_t)p * 3;
        Proj pr;
        ba_eval<true, true>(cs, Xp, Ks, pb.obs[2 * (size_t)o], pb.obs[2 * (size_t)o + 1], pr);
        const double *ci = Cinv + (size_t)p * 6;
        const double q00 = ci[0], q01 = ci[1], q02 = ci[2], q11 = ci[3], q12 = ci[4], q22 = ci[5];
        double Y[6][3];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double e0 = pr.Jc[0][a] * pr.Jp[0][0] + pr.Jc[1][a] * pr.Jp[1][0];
            const double e1 = pr.Jc[0][a] * pr.Jp[0][1] + pr.Jc[1][a] * pr.Jp[1][1];
            const double e2 = pr.Jc[0][a] * pr.Jp[0][2] + pr.Jc[1][a] * pr.Jp[1][2];
            Y[a][0] = e0 * q00 + e1 * q01 + e2 * q02;
            Y[a][1] = e0 * q01 + e1 * q11 + e2 * q12;
            Y[a][2] = e0 * q02 + e1 * q12 + e2 * q22;
        }
        if (blockIdx.y == 0) {
            const double g0 = gp[(size_t)p * 3], g1 = gp[(size_t)p * 3 + 1], g2 = gp[(size_t)p * 3 + 2];
#pragma unroll
            for (int a = 0; a < 6; ++a) vg[a] += Y[a][0] * g0 + Y[a][1] * g1 + Y[a][2] * g2;
        }
        for (int e2i = pb.pt_ptr[p]; e2i < pb.pt_ptr[p + 1]; ++e2i) {
            const int o2 = pb.pt_obs[e2i];
            const int f2 = pb.fi[o2];
            if (f2 < w0 || f2 >= w0 + wn) continue;
            Proj p2;
            ba_eval<true, true>(cams + (size_t)f2 * 6, Xp, Ks, pb.obs[2 * (size_t)o2], pb.obs[2 * (size_t)o2 + 1], p2);
            double *dst = &acc[0][(f2 - w0) * 6];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const double e0 = p2.Jc[0][b] * p2.Jp[0][0] + p2.Jc[1][b] * p2.Jp[1][0];
                const double e1 = p2.Jc[0][b] * p2.Jp[0][1] + p2.Jc[1][b] * p2.Jp[1][1];
                const double e2 = p2.Jc[0][b] * p2.Jp[0][2] + p2.Jc[1][b] * p2.Jp[1][2];
#pragma unroll
                for (int a = 0; a < 6; ++a)
                    atomicAdd(dst + a * (SR_WIN * 6) + b, Y[a][0] * e0 + Y[a][1] * e1 + Y[a][2] * e2);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 6 * wn * 6; e += 256) {
        const int a = e / (wn * 6), c = e % (wn * 6);
        const int col = w0 * 6 + c;
        double val = -acc[a][c];
        if (col / 6 == i) val += Bd[(size_t)i * 36 + a * 6 + (col % 6)];
        S[((size_t)i * 6 + a) * n + col] = val;
    }
    if (blockIdx.y == 0) {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double s = block_sum<256>(vg[a], sm);
            if (threadIdx.x == 0) v[(size_t)i * 6 + a] = gc[(size_t)i * 6 + a] - s;
        }
    }
}

// Banded, deterministic variant driven by the co-observation pair list (mm_ba_build_pairs).  One WAVE per non-empty
// block segment (i, f2 = i - d): lanes stride over the segment's (o, o2) pairs, accumulate Y_o E_o2^T in 36 private
// registers (+6 for the right-hand side on the self pairs) and a fixed shuffle tree adds the lanes, so every entry of
// S is summed in the same order on every run — the trust-region iteration, which is chaotic on outlier-laden matches,
// then repeats bit for bit.  No atomics; every block is written together with its mirror image, so both triangles of the
// band are filled (S is zero-filled first).
// waves (= chunks) per workgroup of the pair kernel.  ONE since the second half of round 4: the workgroup barrier behind the
// descriptor / camera-table loads made every wave wait for the slowest of its workgroup -- 259.7 / 230.7 / 205.6 / 198.4 us
// with 8 / 4 / 2 / 1 waves at the bench shape (512 pairs per chunk)
#ifndef MM_SP_WAVES
#define MM_SP_WAVES 1
#endif
constexpr int SP_WAVES = MM_SP_WAVES;
constexpr int MAX_SLABS = 64;
// slab bookkeeping of the overlapped build + solve (all null / 0 when nobody consumes S concurrently)
struct SlabSync {
    int32_t *ready;   // [n_slabs] flags polled by the consumer
    int32_t *done;    // [n_slabs] finished-segment counters
    int32_t seg_count[MAX_SLABS];
    int cams_per_slab;
};
// Per-camera table for the pair kernel: parameters + rotation coefficients (11 doubles per camera), computed once
// per build instead of a sincos / sqrt / four divisions in every chunk's prologue.
constexpr int CAMTAB = 11;
// per-chunk descriptors of the lean pair kernel (struct ChunkDesc below): 8 int32 per chunk
__device__ __forceinline__ void write_chunk_desc(const mm_ba_problem &pb, int32_t *__restrict__ desc, int64_t first, int64_t stride) {
    for (int64_t k = first; k < pb.n_chunks; k += stride) {
        const int sidx = pb.chunk_seg[k], seg = pb.seg_ids[sidx];
        const int ci = seg / (pb.cam_span + 1), cdd = seg % (pb.cam_span + 1);
        const int c_first = pb.seg_chunk_ptr[sidx];
        int32_t *o = desc + 8 * k;
        o[0] = ci;
        o[1] = ci - cdd;
        o[2] = pb.chunk_begin[k];
        o[3] = pb.chunk_end[k];
        o[4] = sidx;
        o[5] = c_first;
        o[6] = pb.seg_chunk_ptr[sidx + 1] - c_first;
        o[7] = 0;
    }
}
__global__ __launch_bounds__(64) void cam_table_kernel(int F, const double *__restrict__ cams, double *__restrict__ tab,
                                                       int32_t *__restrict__ seg_done, int64_t n_seg, const double *__restrict__ K_lean,
                                                       mm_ba_problem pb, int32_t *__restrict__ desc) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    // (also clears the finished-chunk counters of the build that follows: one fill launch less)
    for (int64_t k = f; k < n_seg; k += (int64_t)gridDim.x * 64) seg_done[k] = 0;
    if (desc) write_chunk_desc(pb, desc, f, (int64_t)gridDim.x * 64);
    if (f >= F) return;
    const double *c = cams + (size_t)f * 6;
    if (K_lean) {      // the lean pair kernel's table (P, q, J_r: 24 doubles per camera)
        cam_table2_row(c, K_lean, tab + (size_t)f * CAMTAB2);
        return;
    }
    const CamCoef k = cam_coef_of(c);
    double *t = tab + (size_t)f * CAMTAB;
    for (int q = 0; q < 6; ++q) t[q] = c[q];
    t[6] = k.c;
    t[7] = k.a;
    t[8] = k.b;
    t[9] = k.a1;
    t[10] = k.b1;
}

__device__ __forceinline__ CamVals cam_vals_from_table(const double *__restrict__ t) {
    CamVals v;
    v.rx = uniform_f64(t[0]);
    v.ry = uniform_f64(t[1]);
    v.rz = uniform_f64(t[2]);
    v.tx = uniform_f64(t[3]);
    v.ty = uniform_f64(t[4]);
    v.tz = uniform_f64(t[5]);
    v.k.c = uniform_f64(t[6]);
    v.k.a = uniform_f64(t[7]);
    v.k.b = uniform_f64(t[8]);
    v.k.a1 = uniform_f64(t[9]);
    v.k.b1 = uniform_f64(t[10]);
    return v;
}

// Sum over the 64 lanes of a wave of 42 per-lane values, in two rounds of 21 through a [21][65] LDS slab of the wave
// (row stride 65: the 21 readers hit 21 different banks): every lane writes its values (conflict-free rows), lane q < 21
// then adds row q in a FIXED order (four interleaved partial sums).  ~170 LDS operations per lane instead of the
// 6 x 42 x 2 cross-lane shuffles of a butterfly, and the totals end up spread over lanes 0..20 -- the block of S is
// then written by 21 lanes at once instead of one.  out0 = total of value lane, out1 = total of value 21 + lane.
constexpr int RED_LD = 65;
__device__ __forceinline__ void wave_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void wave_reduce_42(const double (&acc)[42], double *slab /*[21][65]*/, int lane, double &out0,
                                               double &out1) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int q = 0; q < 21; ++q) slab[q * RED_LD + lane] = acc[21 * h + q];
        wave_lds_barrier();
        double s = 0.0;
        if (lane < 21) {
            const double *row = slab + lane * RED_LD;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                s0 += row[j];
                s1 += row[j + 1];
                s2 += row[j + 2];
                s3 += row[j + 3];
            }
            s = (s0 + s1) + (s2 + s3);
        }
        if (h == 0)
            out0 = s;
        else
            out1 = s;
        wave_lds_barrier();
    }
}

// The round-3 formulation (MM_SCHUR_PAIRS=ref): both observations of a pair go through the sweeps' full evaluator
// (projection, residual, both Jacobians from the Rodrigues coefficients: ~200 f64 instructions each).  Kept as the
// cross-check of the lean kernel below.
// launch bound 2 waves per SIMD (no scratch: the cameras' values live in scalar registers)
__global__ __launch_bounds__(64 * SP_WAVES, 2) void schur_pairs_ref_kernel(mm_ba_problem pb, const double *__restrict__ camtab,
                                                                    const double *__restrict__ pts,
                                                                    const double *__restrict__ Cinv,
                                                                    const double *__restrict__ gp,
                                                                    double *partial, const double *__restrict__ Bd,
                                                                    const double *__restrict__ gc, double *S, double *v,
                                                                    int32_t *seg_done, SlabSync slabs, unsigned wg_begin,
                                                                    unsigned wg_total) {
    __shared__ double Ks[9];
    __shared__ double red[SP_WAVES][21 * RED_LD];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // chunks are ordered by camera; the workgroups take them alternately from the front and from the back, because the
    // two-ended factorisation that may be consuming S concurrently starts at both ends and needs the middle last
    // (a build may be cut into several launches: this one covers the positions wg_begin .. of that order)
    const unsigned gpos = blockIdx.x + wg_begin;
    const unsigned wg = (gpos & 1) ? wg_total - 1 - (gpos >> 1) : (gpos >> 1);
    const int64_t c = (int64_t)wg * SP_WAVES + (threadIdx.x >> 6);
    if (c >= pb.n_chunks) return;  // wave-uniform; no workgroup barriers below
    const int sidx = __builtin_amdgcn_readfirstlane(pb.chunk_seg[c]);
    const int seg = __builtin_amdgcn_readfirstlane(pb.seg_ids[sidx]);
    const int i = seg / (pb.cam_span + 1), d = seg % (pb.cam_span + 1);
    const int f2 = i - d;
    // two cameras per chunk, not per pair: parameters and rotation coefficients as wave-uniform scalars
    const CamVals cv_i = cam_vals_from_table(camtab + (size_t)i * CAMTAB), cv_2 = cam_vals_from_table(camtab + (size_t)f2 * CAMTAB);
    double acc[42];
#pragma unroll
    for (int q = 0; q < 42; ++q) acc[q] = 0.0;
    const int e_end = __builtin_amdgcn_readfirstlane(pb.chunk_end[c]);
    const int e_begin = __builtin_amdgcn_readfirstlane(pb.chunk_begin[c]);
    // A chunk holds at most 256 pairs = four per lane: all their indices are requested up front (one memory round trip
    // instead of one -- two without pair_p -- per trip of the loop), so that a trip only waits for its own data gathers
    constexpr int MAXT = 4;
    int oi[MAXT], o2i[MAXT], pidx[MAXT];
#pragma unroll
    for (int k = 0; k < MAXT; ++k) {
        const int e = e_begin + lane + 64 * k;
        const bool in = e < e_end;
        oi[k] = in ? pb.pair_o[e] : -1;
        o2i[k] = in ? pb.pair_o2[e] : -1;
        pidx[k] = in ? (pb.pair_p ? pb.pair_p[e] : -2) : -1;
    }
    auto pair_body = [&](int o, int o2, int p) {
        const double *Xp = pts + (size_t)p * 3;
        Proj pr;
        ba_eval_vals<true, true>(cv_i, Xp, Ks, pb.obs[2 * (size_t)o], pb.obs[2 * (size_t)o + 1], pr);
        const double *ci = Cinv + (size_t)p * 6;
        const double q00 = ci[0], q01 = ci[1], q02 = ci[2], q11 = ci[3], q12 = ci[4], q22 = ci[5];
        // E_o C^-1 E_o2^T = Jc_o^T (Jp_o C^-1 Jp_o2^T) Jc_o2: the middle factor is 2 x 2, so the 6 x 6 block costs
        // 18 + 12 + 24 + 72 multiply-adds through Z = Jp_o C^-1 (2 x 3), M = Z Jp_o2^T (2 x 2), T = M Jc_o2 (2 x 6)
        // instead of 90 + 36 + 108 through the two 6 x 3 products E_o C^-1 and E_o2
        double Z[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            Z[m][0] = pr.Jp[m][0] * q00 + pr.Jp[m][1] * q01 + pr.Jp[m][2] * q02;
            Z[m][1] = pr.Jp[m][0] * q01 + pr.Jp[m][1] * q11 + pr.Jp[m][2] * q12;
            Z[m][2] = pr.Jp[m][0] * q02 + pr.Jp[m][1] * q12 + pr.Jp[m][2] * q22;
        }
        if (o2 == o) {  // self pair: once per observation -> right-hand side
            const double g0 = gp[(size_t)p * 3], g1 = gp[(size_t)p * 3 + 1], g2 = gp[(size_t)p * 3 + 2];
            const double z0 = Z[0][0] * g0 + Z[0][1] * g1 + Z[0][2] * g2, z1 = Z[1][0] * g0 + Z[1][1] * g1 + Z[1][2] * g2;
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[36 + a] += pr.Jc[0][a] * z0 + pr.Jc[1][a] * z1;
        }
        Proj p2;
        ba_eval_vals<true, true>(cv_2, Xp, Ks, pb.obs[2 * (size_t)o2], pb.obs[2 * (size_t)o2 + 1], p2);
        double M[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n_ = 0; n_ < 2; ++n_) M[m][n_] = Z[m][0] * p2.Jp[n_][0] + Z[m][1] * p2.Jp[n_][1] + Z[m][2] * p2.Jp[n_][2];
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const double t0 = M[0][0] * p2.Jc[0][b] + M[0][1] * p2.Jc[1][b], t1 = M[1][0] * p2.Jc[0][b] + M[1][1] * p2.Jc[1][b];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[a * 6 + b] += pr.Jc[0][a] * t0 + pr.Jc[1][a] * t1;
        }
    };
    const int trips = (e_end - e_begin + 63) / 64;   // wave-uniform
    for (int k = 0; k < MAXT; ++k) {                 // (rolled: the body is ~450 f64 instructions)
        if (k >= trips) break;
        int o = oi[0], o2 = o2i[0], p = pidx[0];
#pragma unroll
        for (int q = 1; q < MAXT; ++q)               // select slot k without dynamic register indexing
            if (q == k) {
                o = oi[q];
                o2 = o2i[q];
                p = pidx[q];
            }
        if (o >= 0) {
            if (p == -2) p = pb.pi[o];
            pair_body(o, o2, p);
        }
    }
    for (int e = e_begin + lane + 64 * MAXT; e < e_end; e += 64) {   // chunks longer than 256 pairs (other callers)
        const int o = pb.pair_o[e], o2 = pb.pair_o2[e];
        pair_body(o, o2, pb.pair_p ? pb.pair_p[e] : pb.pi[o]);
    }
    // totals: lane q < 21 holds value q (tot0) and value 21 + q (tot1); values 0..35 = the 6 x 6 block (row-major),
    // 36..41 = the right-hand side rows of the self segment
    double tot0, tot1;
    wave_reduce_42(acc, red[threadIdx.x >> 6], lane, tot0, tot1);
    // ---- finish the segment: the wave that completes its last chunk writes the block of S (and the rhs rows) ----
    // One chunk (the common case: ~160 pairs per segment): straight from the registers.  Several chunks: every wave
    // leaves its partial sums and counts up; the last one adds the partials IN CHUNK ORDER, so the result does not
    // depend on which wave that was.  Everything another workgroup (or the concurrently running factorisation) reads
    // goes through write-through stores / cache-bypassing loads, and a producer drains its stores before it counts.
    const int c_first = pb.seg_chunk_ptr[sidx], n_ch = pb.seg_chunk_ptr[sidx + 1] - c_first;
    const size_t n = (size_t)pb.F * 6;
    bool writer = true;
    if (n_ch > 1) {
        if (lane < 21) {
            __hip_atomic_store(partial + (size_t)c * 42 + lane, tot0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(partial + (size_t)c * 42 + 21 + lane, tot1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        int last = 0;
        if (lane == 0) last = atomicAdd(seg_done + sidx, 1) == n_ch - 1;
        writer = __shfl(last, 0, 64) != 0;
        if (writer && lane < 21) {
            tot0 = 0.0;
            tot1 = 0.0;
            for (int cc = 0; cc < n_ch; ++cc) {
                tot0 += __hip_atomic_load(partial + (size_t)(c_first + cc) * 42 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tot1 += __hip_atomic_load(partial + (size_t)(c_first + cc) * 42 + 21 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (!writer) return;
    if (lane < 21) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = 21 * h + lane;
            const double sum = h == 0 ? tot0 : tot1;
            if (q < 36) {
                double val = -sum;
                if (d == 0) val += Bd[(size_t)i * 36 + q];
                __hip_atomic_store(S + ((size_t)i * 6 + q / 6) * n + (size_t)f2 * 6 + q % 6, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (d != 0)   // the mirror image: the two-ended factorisation eliminates the last cameras in the upper triangle
                    __hip_atomic_store(S + ((size_t)f2 * 6 + q % 6) * n + (size_t)i * 6 + q / 6, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (d == 0) {
                __hip_atomic_store(v + (size_t)i * 6 + (q - 36), gc[(size_t)i * 6 + (q - 36)] - sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (slabs.ready) {  // a concurrent consumer waits for whole camera slabs: count finished segments per slab
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const int sl = i / slabs.cams_per_slab;
            if (atomicAdd(slabs.done + sl, 1) == slabs.seg_count[sl] - 1)
                __hip_atomic_store(slabs.ready + sl, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


// ---- the lean pair kernel (round 4) ---------------------------------------------------------------------------------------
// S needs the two Jacobians of an observation, not its residual, and both follow from per-CAMERA matrices:
//   u = P X + q           P = K R (3 x 3), q = K t                       (9 FMA)
//   Jp[m] = (P[m] - p_m P[2]) / u_2      d proj / d X                    (12)
//   Jt[m] = (K[m] - p_m K[2]) / u_2      d proj / d t                    (12)
//   Jr[m] = (X x Jp[m])^T J_r            d proj / d rvec                 (6 + 9 per row)
// with J_r = a I - b [r]x + e r r^T the right Jacobian of SO(3) (d (R X) / d r = -R [X]x J_r; a = sin th / th,
// b = (1 - cos th) / th^2, e = (1 - a) / th^2).  64 f64 instructions per observation instead of ~200; the observed
// coordinates are not read at all.  Per pair: two of these, Z = Jp_o C^-1 (18), the 2 x 2 middle factor (12), T = M Jc_o2
// (24) and the 72 multiply-adds of the block: ~255 instructions (was ~530).  Pairs of the diagonal segment (o2 == o) reuse
// the first evaluation.  P, q of both cameras are wave-uniform scalars; J_r of both and K sit in the wave's LDS slab and
// arrive as broadcast reads.
struct CamPQ {
    double P[9], q[3];
};
// 1 / x by v_rcp_f64 (about 26 bits) + two Newton steps: within an ulp or two of the quotient; x is a depth in front of a
// camera, far from the denormal / overflow cases the 12-instruction IEEE sequence exists for
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}

struct LeanJ {
    double Jc[2][6];      // d proj / d (rvec, tvec)
    double Jp[2][3];      // d proj / d X
};
// cs: the wave's LDS slab {K[9], Jr[9]} of this camera (broadcast reads)
__device__ __forceinline__ void lean_eval(const CamPQ &c, const double *__restrict__ Ks, const double *__restrict__ Jr,
                                          double X0, double X1, double X2, LeanJ &o) {
    const double u0 = fma(c.P[0], X0, fma(c.P[1], X1, fma(c.P[2], X2, c.q[0])));
    const double u1 = fma(c.P[3], X0, fma(c.P[4], X1, fma(c.P[5], X2, c.q[1])));
    const double u2 = fma(c.P[6], X0, fma(c.P[7], X1, fma(c.P[8], X2, c.q[2])));
    const double iz = fast_rcp(u2);
    const double p[2] = {u0 * iz, u1 * iz};
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o.Jp[m][k] = fma(-p[m], c.P[6 + k], c.P[3 * m + k]) * iz;
            o.Jc[m][3 + k] = fma(-p[m], Ks[6 + k], Ks[3 * m + k]) * iz;
        }
        const double g0 = X1 * o.Jp[m][2] - X2 * o.Jp[m][1];
        const double g1 = X2 * o.Jp[m][0] - X0 * o.Jp[m][2];
        const double g2 = X0 * o.Jp[m][1] - X1 * o.Jp[m][0];
#pragma unroll
        for (int k = 0; k < 3; ++k) o.Jc[m][k] = fma(g0, Jr[k], fma(g1, Jr[3 + k], g2 * Jr[6 + k]));
    }
}

constexpr int SLAB2 = 32;      // doubles per wave beside the reduction slab: Jr_i[9] (+pad to 16), Jr_2[9]
// Per-chunk descriptor, written by schur_prepare_kernel in front of every build: what a wave needs to start on its chunk
// in ONE (scalar) load instead of the chain chunk_seg -> seg_ids -> seg_chunk_ptr.  The kernel is bound by the latency of
// its chain of dependent loads times the waves in flight (two per SIMD), not by arithmetic: halving the instruction count
// (lean evaluator) alone changed nothing; three dependent hops instead of seven did.
struct ChunkDesc {
    int32_t i, f2, e_begin, e_end, sidx, c_first, n_ch, pad;
};
// t uniform.  FORCE = false: plain loads (kernel arguments: the compiler proves the table read-only and emits s_load);
// FORCE = true: the pointers come out of a batch record in memory, nothing is provable, and the values would live in 48
// vector registers -- v_readfirstlane makes them scalars
template <bool FORCE>
__device__ __forceinline__ CamPQ cam_pq_scalar(const double *__restrict__ t) {
    CamPQ v;
#pragma unroll
    for (int k = 0; k < 9; ++k) v.P[k] = FORCE ? uniform_f64(t[k]) : t[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) v.q[k] = FORCE ? uniform_f64(t[9 + k]) : t[9 + k];
    return v;
}
template <bool FORCE_UNIFORM>
__device__ __forceinline__ void schur_pairs_body(const mm_ba_problem pb, const double *__restrict__ camtab,
                                                                    const ChunkDesc *__restrict__ desc,
                                                                    const double *__restrict__ pts,
                                                                    const double *__restrict__ Cinv,
                                                                    const double *__restrict__ gp,
                                                                    double *partial, const double *__restrict__ Bd,
                                                                    const double *__restrict__ gc, double *S, double *v,
                                                                    int32_t *seg_done, SlabSync slabs, unsigned wg_begin,
                                                                    unsigned wg_total, const unsigned bx) {
    __shared__ double Ks[9];
    __shared__ double red[SP_WAVES][21 * RED_LD];
    __shared__ double jrs[SP_WAVES][SLAB2];
    if (threadIdx.x < 9) Ks[threadIdx.x] = pb.K[threadIdx.x];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned gpos = bx + wg_begin;
    const unsigned wg = (gpos & 1) ? wg_total - 1 - (gpos >> 1) : (gpos >> 1);
    const int64_t c = (int64_t)wg * SP_WAVES + wv;
    const bool live = c < pb.n_chunks;
    ChunkDesc cd = {};
    if (live) cd = desc[c];      // (uniform address: one scalar load)
    if (FORCE_UNIFORM) {
        cd.i = __builtin_amdgcn_readfirstlane(cd.i);
        cd.f2 = __builtin_amdgcn_readfirstlane(cd.f2);
        cd.e_begin = __builtin_amdgcn_readfirstlane(cd.e_begin);
        cd.e_end = __builtin_amdgcn_readfirstlane(cd.e_end);
        cd.sidx = __builtin_amdgcn_readfirstlane(cd.sidx);
        cd.c_first = __builtin_amdgcn_readfirstlane(cd.c_first);
        cd.n_ch = __builtin_amdgcn_readfirstlane(cd.n_ch);
    }
    const int i = cd.i, f2 = cd.f2, d = cd.i - cd.f2, sidx = cd.sidx;
    const int e_begin = cd.e_begin, e_end = cd.e_end;
    // second hop, all at once: the two cameras' tables (scalar P, q; J_r through the wave's LDS slab) and the first trip's
    // point index
    auto point_of = [&](int e) { return pb.pair_p ? pb.pair_p[e] : pb.pi[pb.pair_o[e]]; };
    auto self_of = [&](int e) { return d == 0 && pb.pair_o[e] == pb.pair_o2[e]; };
    int e = e_begin + lane;
    int p_next = -1;
    bool self_next = false;
    if (live) {
        if (e < e_end) {
            p_next = point_of(e);
            self_next = self_of(e);
        }
        if (lane < 9) jrs[wv][lane] = camtab[(size_t)i * CAMTAB2 + 12 + lane];
        else if (lane >= 16 && lane < 25) jrs[wv][lane] = camtab[(size_t)f2 * CAMTAB2 + 12 + lane - 16];
    }
    __syncthreads();      // (Ks; the wave's own slab would only need a wave barrier)
    if (!live) return;    // wave-uniform; no workgroup barriers below
    const CamPQ cp_i = cam_pq_scalar<FORCE_UNIFORM>(camtab + (size_t)i * CAMTAB2), cp_2 = cam_pq_scalar<FORCE_UNIFORM>(camtab + (size_t)f2 * CAMTAB2);
    const double *Jr_i = jrs[wv], *Jr_2 = jrs[wv] + 16;
    double acc[42];
#pragma unroll
    for (int q = 0; q < 42; ++q) acc[q] = 0.0;
    // (the point of a pair is all the kernel gathers by: the observed coordinates are not needed)
    // the point, its C^-1 (upper triangle) and -- diagonal segments only -- its gradient block: everything the arithmetic of
    // a trip reads from memory, requested one trip ahead (vector memory returns in order: a load issued and consumed inside
    // the trip would wait for the prefetches issued before it)
    auto load_xc = [&](int p, double (&x)[12]) {
        if (p >= 0) {
            const double *Xp = pts + (size_t)p * 3, *ci = Cinv + (size_t)p * 6;
            x[0] = Xp[0];
            x[1] = Xp[1];
            x[2] = Xp[2];
#pragma unroll
            for (int q = 0; q < 6; ++q) x[3 + q] = ci[q];
            if (d == 0) {
                x[9] = gp[(size_t)p * 3];
                x[10] = gp[(size_t)p * 3 + 1];
                x[11] = gp[(size_t)p * 3 + 2];
            }
        }
    };
    auto pair_body = [&](int p, bool self, const double (&x)[12]) {
        const double X0 = x[0], X1 = x[1], X2 = x[2];
        const double q00 = x[3], q01 = x[4], q02 = x[5], q11 = x[6], q12 = x[7], q22 = x[8];
        LeanJ a;
        lean_eval(cp_i, Ks, Jr_i, X0, X1, X2, a);
        // E_o C^-1 E_o2^T = Jc_o^T (Jp_o C^-1 Jp_o2^T) Jc_o2 through the 2 x 2 middle factor
        double Z[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            Z[m][0] = a.Jp[m][0] * q00 + a.Jp[m][1] * q01 + a.Jp[m][2] * q02;
            Z[m][1] = a.Jp[m][0] * q01 + a.Jp[m][1] * q11 + a.Jp[m][2] * q12;
            Z[m][2] = a.Jp[m][0] * q02 + a.Jp[m][1] * q12 + a.Jp[m][2] * q22;
        }
        double M[2][2], T[2][6];
        if (d == 0) {
            // (wave-uniform) the diagonal segment: both observations see the point from the SAME camera, so their Jacobians
            // are the same whether or not o2 == o (a point observed twice in one frame) -- no second evaluation.  The
            // right-hand side takes one term per observation: the self pairs.
            if (self) {
                const double g0 = x[9], g1 = x[10], g2 = x[11];
                const double z0 = Z[0][0] * g0 + Z[0][1] * g1 + Z[0][2] * g2, z1 = Z[1][0] * g0 + Z[1][1] * g1 + Z[1][2] * g2;
#pragma unroll
                for (int q = 0; q < 6; ++q) acc[36 + q] += a.Jc[0][q] * z0 + a.Jc[1][q] * z1;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n_ = 0; n_ < 2; ++n_) M[m][n_] = Z[m][0] * a.Jp[n_][0] + Z[m][1] * a.Jp[n_][1] + Z[m][2] * a.Jp[n_][2];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                T[0][b] = M[0][0] * a.Jc[0][b] + M[0][1] * a.Jc[1][b];
                T[1][b] = M[1][0] * a.Jc[0][b] + M[1][1] * a.Jc[1][b];
            }
        } else {
            LeanJ b2;
            lean_eval(cp_2, Ks, Jr_2, X0, X1, X2, b2);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n_ = 0; n_ < 2; ++n_) M[m][n_] = Z[m][0] * b2.Jp[n_][0] + Z[m][1] * b2.Jp[n_][1] + Z[m][2] * b2.Jp[n_][2];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                T[0][b] = M[0][0] * b2.Jc[0][b] + M[0][1] * b2.Jc[1][b];
                T[1][b] = M[1][0] * b2.Jc[0][b] + M[1][1] * b2.Jc[1][b];
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int b = 0; b < 6; ++b) acc[q * 6 + b] += a.Jc[0][q] * T[0][b] + a.Jc[1][q] * T[1][b];
    };
    // software pipeline over the trips of the chunk, two deep: the point index of trip t + 2 and the point / C^-1 of trip
    // t + 1 are requested before the arithmetic of trip t starts, so that a trip waits for neither of its two dependent
    // gathers (the kernel is bound by exactly those latencies at two waves per SIMD)
    double xc[12];
    load_xc(p_next, xc);
    int p_cur = p_next;
    bool self_cur = self_next;
    {
        const int en = e + 64;
        p_next = en < e_end ? point_of(en) : -1;
        self_next = en < e_end && self_of(en);
    }
#pragma clang loop unroll(disable)
    for (; e - lane < e_end; e += 64) {      // (wave-uniform trip count)
        double xn[12];
        load_xc(p_next, xn);
        const int p_nn_e = e + 128;
        const int p_nn = p_nn_e < e_end ? point_of(p_nn_e) : -1;
        const bool self_nn = p_nn_e < e_end && self_of(p_nn_e);
        if (p_cur >= 0) pair_body(p_cur, self_cur, xc);
#pragma unroll
        for (int q = 0; q < 12; ++q) xc[q] = xn[q];
        p_cur = p_next;
        self_cur = self_next;
        p_next = p_nn;
        self_next = self_nn;
    }
    double tot0, tot1;
    wave_reduce_42(acc, red[wv], lane, tot0, tot1);
    // ---- finish the segment (as in the reference formulation above) ----
    const int c_first = cd.c_first, n_ch = cd.n_ch;
    const size_t n = (size_t)pb.F * 6;
    bool writer = true;
    if (n_ch > 1) {
        if (lane < 21) {
            __hip_atomic_store(partial + (size_t)c * 42 + lane, tot0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(partial + (size_t)c * 42 + 21 + lane, tot1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        int last = 0;
        if (lane == 0) last = atomicAdd(seg_done + sidx, 1) == n_ch - 1;
        writer = __shfl(last, 0, 64) != 0;
        if (writer && lane < 21) {
            tot0 = 0.0;
            tot1 = 0.0;
            for (int cc = 0; cc < n_ch; ++cc) {
                tot0 += __hip_atomic_load(partial + (size_t)(c_first + cc) * 42 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tot1 += __hip_atomic_load(partial + (size_t)(c_first + cc) * 42 + 21 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (!writer) return;
    // (Where the time of this kernel goes, by ablation on the tools/bench_schur.py shape, 225 us in all: the bare skeleton --
    // one wave per chunk, descriptor -> tables / point index -> point / C^-1 -- 108 us; arithmetic +45; the 42-value
    // reduction +34; these stores +38 as write-through agent-scope stores.  Nobody reads S before the launch ends unless a
    // concurrent factorisation consumes it slab by slab: plain stores otherwise.)
    const bool through = slabs.ready != nullptr;
    if (lane < 21) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = 21 * h + lane;
            const double sum = h == 0 ? tot0 : tot1;
            if (q < 36) {
                double val = -sum;
                if (d == 0) val += Bd[(size_t)i * 36 + q];
                double *dst = S + ((size_t)i * 6 + q / 6) * n + (size_t)f2 * 6 + q % 6;
                double *mir = S + ((size_t)f2 * 6 + q % 6) * n + (size_t)i * 6 + q / 6;
                if (through) {
                    __hip_atomic_store(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (d != 0) __hip_atomic_store(mir, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    *dst = val;
                    if (d != 0) *mir = val;
                }
            } else if (d == 0) {
                double *dst = v + (size_t)i * 6 + (q - 36);
                const double val = gc[(size_t)i * 6 + (q - 36)] - sum;
                if (through) __hip_atomic_store(dst, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *dst = val;
            }
        }
    }
    if (slabs.ready) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const int sl = i / slabs.cams_per_slab;
            if (atomicAdd(slabs.done + sl, 1) == slabs.seg_count[sl] - 1)
                __hip_atomic_store(slabs.ready + sl, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <int OCC>      // waves per SIMD the register allocation aims at (2; 3 spills into scratch and is 1.8x slower: measured)
__global__ __launch_bounds__(64 * SP_WAVES, OCC) void schur_pairs_kernel(mm_ba_problem pb, const double *__restrict__ camtab,
                                                                    const ChunkDesc *__restrict__ desc,
                                                                    const double *__restrict__ pts,
                                                                    const double *__restrict__ Cinv,
                                                                    const double *__restrict__ gp,
                                                                    double *partial, const double *__restrict__ Bd,
                                                                    const double *__restrict__ gc, double *S, double *v,
                                                                    int32_t *seg_done, SlabSync slabs, unsigned wg_begin,
                                                                    unsigned wg_total) {
    schur_pairs_body<false>(pb, camtab, desc, pts, Cinv, gp, partial, Bd, gc, S, v, seg_done, slabs, wg_begin, wg_total, blockIdx.x);
}
// batched (mm_ba_trf_batched): blockIdx.y picks the problem
__global__ __launch_bounds__(256) void zero_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    const mm_batch_prob &bp = tab[list[blockIdx.y]];
    if (blockIdx.x >= bp.g_zero) return;
    double2 *S2 = reinterpret_cast<double2 *>(bp.S);      // (nc is even: nc * nc doubles = nc * nc / 2 pairs)
    const int64_t n2 = bp.nc * bp.nc / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)bp.g_zero * 256) S2[i] = make_double2(0.0, 0.0);
}
__global__ __launch_bounds__(64 * SP_WAVES, 2) void schur_pairs_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    const mm_batch_prob &bp = tab[list[blockIdx.y]];
    if (blockIdx.x >= bp.g_pairs) return;
    SlabSync none = {};
    schur_pairs_body<true>(bp.pb, bp.schur_camtab, (const ChunkDesc *)bp.schur_desc, bp.x + bp.nc, bp.Cinv, bp.g + bp.nc, bp.schur_partial, bp.Bd,
                     bp.g, bp.S, bp.v, bp.schur_seg_done, none, 0u, bp.g_pairs, blockIdx.x);
}



// (Round 4, tried and dropped: a PERSISTENT variant of this kernel in which a wave walks a sequence of chunks and requests the
// next chunks' descriptors, camera tables, point indices and first point / C^-1 while it works on the current one.  It
// needs ~40 more registers than the 218 used here; with the 256 that two waves per SIMD leave, the compiler spills ~45
// values around every chunk and the kernel ran at 320-350 us against 222 us for this one (tools/bench_schur.py shape).
// Three waves per SIMD (168 registers, 55 values in scratch) took 411 us.  The persistent variant rebuilt for ONE wave per
// SIMD (289 registers, no scratch: latency to be hidden by the pipeline instead of by a second wave) took 325-373 us:
// vector memory returns in order, so the first load a trip consumes -- the next index, gp of a self pair -- waits for every
// prefetch issued before it; the second wave is what hides latency here, not issue order.)

// marks a camera slab without any segment as complete (its rows only hold what schur_diag_fill wrote)
__global__ void slab_flag_kernel(int32_t *flags, int s) {
    __hip_atomic_store(flags + s, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// cameras without any observation never appear in a segment: their diagonal block / rhs is just (Bd, gc)
__global__ void schur_diag_fill_kernel(mm_ba_problem pb, const double *__restrict__ Bd, const double *__restrict__ gc,
                                       double *__restrict__ S, double *__restrict__ v) {
    const int i = blockIdx.x;
    if (pb.cam_ptr[i + 1] != pb.cam_ptr[i]) return;
    const size_t n = (size_t)pb.F * 6;
    if (threadIdx.x < 36) S[((size_t)i * 6 + threadIdx.x / 6) * n + (size_t)i * 6 + threadIdx.x % 6] = Bd[(size_t)i * 36 + threadIdx.x];
    if (threadIdx.x < 6) v[(size_t)i * 6 + threadIdx.x] = gc[(size_t)i * 6 + threadIdx.x];
}

// point_inverse + schur_diag_fill + cam_table in ONE launch (the banded build's three small preparation steps: each was a
// launch of a few microseconds in front of every build).  Workgroup b: the points 256 b .., camera b's diagonal block if it
// has no observation, the table rows of cameras 256 b .. (and its share of the finished-chunk counters).
// What the prepare kernel does on behalf of its neighbours when mm_ba_trf hands the whole reduced solve to
// mm_ba_schur_solve_damped: the damping (mm_ba_damp's arithmetic, fma(si^2, reg, .) on the diagonals), the zero fill of
// the band tiles of S, and the fills of the single-launch factorisation.
struct PrepExtra {
    mm_damp_spec dmp;          // dmp.B == nullptr: Bd / Cd come damped from the caller
    int zero_bwb;              // >= 0: zero the 64 x 64 tiles (bi, bj), |bi - bj| <= zero_bwb, of S here
    int chol_init;             // run mm_chol_init_body(init)
    mm_chol_init_args init;
};
__device__ __forceinline__ void schur_prepare_body(mm_ba_problem pb, const double *__restrict__ cams,
                                                            double *__restrict__ Bd, const double *__restrict__ Cd,
                                                            const double *__restrict__ gc, double *__restrict__ Cinv,
                                                            double *__restrict__ S, double *__restrict__ v,
                                                            double *__restrict__ tab, int32_t *__restrict__ seg_done, bool lean,
                                                            int32_t *__restrict__ desc /* [n_chunks][8] or NULL */, const unsigned bx, const unsigned gx,
                                                            const PrepExtra *ex = nullptr) {
    const int b = (int)bx, tid = threadIdx.x;
    for (int64_t k = (int64_t)b * 256 + tid; k < pb.n_seg; k += (int64_t)gx * 256) seg_done[k] = 0;
    if (desc) write_chunk_desc(pb, desc, (int64_t)b * 256 + tid, (int64_t)gx * 256);
    const bool fold = ex && ex->dmp.B;
    double reg = 0.0;
    if (fold) {
        if (ex->dmp.gh2) {
            const double r = trf_damping_value(ex->dmp.gh2[0], ex->dmp.d11[0], ex->dmp.Delta);
            reg = fmax(r, ex->dmp.min_damping);
            if (b == 0 && tid == 0) {
                ex->dmp.damp_out[0] = r;
                ex->dmp.damp_out[1] = reg;
            }
        } else {
            reg = ex->dmp.reg[0];
        }
    }
    if (ex && ex->chol_init) mm_chol_init_body(ex->init, bx, gx);
    const size_t n = (size_t)pb.F * 6;
    if (ex && ex->zero_bwb >= 0) {
        // row r of S: the columns of the tiles within the band, two at a time (n is even); the cameras' own 6 x 6 diagonal
        // blocks are left to the workgroup of the camera (below), which may have to put Bd there
        const int nblk = (int)((n + 63) / 64);
        for (size_t r = bx; r < n; r += gx) {
            const int bi = (int)(r >> 6);
            const int lo = bi - ex->zero_bwb < 0 ? 0 : bi - ex->zero_bwb, hi = bi + ex->zero_bwb >= nblk ? nblk - 1 : bi + ex->zero_bwb;
            const size_t c0 = (size_t)lo * 64, c1 = (size_t)(hi + 1) * 64 < n ? (size_t)(hi + 1) * 64 : n;
            const size_t d0 = r / 6 * 6;
            for (size_t c = c0 + 2 * (size_t)tid; c < c1; c += 512)
                if (c < d0 || c >= d0 + 6) *reinterpret_cast<double2 *>(S + r * n + c) = double2{0.0, 0.0};
        }
    }
    const int f = b * 256 + tid;
    if (f < pb.F) {
        const double *c = cams + (size_t)f * 6;
        if (lean) {
            cam_table2_row(c, pb.K, tab + (size_t)f * CAMTAB2);
        } else {
            const CamCoef k = cam_coef_of(c);
            double *t = tab + (size_t)f * CAMTAB;
            for (int q = 0; q < 6; ++q) t[q] = c[q];
            t[6] = k.c;
            t[7] = k.a;
            t[8] = k.b;
            t[9] = k.a1;
            t[10] = k.b1;
        }
    }
    if (b < pb.F) {
        const bool empty = pb.cam_ptr[b + 1] == pb.cam_ptr[b];      // camera b never appears in a segment: (Bd, gc) as they are
        if (tid < 36) {
            double bd;
            if (fold) {
                bd = ex->dmp.B[(size_t)b * 36 + tid];
                if (tid % 7 == 0) {
                    const double s = ex->dmp.si[(size_t)b * 6 + tid / 7];
                    bd = fma(s * s, reg, bd);
                }
                Bd[(size_t)b * 36 + tid] = bd;
            } else {
                bd = Bd[(size_t)b * 36 + tid];
            }
            if (empty)
                S[((size_t)b * 6 + tid / 6) * n + (size_t)b * 6 + tid % 6] = bd;
            else if (ex && ex->zero_bwb >= 0)
                S[((size_t)b * 6 + tid / 6) * n + (size_t)b * 6 + tid % 6] = 0.0;
        }
        if (empty && tid < 6) v[(size_t)b * 6 + tid] = gc[(size_t)b * 6 + tid];
    }
    const int p = b * 256 + tid;
    if (p < pb.P) {
        double a, bq, d, e, ff, g;
        if (fold) {
            const double *c = ex->dmp.C + (size_t)p * 6;
            const double *sp = ex->dmp.si + n + (size_t)p * 3;
            a = fma(sp[0] * sp[0], reg, c[0]);
            bq = c[1];
            d = c[2];
            e = fma(sp[1] * sp[1], reg, c[3]);
            ff = c[4];
            g = fma(sp[2] * sp[2], reg, c[5]);
        } else {
            const double *c = Cd + (size_t)p * 6;
            a = c[0], bq = c[1], d = c[2], e = c[3], ff = c[4], g = c[5];
        }
        const double m00 = e * g - ff * ff, m01 = d * ff - bq * g, m02 = bq * ff - d * e;
        const double det = a * m00 + bq * m01 + d * m02;
        const double id = 1.0 / det;
        double *o = Cinv + (size_t)p * 6;
        o[0] = m00 * id;
        o[1] = m01 * id;
        o[2] = m02 * id;
        o[3] = (a * g - d * d) * id;
        o[4] = (bq * d - a * ff) * id;
        o[5] = (a * e - bq * bq) * id;
    }
}
__global__ __launch_bounds__(256) void schur_prepare_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                            double *__restrict__ Bd, const double *__restrict__ Cd,
                                                            const double *__restrict__ gc, double *__restrict__ Cinv,
                                                            double *__restrict__ S, double *__restrict__ v,
                                                            double *__restrict__ tab, int32_t *__restrict__ seg_done, bool lean,
                                                            int32_t *__restrict__ desc) {
    schur_prepare_body(pb, cams, Bd, Cd, gc, Cinv, S, v, tab, seg_done, lean, desc, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void schur_prepare_damped_kernel(mm_ba_problem pb, const double *__restrict__ cams,
                                                                   double *__restrict__ Bd, const double *__restrict__ gc,
                                                                   double *__restrict__ Cinv, double *__restrict__ S,
                                                                   double *__restrict__ v, double *__restrict__ tab,
                                                                   int32_t *__restrict__ seg_done, int32_t *__restrict__ desc,
                                                                   PrepExtra ex) {
    schur_prepare_body(pb, cams, Bd, nullptr, gc, Cinv, S, v, tab, seg_done, true, desc, blockIdx.x, gridDim.x, &ex);
}
__global__ __launch_bounds__(256) void schur_prepare_batch_kernel(const mm_batch_prob *__restrict__ tab, const int32_t *__restrict__ list) {
    const mm_batch_prob &bp = tab[list[blockIdx.y]];
    if (blockIdx.x >= bp.g_prep) return;
    schur_prepare_body(bp.pb, bp.x, bp.Bd, bp.Cd, bp.g, bp.Cinv, bp.S, bp.v, bp.schur_camtab, bp.schur_seg_done, true, bp.schur_desc,
                       blockIdx.x, bp.g_prep);
}


// ---- device-side construction of the co-observation pair list ------------------------------------------------------
// cnt[o] = number of observations o2 of the same point with camera(o2) <= camera(o); span = max camera distance.
__global__ __launch_bounds__(256) void pairs_count_kernel(mm_ba_problem pb, int32_t *__restrict__ cnt,
                                                          int32_t *__restrict__ span_out) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int dmax = 0;
    if (o < pb.O) {
        const int f = pb.fi[o], p = pb.pi[o];
        int c = 0;
        for (int e = pb.pt_ptr[p]; e < pb.pt_ptr[p + 1]; ++e) {
            const int d = f - pb.fi[pb.pt_obs[e]];
            c += d >= 0;
            dmax = max(dmax, d);
        }
        cnt[o] = c;
    }
    for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, __shfl_down(dmax, off, 64));
    // (a returning atomic on ONE word per wave is ~11 ns each behind one another -- 22 k waves: 0.25 ms; nearly every wave finds the
    // maximum already there)
    if ((threadIdx.x & 63) == 0 && dmax > __hip_atomic_load(span_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(span_out, dmax);
}

// writes the pairs of observation o at offsets[o]..: key = camera(o) * (span + 1) + camera(o) - camera(o2)
__global__ __launch_bounds__(256) void pairs_emit_kernel(mm_ba_problem pb, const int64_t *__restrict__ offsets, int span,
                                                         int32_t *__restrict__ key, int32_t *__restrict__ po,
                                                         int32_t *__restrict__ po2) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= pb.O) return;
    const int f = pb.fi[o], p = pb.pi[o];
    int64_t w = offsets[o];
    for (int e = pb.pt_ptr[p]; e < pb.pt_ptr[p + 1]; ++e) {
        const int o2 = pb.pt_obs[e];
        const int d = f - pb.fi[o2];
        if (d < 0) continue;
        key[w] = f * (span + 1) + d;
        po[w] = (int32_t)o;
        po2[w] = o2;
        ++w;
    }
}

}  // namespace

extern "C" int mm_ba_pairs_count(mm_ctx *ctx, const mm_ba_problem *pb, int32_t *cnt, int32_t *span_out) {
    if (!ctx) return MM_ERR_ARG;
    if (!pb || !cnt || !span_out || (pb->O > 0 && (!pb->fi || !pb->pi || !pb->pt_ptr || !pb->pt_obs)))
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_pairs_count: bad argument");
    MM_HIP(ctx, hipMemsetAsync(span_out, 0, sizeof(int32_t), ctx->stream));
    if (pb->O == 0) return MM_OK;
    MM_LAUNCH(ctx, "pairs_count_kernel", pairs_count_kernel, dim3((unsigned)((pb->O + 255) / 256)), dim3(256), 0, *pb, cnt,
              span_out);
    return MM_OK;
}

extern "C" int mm_ba_pairs_emit(mm_ctx *ctx, const mm_ba_problem *pb, const int64_t *offsets, int span, int32_t *key,
                                int32_t *pair_o, int32_t *pair_o2) {
    if (!ctx) return MM_ERR_ARG;
    if (!pb || !offsets || !key || !pair_o || !pair_o2 || span < 0) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_pairs_emit: bad argument");
    if (pb->O == 0) return MM_OK;
    if ((int64_t)pb->F * (span + 1) > 0x7fffffffLL) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_pairs_emit: F * (span + 1) overflows the key");
    MM_LAUNCH(ctx, "pairs_emit_kernel", pairs_emit_kernel, dim3((unsigned)((pb->O + 255) / 256)), dim3(256), 0, *pb, offsets,
              span, key, pair_o, pair_o2);
    return MM_OK;
}

extern "C" size_t mm_ba_schur_workspace_bytes(const mm_ba_problem *pb) {
    if (!pb || pb->n_chunks <= 0) return 0;
    return mm_align_up((size_t)pb->n_chunks * 42 * sizeof(double), 256) /* partial sums of multi-chunk segments */ +
           mm_align_up((size_t)pb->n_seg * sizeof(int32_t), 256) /* finished-chunk counters */ +
           mm_align_up(2 * MAX_SLABS * sizeof(int32_t), 256) /* slab flags, slab counters */ +
           mm_align_up((size_t)pb->F * CAMTAB2 * sizeof(double), 256) /* per-camera table */ +
           mm_align_up((size_t)pb->n_chunks * 32, 256) /* per-chunk descriptors */;
}

namespace {
// MM_SCHUR_PAIRS=ref: the round-3 pair kernel (full evaluator per observation, index chain); default: the lean one
bool schur_pairs_lean() {
    static const bool lean = [] {
        const char *e = getenv("MM_SCHUR_PAIRS");
        return !(e && e[0] == 'r');
    }();
    return lean;
}
#define MM_LAUNCH_PAIRS(ctx, grid, pbv, camtab_, desc_, ...)                                                             \
    do {                                                                                                                 \
        if (schur_pairs_lean())                                                                                          \
            MM_LAUNCH(ctx, "schur_pairs_kernel", schur_pairs_kernel<2>, grid, dim3(64 * SP_WAVES), 0, pbv, camtab_,      \
                      (const ChunkDesc *)(desc_), __VA_ARGS__);                                                          \
        else                                                                                                             \
            MM_LAUNCH(ctx, "schur_pairs_ref_kernel", schur_pairs_ref_kernel, grid, dim3(64 * SP_WAVES), 0, pbv, camtab_, \
                      __VA_ARGS__);                                                                                      \
    } while (0)
struct SchurWs {
    double *partial, *camtab;
    int32_t *seg_done, *slab_ready, *slab_done, *desc;
};
SchurWs carve_schur_ws(const mm_ba_problem *pb, void *ws) {
    SchurWs w;
    char *p = (char *)ws;
    w.partial = (double *)p;
    p += mm_align_up((size_t)pb->n_chunks * 42 * sizeof(double), 256);
    w.seg_done = (int32_t *)p;
    p += mm_align_up((size_t)pb->n_seg * sizeof(int32_t), 256);
    w.slab_ready = (int32_t *)p;
    w.slab_done = w.slab_ready + MAX_SLABS;
    p += mm_align_up(2 * MAX_SLABS * sizeof(int32_t), 256);
    w.camtab = (double *)p;
    p += mm_align_up((size_t)pb->F * CAMTAB2 * sizeof(double), 256);
    w.desc = (int32_t *)p;
    return w;
}
}  // namespace

extern "C" int mm_ba_schur(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts,
                           const double *Bd, const double *Cd, const double *gc, const double *gp, double *S, double *v,
                           double *Cinv, void *ws, size_t ws_bytes) {
    if (!ctx) return MM_ERR_ARG;
    if (!pb || pb->F < 0 || pb->P < 0 || pb->O < 0 || !pb->K) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_schur: bad problem");
    if (!cams || !pts || !Bd || !Cd || !gc || !gp || !S || !v || !Cinv || !pb->pt_ptr || !pb->cam_ptr ||
        (pb->O > 0 && (!pb->pt_obs || !pb->cam_obs || !pb->fi || !pb->pi || !pb->obs)))
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_schur: null pointer");
    if (pb->F == 0) return MM_OK;
    if (pb->n_seg > 0 && pb->n_chunks > 0 && pb->seg_ids && pb->seg_chunk_ptr && pb->chunk_seg && pb->chunk_begin &&
        pb->chunk_end && pb->pair_o && pb->pair_o2) {
        if (!ws || ws_bytes < mm_ba_schur_workspace_bytes(pb)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_ba_schur: workspace too small");
        // banded + deterministic: zero S (the Cholesky touches whole 64-blocks of the band), then the lower blocks
        MM_HIP(ctx, hipMemsetAsync(S, 0, (size_t)pb->F * 6 * pb->F * 6 * sizeof(double), ctx->stream));
        const int64_t wgs = (pb->n_chunks + SP_WAVES - 1) / SP_WAVES;
        const SchurWs w = carve_schur_ws(pb, ws);
        SlabSync none = {};
        const int prep = (pb->P + 255) / 256 > pb->F ? (pb->P + 255) / 256 : pb->F;
        MM_LAUNCH(ctx, "schur_prepare_kernel", schur_prepare_kernel, dim3(prep), dim3(256), 0, *pb, cams, const_cast<double *>(Bd), Cd, gc,
                  Cinv, S, v, w.camtab, w.seg_done, schur_pairs_lean(), schur_pairs_lean() ? w.desc : (int32_t *)nullptr);
        MM_LAUNCH_PAIRS(ctx, dim3((unsigned)wgs), *pb, (const double *)w.camtab, w.desc, pts, (const double *)Cinv, gp, w.partial, Bd, gc, S, v,
                        w.seg_done, none, 0u, (unsigned)wgs);
        return MM_OK;
    }
    if (pb->P > 0)
        MM_LAUNCH(ctx, "point_inverse_kernel", point_inverse_kernel, dim3((pb->P + 255) / 256), dim3(256), 0, pb->P, Cd,
                  Cinv);
    const int nwin = (pb->F + SR_WIN - 1) / SR_WIN;
    MM_LAUNCH(ctx, "schur_rows_kernel", schur_rows_kernel, dim3(pb->F, nwin), dim3(256), 0, *pb, cams, pts, Bd, Cinv, gc,
              gp, S, v);
    return MM_OK;
}

// Reduced camera system + its solution, overlapped: S is built in camera slabs (ascending) on the context's stream
// while the single-launch banded Cholesky, started first on a second stream, consumes block rows as their slabs are
// flagged complete.  The factorisation is a chain of dependent block columns that keeps ~46 CUs busy; building S keeps
// the other ~210 busy; one after the other they cost 0.52 + 1.05 ms at C3, overlapped about the longer of the two.
// Falls back to mm_ba_schur + mm_chol_solve when the problem has no pair list / slabs or the band is too wide.
extern "C" int mm_ba_schur_solve(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts,
                                 const double *Bd, const double *Cd, const double *gc, const double *gp, double *S,
                                 double *v, double *Cinv, int half_bandwidth, int32_t *info, void *ws_schur,
                                 size_t ws_schur_bytes, void *ws_chol, size_t ws_chol_bytes, int n_slabs,
                                 int cams_per_slab, const int64_t *slab_seg_ptr, const int64_t *slab_chunk_ptr) {
    if (!ctx) return MM_ERR_ARG;
    if (!pb || !info) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_schur_solve: bad argument");
    const int n = pb->F * 6;
    const bool pairs_ok = pb->n_seg > 0 && pb->n_chunks > 0 && pb->seg_ids && pb->seg_chunk_ptr && pb->chunk_seg &&
                          pb->chunk_begin && pb->chunk_end && pb->pair_o && pb->pair_o2;
    const bool overlap = pairs_ok && n_slabs >= 2 && n_slabs <= MAX_SLABS && cams_per_slab >= 16 && slab_seg_ptr &&
                         slab_chunk_ptr && mm_chol_fused_eligible(n, half_bandwidth) && pb->P > 0;
    if (!overlap) {
        int rc = mm_ba_schur(ctx, pb, cams, pts, Bd, Cd, gc, gp, S, v, Cinv, ws_schur, ws_schur_bytes);
        if (rc) return rc;
        // (the pair-list build fills both triangles of the band; the general kernel all of S)
        return mm_chol_solve_sym(ctx, S, n, v, half_bandwidth, 1, info, ws_chol, ws_chol_bytes);
    }
    if (!cams || !pts || !Bd || !Cd || !gc || !gp || !S || !v || !Cinv) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_schur_solve: null pointer");
    if (!ws_schur || ws_schur_bytes < mm_ba_schur_workspace_bytes(pb)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_ba_schur_solve: workspace too small");
    if (slab_seg_ptr[0] != 0 || slab_seg_ptr[n_slabs] != pb->n_seg || slab_chunk_ptr[0] != 0 || slab_chunk_ptr[n_slabs] != pb->n_chunks ||
        (int64_t)n_slabs * cams_per_slab < pb->F)
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_schur_solve: slab tables do not cover the problem");
    if (!ctx->aux) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically lowest = highest priority
        MM_HIP(ctx, hipStreamCreateWithPriority(&ctx->aux, hipStreamNonBlocking, hi));
        MM_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        MM_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    const SchurWs w = carve_schur_ws(pb, ws_schur);
    MM_LAUNCH(ctx, "point_inverse_kernel", point_inverse_kernel, dim3((pb->P + 255) / 256), dim3(256), 0, pb->P, Cd, Cinv);
    SlabSync slabs = {};
    slabs.ready = w.slab_ready;
    slabs.done = w.slab_done;
    slabs.cams_per_slab = cams_per_slab;
    for (int sl = 0; sl < n_slabs; ++sl) slabs.seg_count[sl] = (int32_t)(slab_seg_ptr[sl + 1] - slab_seg_ptr[sl]);
    MM_LAUNCH(ctx, "cam_table_kernel", cam_table_kernel, dim3((pb->F + 63) / 64), dim3(64), 0, pb->F, cams, w.camtab, w.seg_done,
              (int64_t)pb->n_seg, schur_pairs_lean() ? pb->K : (const double *)nullptr, *pb,
              schur_pairs_lean() ? w.desc : (int32_t *)nullptr);
    MM_HIP(ctx, hipMemsetAsync(S, 0, (size_t)n * n * sizeof(double), ctx->stream));
    MM_HIP(ctx, hipMemsetAsync(w.slab_ready, 0, 2 * MAX_SLABS * sizeof(int32_t), ctx->stream));
    MM_LAUNCH(ctx, "schur_diag_fill_kernel", schur_diag_fill_kernel, dim3(pb->F), dim3(64), 0, *pb, Bd, gc, S, v);
    for (int sl = 0; sl < n_slabs; ++sl)   // a slab without any segment is complete as it is
        if (slabs.seg_count[sl] == 0) MM_LAUNCH(ctx, "slab_flag_kernel", slab_flag_kernel, dim3(1), dim3(1), 0, w.slab_ready, sl);
    // The build is ONE ordered list of chunks taken alternately from the front and the back (the two-ended factorisation
    // needs the outer cameras first, the middle last), cut into two launches: the first half runs alone at full speed;
    // the factorisation -- whose workgroups use the whole register file of their CUs, so no wave of the build fits
    // beside them -- starts behind it on a second stream and shares the chip with the second half only, consuming
    // block rows as their slabs are flagged complete.
    const unsigned wg_total = (unsigned)((pb->n_chunks + SP_WAVES - 1) / SP_WAVES);
    const unsigned wg_first = wg_total / 2;
    if (wg_first > 0)
        MM_LAUNCH_PAIRS(ctx, dim3(wg_first), *pb, (const double *)w.camtab, w.desc, pts, (const double *)Cinv, gp, w.partial, Bd, gc, S, v,
                        w.seg_done, slabs, 0u, wg_total);
    MM_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    MM_HIP(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
    {
        mm_stream_swap sw(ctx, ctx->aux);
        int rc = mm_chol_solve_gated(ctx, S, n, v, 1, half_bandwidth, info, ws_chol, ws_chol_bytes, w.slab_ready, cams_per_slab, pb->F, 2);
        if (rc) {  // whatever did get enqueued on the second stream is joined before the error travels up
            (void)hipEventRecord(ctx->ev_join, ctx->aux);
            (void)hipStreamWaitEvent(sw.saved, ctx->ev_join, 0);
            return rc;
        }
        MM_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
    }
    MM_LAUNCH_PAIRS(ctx, dim3(wg_total - wg_first), *pb, (const double *)w.camtab, w.desc, pts, (const double *)Cinv, gp, w.partial, Bd, gc, S, v,
                    w.seg_done, slabs, wg_first, wg_total);
    MM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return MM_OK;
}

// mm_common.h: damping + reduced system + solution for mm_ba_trf (one problem, not sharded)
int mm_ba_schur_solve_damped(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const mm_damp_spec *dmp,
                             double *Bd, double *Cd, const double *gc, const double *gp, double *S, double *v, double *Cinv,
                             int half_bandwidth, int32_t *info, void *ws_schur, size_t ws_schur_bytes, void *ws_chol,
                             size_t ws_chol_bytes) {
    if (!ctx) return MM_ERR_ARG;
    if (!pb || !dmp || !info || !cams || !pts || !Bd || !gc || !gp || !S || !v || !Cinv || !dmp->B || !dmp->C || !dmp->si ||
        (!dmp->gh2 && !dmp->reg) || (dmp->gh2 && (!dmp->d11 || !dmp->damp_out)))
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_schur_solve_damped: bad argument");
    const int n = pb->F * 6;
    static const bool off = [] { const char *e = getenv("MM_SCHUR_FOLD"); return e && atoi(e) == 0; }();
    mm_chol_init_args ia = {};
    int sides = 0, bwb = 0;
    const bool pairs_ok = pb->n_seg > 0 && pb->n_chunks > 0 && pb->seg_ids && pb->seg_chunk_ptr && pb->chunk_seg && pb->chunk_begin &&
                          pb->chunk_end && pb->pair_o && pb->pair_o2 && pb->P > 0 && schur_pairs_lean();
    const bool fold = !off && pairs_ok && ws_schur && ws_schur_bytes >= mm_ba_schur_workspace_bytes(pb) &&
                      mm_chol_init_plan(ctx, n, half_bandwidth, info, ws_chol, ws_chol_bytes, &ia, &sides, &bwb);
    if (!fold) {
        if (!Cd) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_schur_solve_damped: the separate calls need Cd");
        int rc = dmp->gh2 ? mm_ba_damp_damping(ctx, pb->F, pb->P, dmp->B, dmp->C, dmp->si, dmp->gh2, dmp->d11, dmp->Delta, dmp->min_damping,
                                                dmp->damp_out, Bd, Cd)
                          : mm_ba_damp(ctx, pb->F, pb->P, dmp->B, dmp->C, dmp->si, dmp->reg, Bd, Cd);
        if (rc) return rc;
        return mm_ba_schur_solve(ctx, pb, cams, pts, Bd, Cd, gc, gp, S, v, Cinv, half_bandwidth, info, ws_schur, ws_schur_bytes, ws_chol,
                                 ws_chol_bytes, 0, 0, nullptr, nullptr);
    }
    const SchurWs w = carve_schur_ws(pb, ws_schur);
    PrepExtra ex = {};
    ex.dmp = *dmp;
    ex.zero_bwb = bwb;
    ex.chol_init = 1;
    ex.init = ia;
    const int prep = (pb->P + 255) / 256 > pb->F ? (pb->P + 255) / 256 : pb->F;
    MM_LAUNCH(ctx, "schur_prepare_kernel", schur_prepare_damped_kernel, dim3(prep), dim3(256), 0, *pb, cams, Bd, gc, Cinv, S, v, w.camtab,
              w.seg_done, w.desc, ex);
    const int64_t wgs = (pb->n_chunks + SP_WAVES - 1) / SP_WAVES;
    SlabSync none = {};
    MM_LAUNCH_PAIRS(ctx, dim3((unsigned)wgs), *pb, (const double *)w.camtab, w.desc, pts, (const double *)Cinv, gp, w.partial,
                    (const double *)Bd, gc, S, v, w.seg_done, none, 0u, (unsigned)wgs);
    mm_chol_init_done(ctx, ws_chol, sides);
    const int rc = mm_chol_solve_sym(ctx, S, n, v, half_bandwidth, 1, info, ws_chol, ws_chol_bytes);
    mm_chol_init_done(ctx, nullptr, 0);
    return rc;
}

// ---- batched build of the reduced camera systems (mm_ba_trf_batched, trf.hip) ------------------------------------------------
int mm_batch_schur_setup(mm_ctx *ctx, mm_batch_prob *bp, void *ws_schur, size_t ws_schur_bytes) {
    const mm_ba_problem *pb = &bp->pb;
    if (!(pb->n_seg > 0 && pb->n_chunks > 0 && pb->seg_ids && pb->seg_chunk_ptr && pb->chunk_seg && pb->chunk_begin && pb->chunk_end &&
          pb->pair_o && pb->pair_o2 && pb->pair_p) || pb->P <= 0)
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_trf_batched: a problem without a co-observation pair list");
    if (!ws_schur || ws_schur_bytes < mm_ba_schur_workspace_bytes(pb)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_ba_trf_batched: Schur workspace");
    const SchurWs w = carve_schur_ws(pb, ws_schur);
    bp->schur_partial = w.partial;
    bp->schur_camtab = w.camtab;
    bp->schur_seg_done = w.seg_done;
    bp->schur_desc = w.desc;
    bp->g_prep = (uint32_t)((pb->P + 255) / 256 > pb->F ? (pb->P + 255) / 256 : pb->F);
    bp->g_pairs = (uint32_t)((pb->n_chunks + SP_WAVES - 1) / SP_WAVES);
    const int64_t n2 = bp->nc * bp->nc / 2;
    const int64_t gz = (n2 + 2047) / 2048;
    bp->g_zero = (uint32_t)(gz < 1 ? 1 : (gz > 1024 ? 1024 : gz));
    return MM_OK;
}
int mm_batch_schur(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g_zero, unsigned max_g_prep,
                   unsigned max_g_pairs) {
    if (n_list <= 0) return MM_OK;
    MM_LAUNCH(ctx, "schur_zero_kernel", zero_batch_kernel, dim3(max_g_zero, (unsigned)n_list), dim3(256), 0, tab, list);
    MM_LAUNCH(ctx, "schur_prepare_kernel", schur_prepare_batch_kernel, dim3(max_g_prep, (unsigned)n_list), dim3(256), 0, tab, list);
    MM_LAUNCH(ctx, "schur_pairs_kernel", schur_pairs_batch_kernel, dim3(max_g_pairs, (unsigned)n_list), dim3(64 * SP_WAVES), 0, tab, list);
    return MM_OK;
}
