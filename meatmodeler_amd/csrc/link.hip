// Track linking over a whole clip ON THE DEVICE (gfx950): the data-parallel restatement of processor.pointTracking
// (reference processor.py:190-243) applied to every consecutive keyframe pair, followed by `popped_tracks += tracks`
// (processor.py:418) and the flattening order of managePoints (processor.py:264-291).
//
// Semantics kept exactly (same as the host linker mm_link_tracks_clip, which stays as the cross-check):
//   * a match continues the FIRST live track (list order) whose coordinate at the previous keyframe EQUALS the match's
//     previous-frame point (float equality, processor.py:220)  -> scatter-min of the live position over a canonical
//     key point id (smallest index with the same coordinates in that frame);
//   * several matches hitting one track: the last one wins (Track.update overwrites, track.py:17-19) -> scatter-max of
//     the match index;
//   * unmatched feature points spawn new tracks after the survivors, in match order; tracks not updated are popped in
//     list order -> three block-wide prefix sums per keyframe pair.
// Two formulations, identical results:
//   * serial (link_kernel, MM_LINK_VARIANT=serial): ONE resident workgroup walks the clip pair by pair (a handful of
//     scatters / scans over <= 8192 items per pair, ~34 us each: 17 ms per 500-frame clip);
//   * parallel (the default, "lp_" kernels below): the live list is always ordered by BIRTH (survivors keep their
//     order, new tracks are appended), so "the first live track at a coordinate" is the OLDEST track there, and the
//     whole clip becomes a forest over nodes (frame, canonical key point) that pointer doubling resolves in
//     log2(frames) rounds over all matches of the clip at once.
#include "mm_common.h"
#include <climits>
#include <type_traits>

namespace {

constexpr int LK_THREADS = 1024;
constexpr int LK_IPT = 8;                       // items per thread in the block scans
constexpr int LK_MAX_CAP = LK_THREADS * LK_IPT;  // key points / matches per frame

// canon[f][i] = smallest j with the same (x, y) in frame f (coordinates compare with ==, so -0.0 == +0.0).
// One workgroup per frame and an open-addressing hash table of key point indices in LDS (2 cap rounded up to a power of
// two slots): a key point claims the first empty slot of its probe sequence or, when it meets an occupant with equal
// coordinates, lowers that slot to its own index (atomicMin: the occupant may change, its coordinates do not), so every
// group of equal coordinates ends up in ONE slot holding its smallest index.  (All-pairs comparison was 1.5 ms per
// 500 x 4000 clip and grows with the square of the key point count.)
constexpr int CANON_THREADS = 1024;
__device__ __forceinline__ unsigned canon_hash(float x, float y) {
    const unsigned a = __float_as_uint(x + 0.0f), b = __float_as_uint(y + 0.0f);      // (-0.0 + 0.0 = +0.0)
    unsigned h = a * 0x9E3779B1u ^ (b * 0x85EBCA77u + (a >> 15));
    return h ^ (h >> 13);
}
__global__ __launch_bounds__(CANON_THREADS) void link_canon_kernel(int cap, int slots /*power of two*/, const int32_t *__restrict__ kp_count,
                                                                  const float *__restrict__ kp_xy, int32_t *__restrict__ canon) {
    extern __shared__ int32_t table[];
    const int f = blockIdx.x;
    const int n = min(kp_count[f], cap);
    const float2 *xy = reinterpret_cast<const float2 *>(kp_xy) + (size_t)f * cap;
    for (int s_ = threadIdx.x; s_ < slots; s_ += CANON_THREADS) table[s_] = INT_MAX;
    __syncthreads();
    const unsigned mask = (unsigned)slots - 1u;
    for (int i = threadIdx.x; i < n; i += CANON_THREADS) {
        const float2 me = xy[i];
        if (me.x != me.x || me.y != me.y) continue;      // NaN equals nothing: its own canonical id, never in the table
        unsigned s_ = canon_hash(me.x, me.y) & mask;
        for (int probes = 0; probes < slots; ++probes) {      // (bounded: the table has 2 cap slots, it cannot fill up)
            int occ = __hip_atomic_load(&table[s_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (occ == INT_MAX) {
                occ = atomicCAS(&table[s_], INT_MAX, i);
                if (occ == INT_MAX) break;      // claimed
            }
            const float2 o = xy[occ];           // (whoever sits here now: equal coordinates stay equal)
            if (o.x == me.x && o.y == me.y) {
                atomicMin(&table[s_], i);
                break;
            }
            s_ = (s_ + 1) & mask;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += CANON_THREADS) {
        const float2 me = xy[i];
        int best = i;
        if (me.x == me.x && me.y == me.y) {
            unsigned s_ = canon_hash(me.x, me.y) & mask;
            for (int probes = 0; probes < slots; ++probes) {
                const int occ = table[s_];
                if (occ == INT_MAX) break;      // (not reached: every finite point is in the table)
                const float2 o = xy[occ];
                if (o.x == me.x && o.y == me.y) {
                    best = occ;
                    break;
                }
                s_ = (s_ + 1) & mask;
            }
        }
        canon[(size_t)f * cap + i] = best;
    }
}

struct LinkWs {
    int32_t *canon;                      // [F, cap]
    int32_t *owner, *lastm, *hitpos;     // [cap]
    int32_t *live_track[2], *live_kp[2], *live_node[2];  // [cap]
    int32_t *node_kp, *node_frame, *node_prev;           // [max_nodes]
    int32_t *track_tail, *track_len, *final_order;       // [max_tracks]
};

// owner / lastm are written by device-scope atomics: read them past the per-CU vector cache
__device__ __forceinline__ int ld_agent(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// exclusive ranks of `flag[0..LK_IPT)` (items tid*LK_IPT ...) over the workgroup; returns the total
__device__ __forceinline__ int block_exscan(const int (&flag)[LK_IPT], int (&rank)[LK_IPT], int *s_wave /*[17]*/) {
    int local = 0;
#pragma unroll
    for (int q = 0; q < LK_IPT; ++q) {
        rank[q] = local;
        local += flag[q];
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < LK_THREADS / 64; ++i) {
            const int t = s_wave[i];
            s_wave[i] = run;
            run += t;
        }
        s_wave[16] = run;
    }
    __syncthreads();
    const int base = s_wave[w] + incl - local;
#pragma unroll
    for (int q = 0; q < LK_IPT; ++q) rank[q] += base;
    return s_wave[16];
}

// three exclusive scans at once (same tree, one set of barriers); totals in tot[]
__device__ __forceinline__ void block_exscan3(const int (&f0)[LK_IPT], const int (&f1)[LK_IPT], const int (&f2)[LK_IPT],
                                              int (&r0)[LK_IPT], int (&r1)[LK_IPT], int (&r2)[LK_IPT], int (&tot)[3],
                                              int (*s_w)[17] /*[3][17]*/) {
    int l0 = 0, l1 = 0, l2 = 0;
#pragma unroll
    for (int q = 0; q < LK_IPT; ++q) {
        r0[q] = l0;
        r1[q] = l1;
        r2[q] = l2;
        l0 += f0[q];
        l1 += f1[q];
        l2 += f2[q];
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int i0 = l0, i1 = l1, i2 = l2;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t0 = __shfl_up(i0, off, 64), t1 = __shfl_up(i1, off, 64), t2 = __shfl_up(i2, off, 64);
        if (lane >= off) {
            i0 += t0;
            i1 += t1;
            i2 += t2;
        }
    }
    __syncthreads();
    if (lane == 63) {
        s_w[0][w] = i0;
        s_w[1][w] = i1;
        s_w[2][w] = i2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        int run = 0;
        for (int i = 0; i < LK_THREADS / 64; ++i) {
            const int t = s_w[threadIdx.x][i];
            s_w[threadIdx.x][i] = run;
            run += t;
        }
        s_w[threadIdx.x][16] = run;
    }
    __syncthreads();
    const int b0 = s_w[0][w] + i0 - l0, b1 = s_w[1][w] + i1 - l1, b2 = s_w[2][w] + i2 - l2;
#pragma unroll
    for (int q = 0; q < LK_IPT; ++q) {
        r0[q] += b0;
        r1[q] += b1;
        r2[q] += b2;
    }
    tot[0] = s_w[0][16];
    tot[1] = s_w[1][16];
    tot[2] = s_w[2][16];
}

template <int LDS_MODE>      // 2: the whole per-pair state in LDS; 1: the three scatter tables only; 0: all in global memory
__global__ __launch_bounds__(LK_THREADS) void link_kernel(int F, int cap, const int32_t *__restrict__ kp_count,
                                                          const int32_t *__restrict__ match_count,
                                                          const int32_t *__restrict__ matches, LinkWs ws,
                                                          int32_t *__restrict__ track_ptr, int64_t *__restrict__ counts) {
    // The per-pair state (first-owner table, last-match table, hit positions, the two live lists: 9 arrays of `cap`
    // words) lives in LDS when it fits (cap <= 4096: 144 KB of the CU's 160 KB; up to the 8192 key points of a 4K frame the
    // three scatter tables still do: 96 KB) -- every phase of a pair is a dependent
    // round trip to these arrays, ~2 us each through global memory, a few hundred ns through LDS.
    // (A template parameter, not a run-time choice, and the two live lists addressed by offset rather than through an
    // array of pointers indexed by `cur`: with one provenance per access the compiler emits ds_* instructions.  Flat
    // accesses to LDS count on vmcnt as well, so every read of the live lists waited for the global stores issued just
    // before it: 8 of the 14 us of a pair.)
    extern __shared__ int32_t lds_state[];
    int32_t *owner, *lastm, *hitpos;
    if constexpr (LDS_MODE >= 1) {
        owner = lds_state;
        lastm = lds_state + cap;
        hitpos = lds_state + 2 * cap;
    } else {
        owner = ws.owner;
        lastm = ws.lastm;
        hitpos = ws.hitpos;
    }
    auto live_track = [&](int b, int i) -> int32_t & {
        if constexpr (LDS_MODE == 2) return lds_state[(3 + 3 * b) * cap + i];
        else return ws.live_track[b][i];
    };
    auto live_kp = [&](int b, int i) -> int32_t & {
        if constexpr (LDS_MODE == 2) return lds_state[(4 + 3 * b) * cap + i];
        else return ws.live_kp[b][i];
    };
    auto live_node = [&](int b, int i) -> int32_t & {
        if constexpr (LDS_MODE == 2) return lds_state[(5 + 3 * b) * cap + i];
        else return ws.live_node[b][i];
    };
    __shared__ int s_wave[17];
    __shared__ int s_wave3[3][17];
    __shared__ int s_T, s_ntracks, s_popbase, s_nodebase, s_bad;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_T = 0;
        s_ntracks = 0;
        s_popbase = 0;
        s_nodebase = 0;
        s_bad = 0;
    }
    __syncthreads();
    int cur = 0;
    for (int k = 0; k + 1 < F; ++k) {
        const int T = s_T, n_tracks = s_ntracks, pop_base = s_popbase, node_base = s_nodebase;
        const int M = min(max(match_count[k], 0), cap);
        const int nk = min(kp_count[k], cap), nk1 = min(kp_count[k + 1], cap);
        const int32_t *mk = matches + (size_t)k * cap * 2;
        const int32_t *ck = ws.canon + (size_t)k * cap;
        for (int i = tid; i < cap; i += LK_THREADS) {
            owner[i] = INT_MAX;
            lastm[i] = -1;
        }
        __syncthreads();
        // A: first live position per canonical key point of frame k
        for (int pos = tid; pos < T; pos += LK_THREADS) atomicMin(&owner[ck[live_kp(cur, pos)]], pos);
        __syncthreads();
        // B: each match finds its track (or none); the last match on a track wins
        for (int m = tid; m < M; m += LK_THREADS) {
            const int q = mk[2 * m], t = mk[2 * m + 1];
            int pos = -1;
            if (q < 0 || q >= nk || t < 0 || t >= nk1) {
                s_bad = 1;  // malformed match: ignored (and reported)
                pos = -2;
            } else {
                const int o = ld_agent(&owner[ck[q]]);
                if (o != INT_MAX) {
                    pos = o;
                    atomicMax(&lastm[o], m);
                }
            }
            hitpos[m] = pos;
        }
        __syncthreads();
        // C: ranks of survivors / popped tracks (over live positions) and of new tracks (over matches)
        int f_surv[LK_IPT], f_pop[LK_IPT], f_new[LK_IPT], r_surv[LK_IPT], r_pop[LK_IPT], r_new[LK_IPT], l_last[LK_IPT];
#pragma unroll
        for (int q = 0; q < LK_IPT; ++q) {
            const int i = tid * LK_IPT + q;
            const bool live = i < T;
            const int lm = live ? ld_agent(&lastm[i]) : -1;
            const bool upd = lm >= 0;
            l_last[q] = lm;
            f_surv[q] = upd;
            f_pop[q] = live && !upd;
            f_new[q] = (i < M) && hitpos[i] == -1;
        }
        int tot3[3];
        block_exscan3(f_surv, f_pop, f_new, r_surv, r_pop, r_new, tot3, s_wave3);
        const int n_surv = tot3[0], n_pop = tot3[1], n_new = tot3[2];
        // D: next live list, observation nodes, popped tracks.  The ranks were computed on contiguous chunks per thread;
        // written with that mapping, the node / list stores of a wave land 8 words apart (64 cache sectors per store
        // instruction, and the one CU's store path was 40 % of a pair).  So the ranks go through LDS and phase D runs
        // with the interleaved mapping i = tid + 1024 q: consecutive lanes, (mostly) consecutive destinations.
        //   rank_live (in `hitpos`):  2 r_surv  |  2 r_pop + 1  |  -1 (no live track at this position)
        //   rank_new  (in `owner`, free since phase B):  r_new  |  -1
        const int nxt = cur ^ 1;
#pragma unroll
        for (int q = 0; q < LK_IPT; ++q) {
            const int i = tid * LK_IPT + q;
            if (i < cap) {
                const int rn = f_new[q] ? r_new[q] : -1;
                const int rl = f_surv[q] ? 2 * r_surv[q] : (f_pop[q] ? 2 * r_pop[q] + 1 : -1);
                owner[i] = rn;
                hitpos[i] = rl;
            }
        }
        __syncthreads();
        for (int i = tid; i < cap; i += LK_THREADS) {
            const int rl = hitpos[i], rn = owner[i];
            if (rl >= 0 && !(rl & 1)) {
                const int r = rl >> 1;
                const int tr = live_track(cur, i);
                const int t = mk[2 * lastm[i] + 1];
                const int nid = node_base + r;
                ws.node_kp[nid] = t;
                ws.node_frame[nid] = k + 1;
                ws.node_prev[nid] = live_node(cur, i);
                live_track(nxt, r) = tr;
                live_kp(nxt, r) = t;
                live_node(nxt, r) = nid;
                atomicAdd(&ws.track_len[tr], 1);   // (no result needed: fire and forget)
            } else if (rl >= 0) {
                const int tr = live_track(cur, i);
                ws.final_order[pop_base + (rl >> 1)] = tr;
                ws.track_tail[tr] = live_node(cur, i);
            }
            if (rn >= 0) {
                const int tr = n_tracks + rn;
                const int nid = node_base + n_surv + 2 * rn;
                const int2 qt = *reinterpret_cast<const int2 *>(mk + 2 * i);
                ws.node_kp[nid] = qt.x;
                ws.node_frame[nid] = k;
                ws.node_prev[nid] = -1;
                ws.node_kp[nid + 1] = qt.y;
                ws.node_frame[nid + 1] = k + 1;
                ws.node_prev[nid + 1] = nid;
                const int p = n_surv + rn;
                live_track(nxt, p) = tr;
                live_kp(nxt, p) = qt.y;
                live_node(nxt, p) = nid + 1;
                ws.track_len[tr] = 2;
            }
        }
        __syncthreads();
        if (tid == 0) {
            s_T = n_surv + n_new;
            s_ntracks = n_tracks + n_new;
            s_popbase = pop_base + n_pop;
            s_nodebase = node_base + n_surv + 2 * n_new;
        }
        cur = nxt;
        __syncthreads();
    }
    // the tracks still alive come last (processor.py:418)
    const int T = s_T, n_tracks = s_ntracks, pop_base = s_popbase;
    for (int pos = tid; pos < T; pos += LK_THREADS) {
        const int tr = live_track(cur, pos);
        ws.final_order[pop_base + pos] = tr;
        ws.track_tail[tr] = live_node(cur, pos);
    }
    __syncthreads();
    // CSR offsets in final order: chunked block scan with a running base
    int base = 0;
    for (int c0 = 0; c0 < n_tracks; c0 += LK_MAX_CAP) {
        int len[LK_IPT], rk[LK_IPT];
#pragma unroll
        for (int q = 0; q < LK_IPT; ++q) {
            const int i = c0 + tid * LK_IPT + q;
            len[q] = i < n_tracks ? ws.track_len[ws.final_order[i]] : 0;
        }
        const int tot = block_exscan(len, rk, s_wave);
#pragma unroll
        for (int q = 0; q < LK_IPT; ++q) {
            const int i = c0 + tid * LK_IPT + q;
            if (i < n_tracks) track_ptr[i] = base + rk[q];
        }
        base += tot;
        __syncthreads();
    }
    if (tid == 0) {
        track_ptr[n_tracks] = base;
        counts[0] = n_tracks;
        counts[1] = base;
        counts[2] = s_bad;
    }
}

// one thread per track: walk the observation list backwards, fill the CSR forwards
__global__ __launch_bounds__(256) void link_emit_kernel(LinkWs ws, const int32_t *__restrict__ track_ptr,
                                                        const int64_t *__restrict__ counts,
                                                        int32_t *__restrict__ obs_frame, int32_t *__restrict__ obs_kp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= counts[0]) return;
    const int tr = ws.final_order[i];
    int w = track_ptr[i] + ws.track_len[tr] - 1;
    for (int nid = ws.track_tail[tr]; nid >= 0; nid = ws.node_prev[nid], --w) {
        obs_frame[w] = ws.node_frame[nid];
        obs_kp[w] = ws.node_kp[nid];
    }
}

// ---------------------------------------------------------------------------------------------- the parallel formulation
// Nodes are (frame f, canonical key point c), n = f cap + c; a match g = k cap + m of pair k is an edge from node
// (k, canon[q]) to node (k + 1, canon[t]).  With has(n) = "a live track sits at n":
//   * a node with a track and matches leaving it continues its OLDEST track along its LAST match (the others are absorbed:
//     Track.update overwrites): next(n);
//   * every match leaving a node without a track starts a track, born at g (birth order = list order, see above);
//   * several tracks arriving at one node: the oldest stays the owner, the others are never found again (popped there).
// So the owner of a node is the smallest birth among the tracks born in the in-tree below it: birth(n) = min over the
// subtree, computed by pointer doubling (round i pushes a node's minimum to its ancestor 2^i levels up).  The track born
// at g = k0 cap + m0 then owns every continuation edge leaving a node whose owner it is; its observation at frame f
// sits at index f - k0; its last frame is the largest one reached.  Final order (popped_tracks in pop order, then the
// survivors: processor.py:418) = ascending (last frame, birth): a bucket per last frame -- at most `cap` tracks, they were
// all live at that frame -- sorted by birth inside one workgroup.
// has(n) itself = some EFFECTIVE edge arrives: the last match of any node always is (continuation or birth), the other
// matches of a node with several (key points with equal coordinates) only if that node has no track: resolved by a
// fixed-point pass over those few matches (layered by frame: exact after as many rounds as such nodes chain up).
// Cost of that pass: one workgroup, three sweeps over the uncertain matches per round (36 us at 500 x 4000, 0.64 ms at
// 2000 x 8000 key points, where ORB levels produce more coincident coordinates); the loop is bounded by the number of frames.
// A clip whose key points ALL share one coordinate (every match uncertain, chains as long as the clip) is the worst case:
// seconds, not a hang -- MM_LINK_VARIANT=serial has no such case and is what to use for synthetic stress inputs of that kind.
struct ParWs {
    int32_t *canon;                  // [F, cap]
    int32_t *lastm, *jump[2];        // nodes [F cap]: last match leaving the node (-1), doubling pointers (-1)
    uint32_t *birth;                 // nodes: smallest birth in the subtree (0xFFFFFFFF = no track)
    uint8_t *cert, *uin;             // nodes: an effective edge arrives (last matches | resolved others)
    int32_t *src, *dst;              // matches [(F-1) cap]: node ids, -1 = malformed (ignored)
    uint8_t *kind, *eff;             // matches: 0 absorbed, 1 starts a track, 2 continues one; scratch of the fixed point
    int32_t *unc;                    // the matches that are not the last one of their node
    int32_t *ptr_of;                 // by birth g: the track's first observation
    int32_t *bcount, *bobs, *tbase, *obase;      // frames [F]: tracks ending there, their observations, exclusive sums
    int32_t *bucket, *loff;          // [F cap]: births ending at a frame (slot = match index of the arriving edge, then
                                     // closed up and sorted), observation offset inside the bucket
    int32_t *scal;                   // [0] uncertain matches, [1] malformed matches, [2] the fixed-point pass gave up
    size_t ff_bytes, zero_off, zero_bytes;       // the two memset regions
};
constexpr uint32_t LP_NONE = 0xFFFFFFFFu;

struct MatchAt {
    int k, m, M;
};
__device__ __forceinline__ bool lp_match_at(int64_t g, int cap, const int32_t *match_count, MatchAt &a) {
    a.k = (int)(g / cap);
    a.m = (int)(g - (int64_t)a.k * cap);
    a.M = min(max(match_count[a.k], 0), cap);
    return a.m < a.M;
}

__global__ __launch_bounds__(256) void lp_edges_kernel(int64_t n_match, int cap, const int32_t *__restrict__ kp_count,
                                                       const int32_t *__restrict__ match_count, const int32_t *__restrict__ matches,
                                                       ParWs w) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    MatchAt a;
    if (g >= n_match || !lp_match_at(g, cap, match_count, a)) return;
    const int2 qt = reinterpret_cast<const int2 *>(matches)[g];
    const int nk = min(kp_count[a.k], cap), nk1 = min(kp_count[a.k + 1], cap);
    if (qt.x < 0 || qt.x >= nk || qt.y < 0 || qt.y >= nk1) {      // malformed match: ignored (and reported)
        w.scal[1] = 1;
        w.src[g] = -1;
        return;
    }
    const int s_ = a.k * cap + w.canon[(size_t)a.k * cap + qt.x], d = (a.k + 1) * cap + w.canon[(size_t)(a.k + 1) * cap + qt.y];
    w.src[g] = s_;
    w.dst[g] = d;
    atomicMax(&w.lastm[s_], a.m);
}

__global__ __launch_bounds__(256) void lp_certain_kernel(int64_t n_match, int cap, const int32_t *__restrict__ match_count, ParWs w) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    MatchAt a;
    if (g >= n_match || !lp_match_at(g, cap, match_count, a)) return;
    const int s_ = w.src[g];
    if (s_ < 0) return;
    if (w.lastm[s_] == a.m) w.cert[w.dst[g]] = 1;
    else w.unc[atomicAdd(&w.scal[0], 1)] = (int32_t)g;
}

// effective(u) = the node u leaves has no track = no certain edge and no effective uncertain edge arrives there
// Work budget: the pass is one workgroup, so rounds x uncertain matches is its running time (~1 ns per entry and sweep).
// Past `budget` entries (a clip whose key points coincide clip-wide: every match uncertain, chains as long as the clip) it
// gives up and says so in scal[2]; mm_link_tracks_device then takes the serial formulation, whose cost does not depend on
// the input (no input may occupy the GPU for seconds).
__global__ __launch_bounds__(1024) void lp_uncertain_kernel(int F, ParWs w, long long budget) {
    __shared__ int s_changed;
    const int n = w.scal[0];
    if (n == 0) return;
    if ((long long)n > budget) {
        if (threadIdx.x == 0) w.scal[2] = 1;
        return;
    }
    for (int i = threadIdx.x; i < n; i += 1024) w.eff[i] = !w.cert[w.src[w.unc[i]]];
    __syncthreads();
    for (int round = 0; round <= F; ++round) {
        if ((long long)(round + 1) * n > budget) {      // (uniform: every thread sees the same round and n)
            if (threadIdx.x == 0) w.scal[2] = 1;
            return;
        }
        for (int i = threadIdx.x; i < n; i += 1024) w.uin[w.dst[w.unc[i]]] = 0;
        if (threadIdx.x == 0) s_changed = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 1024)
            if (w.eff[i]) w.uin[w.dst[w.unc[i]]] = 1;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 1024) {
            const int s_ = w.src[w.unc[i]];
            const uint8_t e = !(w.cert[s_] | w.uin[s_]);
            if (e != w.eff[i]) {
                w.eff[i] = e;
                s_changed = 1;
            }
        }
        __syncthreads();
        if (!s_changed) break;      // (uin was built from the effective set that just proved stable)
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void lp_classify_kernel(int64_t n_match, int cap, const int32_t *__restrict__ match_count, ParWs w) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    MatchAt a;
    if (g >= n_match || !lp_match_at(g, cap, match_count, a)) return;
    const int s_ = w.src[g];
    if (s_ < 0) return;
    uint8_t kind = 0;
    if (w.cert[s_] | w.uin[s_]) {
        if (w.lastm[s_] == a.m) {
            w.jump[0][s_] = w.dst[g];
            kind = 2;
        }
    } else {
        atomicMin(&w.birth[w.dst[g]], (uint32_t)g);
        kind = 1;
    }
    w.kind[g] = kind;
}

// one doubling round: push the node's minimum to the ancestor `jin` points at, then point twice as far
__global__ __launch_bounds__(256) void lp_jump_kernel(int64_t n_nodes, uint32_t *__restrict__ birth, const int32_t *__restrict__ jin,
                                                      int32_t *__restrict__ jout) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_nodes) return;
    const int j = jin[u];
    int jj = -1;
    if (j >= 0) {
        const uint32_t b = __hip_atomic_load(&birth[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b != LP_NONE) atomicMin(&birth[j], b);
        jj = jin[j];
    }
    jout[u] = jj;
}

// The edge that brings a track to its LAST frame is the one whose target either has no match leaving it or is owned by an
// older track.  That edge is unique per track and -- every live track of a frame arrived by its own match of the pair
// before -- its match index is a slot of the frame's bucket nobody else writes: plain stores, no atomics.
__global__ __launch_bounds__(256) void lp_final_kernel(int64_t n_match, int cap, const int32_t *__restrict__ match_count, ParWs w) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    MatchAt a;
    if (g >= n_match || !lp_match_at(g, cap, match_count, a)) return;
    if (w.src[g] < 0) return;
    const uint8_t kind = w.kind[g];
    if (kind == 0) return;
    uint32_t g0 = (uint32_t)g;
    if (kind == 2) {
        g0 = w.birth[w.src[g]];
        if (g0 == LP_NONE) {      // (not reached: a node that continues a track owns one)
            w.scal[1] = 2;
            return;
        }
    }
    const int d = w.dst[g];
    if (w.lastm[d] < 0 || w.birth[d] != g0) w.bucket[(size_t)(a.k + 1) * cap + a.m] = (int32_t)g0;
}

// one workgroup per frame: the births ending there in ascending order (bitonic network in LDS), and the offsets of their
// observation runs inside the bucket (a track born in pair k0 that ends at frame f has f - k0 + 1 observations)
__global__ __launch_bounds__(LK_THREADS) void lp_sort_kernel(int cap, ParWs w) {
    __shared__ uint32_t keys[LK_MAX_CAP];
    __shared__ int s_wave[17];
    const int f = blockIdx.x, tid = threadIdx.x;
    int32_t *bk = w.bucket + (size_t)f * cap;
    // the bucket has holes (slots = match indices of the pair before): close them up in LDS
    int len[LK_IPT], off[LK_IPT];
    uint32_t mine[LK_IPT];
#pragma unroll
    for (int q = 0; q < LK_IPT; ++q) {
        const int i = tid * LK_IPT + q;
        mine[q] = i < cap ? (uint32_t)bk[i] : LP_NONE;
        len[q] = mine[q] != LP_NONE;
    }
    const int n = block_exscan(len, off, s_wave);
    if (n == 0) {
        if (tid == 0) {
            w.bobs[f] = 0;
            w.bcount[f] = 0;
        }
        return;
    }
    int P = 2;
    while (P < n) P <<= 1;
    for (int i = n + tid; i < P; i += LK_THREADS) keys[i] = LP_NONE;
#pragma unroll
    for (int q = 0; q < LK_IPT; ++q)
        if (len[q]) keys[off[q]] = mine[q];
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P / 2; t += LK_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), x = i | j;      // the pair (i, i + j)
                const uint32_t a = keys[i], b = keys[x];
                if ((a > b) == ((i & k2) == 0)) {
                    keys[i] = b;
                    keys[x] = a;
                }
            }
            __syncthreads();
        }
#pragma unroll
    for (int q = 0; q < LK_IPT; ++q) {
        const int i = tid * LK_IPT + q;
        len[q] = i < n ? f - (int)(keys[i] / (uint32_t)cap) + 1 : 0;
    }
    const int total = block_exscan(len, off, s_wave);
#pragma unroll
    for (int q = 0; q < LK_IPT; ++q) {
        const int i = tid * LK_IPT + q;
        if (i < n) {
            bk[i] = (int32_t)keys[i];
            w.loff[(size_t)f * cap + i] = off[q];
        }
    }
    if (tid == 0) {
        w.bobs[f] = total;
        w.bcount[f] = n;
    }
}

// exclusive sums over the frames (tracks, observations) and the totals
__global__ __launch_bounds__(LK_THREADS) void lp_frames_kernel(int F, int cap, ParWs w, int32_t *__restrict__ track_ptr,
                                                              int64_t *__restrict__ counts) {
    __shared__ int s_wave[17];
    const int tid = threadIdx.x;
    int tb = 0, ob = 0;
    for (int c0 = 0; c0 < F; c0 += LK_MAX_CAP) {
        int a[LK_IPT], b[LK_IPT], ra[LK_IPT], rb[LK_IPT];
#pragma unroll
        for (int q = 0; q < LK_IPT; ++q) {
            const int f = c0 + tid * LK_IPT + q;
            a[q] = f < F ? min(w.bcount[f], cap) : 0;
            b[q] = f < F ? w.bobs[f] : 0;
        }
        const int ta = block_exscan(a, ra, s_wave);
        __syncthreads();
        const int tb_ = block_exscan(b, rb, s_wave);
#pragma unroll
        for (int q = 0; q < LK_IPT; ++q) {
            const int f = c0 + tid * LK_IPT + q;
            if (f < F) {
                w.tbase[f] = tb + ra[q];
                w.obase[f] = ob + rb[q];
            }
        }
        tb += ta;
        ob += tb_;
        __syncthreads();
    }
    if (tid == 0) {
        track_ptr[tb] = ob;
        counts[0] = tb;
        counts[1] = ob;
        counts[2] = w.scal[1];
    }
}

__global__ __launch_bounds__(256) void lp_ptr_kernel(int cap, ParWs w, int32_t *__restrict__ track_ptr) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(w.bcount[f], cap)) return;
    const int p = w.obase[f] + w.loff[(size_t)f * cap + i];
    track_ptr[w.tbase[f] + i] = p;
    w.ptr_of[w.bucket[(size_t)f * cap + i]] = p;
}

__global__ __launch_bounds__(256) void lp_emit_kernel(int64_t n_match, int cap, const int32_t *__restrict__ match_count,
                                                      const int32_t *__restrict__ matches, ParWs w, int32_t *__restrict__ obs_frame,
                                                      int32_t *__restrict__ obs_kp) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    MatchAt a;
    if (g >= n_match || !lp_match_at(g, cap, match_count, a)) return;
    if (w.src[g] < 0) return;
    const uint8_t kind = w.kind[g];
    if (kind == 0) return;
    const int2 qt = reinterpret_cast<const int2 *>(matches)[g];
    if (kind == 1) {
        const int p = w.ptr_of[g];
        obs_frame[p] = a.k;
        obs_kp[p] = qt.x;
        obs_frame[p + 1] = a.k + 1;
        obs_kp[p + 1] = qt.y;
    } else {
        const uint32_t g0 = w.birth[w.src[g]];
        if (g0 == LP_NONE) return;
        const int p = w.ptr_of[g0] + (a.k + 1 - (int)(g0 / (uint32_t)cap));
        obs_frame[p] = a.k + 1;
        obs_kp[p] = qt.y;
    }
}

size_t carve_par(ParWs &w, uint8_t *base, int F, int cap) {
    size_t off = 0;
    auto take = [&](auto *&p, size_t n) {
        using T = std::remove_reference_t<decltype(*p)>;
        p = reinterpret_cast<T *>(base + off);
        off += mm_align_up(n * sizeof(T), 256);
    };
    const size_t nodes = (size_t)F * cap, nm = (size_t)(F > 1 ? F - 1 : 0) * cap;
    // (0xFF region)
    take(w.lastm, nodes);
    take(w.jump[0], nodes);
    take(w.birth, nodes);
    take(w.bucket, nodes);
    w.ff_bytes = off;
    // (zero region)
    w.zero_off = off;
    take(w.cert, nodes);
    take(w.uin, nodes);
    take(w.scal, 4);
    w.zero_bytes = off - w.zero_off;
    // (written before read)
    take(w.canon, nodes);
    take(w.jump[1], nodes);
    take(w.src, nm + 1);
    take(w.dst, nm + 1);
    take(w.kind, nm + 1);
    take(w.eff, nm + 1);
    take(w.unc, nm + 1);
    take(w.ptr_of, nm + 1);
    take(w.bobs, (size_t)F + 1);
    take(w.tbase, (size_t)F + 1);
    take(w.obase, (size_t)F + 1);
    take(w.bcount, (size_t)F + 1);
    take(w.loff, nodes);
    return off;
}

size_t carve(LinkWs &w, uint8_t *base, int F, int cap) {
    size_t off = 0;
    auto take = [&](int32_t *&p, size_t n) {
        p = reinterpret_cast<int32_t *>(base + off);
        off += mm_align_up(n * sizeof(int32_t), 256);
    };
    const size_t pairs = F > 1 ? (size_t)(F - 1) : 0;
    const size_t max_tracks = pairs * cap + 1, max_nodes = 2 * pairs * cap + 2;
    take(w.canon, (size_t)F * cap);
    take(w.owner, cap);
    take(w.lastm, cap);
    take(w.hitpos, cap);
    for (int b = 0; b < 2; ++b) {
        take(w.live_track[b], cap);
        take(w.live_kp[b], cap);
        take(w.live_node[b], cap);
    }
    take(w.node_kp, max_nodes);
    take(w.node_frame, max_nodes);
    take(w.node_prev, max_nodes);
    take(w.track_tail, max_tracks);
    take(w.track_len, max_tracks);
    take(w.final_order, max_tracks);
    return off;
}

}  // namespace

extern "C" {

size_t mm_link_workspace_bytes(int n_frames, int cap) {
    if (n_frames < 0 || cap < 0) return 0;
    LinkWs w;
    ParWs pw;
    const size_t a = carve(w, nullptr, n_frames, cap), b = carve_par(pw, nullptr, n_frames, cap);
    return a > b ? a : b;
}

int mm_link_tracks_device(mm_ctx *ctx, int n_frames, int cap, const int32_t *kp_count, const float *kp_xy,
                          const int32_t *match_count, const int32_t *matches, void *ws, size_t ws_bytes,
                          int32_t *track_ptr, int32_t *obs_frame, int32_t *obs_kp, int64_t *counts) {
    if (!ctx) return MM_ERR_ARG;
    if (n_frames < 0 || cap < 0 || !counts) return mm_fail(ctx, MM_ERR_ARG, "mm_link_tracks_device: bad argument");
    if (cap > LK_MAX_CAP) return mm_fail(ctx, MM_ERR_ARG, "mm_link_tracks_device: cap > %d key points per frame", LK_MAX_CAP);
    if ((int64_t)(n_frames > 1 ? n_frames - 1 : 0) * cap * 2 + 2 > INT_MAX)
        return mm_fail(ctx, MM_ERR_ARG, "mm_link_tracks_device: clip too large for 32-bit node ids");
    if (n_frames < 2 || cap == 0) {
        MM_HIP(ctx, hipMemsetAsync(counts, 0, 3 * sizeof(int64_t), ctx->stream));
        if (track_ptr) MM_HIP(ctx, hipMemsetAsync(track_ptr, 0, sizeof(int32_t), ctx->stream));
        return MM_OK;
    }
    if (!kp_count || !kp_xy || !match_count || !matches || !ws || !track_ptr || !obs_frame || !obs_kp)
        return mm_fail(ctx, MM_ERR_ARG, "mm_link_tracks_device: null pointer");
    if (((uintptr_t)ws & 255) || ((uintptr_t)kp_xy & 7)) return mm_fail(ctx, MM_ERR_ARG, "mm_link_tracks_device: alignment");
    if (ws_bytes < mm_link_workspace_bytes(n_frames, cap)) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_link_tracks_device: workspace too small");
    const char *variant = getenv("MM_LINK_VARIANT");
    bool serial = variant && variant[0] == 's';
    // entries (rounds x uncertain matches) the parallel formulation's fixed-point pass may sweep before the call switches
    // to the serial formulation: ~1 ns each, i.e. a few milliseconds -- what the serial walk costs on a large clip
    long long unc_budget = 4000000;
    if (const char *e = getenv("MM_LINK_UNCERTAIN_BUDGET")) unc_budget = atoll(e);
    LinkWs w;
    ParWs pw;
    carve(w, (uint8_t *)ws, n_frames, cap);
    carve_par(pw, (uint8_t *)ws, n_frames, cap);
    int slots = 64;
    while (slots < 2 * cap) slots *= 2;      // <= 16384 for cap <= LK_MAX_CAP: 64 KB of LDS
    // (the attribute belongs to the device's code object: set per call -- a host-side table lookup -- rather than remembered
    // in a per-process flag that a second device would never see)
    if ((size_t)slots * sizeof(int32_t) > 48 * 1024)
        MM_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(link_canon_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        16384 * (int)sizeof(int32_t)));
    MM_LAUNCH(ctx, "link_canon_kernel", link_canon_kernel, dim3(n_frames), dim3(CANON_THREADS), (size_t)slots * sizeof(int32_t), cap,
              slots, kp_count, kp_xy, serial ? w.canon : pw.canon);
    if (!serial) {
        const int64_t n_match = (int64_t)(n_frames - 1) * cap, n_nodes = (int64_t)n_frames * cap;
        const dim3 gm((unsigned)((n_match + 255) / 256)), gn((unsigned)((n_nodes + 255) / 256)), b256(256);
        MM_HIP(ctx, hipMemsetAsync(ws, 0xFF, pw.ff_bytes, ctx->stream));
        MM_HIP(ctx, hipMemsetAsync((uint8_t *)ws + pw.zero_off, 0, pw.zero_bytes, ctx->stream));
        MM_LAUNCH(ctx, "lp_edges_kernel", lp_edges_kernel, gm, b256, 0, n_match, cap, kp_count, match_count, matches, pw);
        MM_LAUNCH(ctx, "lp_certain_kernel", lp_certain_kernel, gm, b256, 0, n_match, cap, match_count, pw);
        MM_LAUNCH(ctx, "lp_uncertain_kernel", lp_uncertain_kernel, dim3(1), dim3(1024), 0, n_frames, pw, unc_budget);
        {   // safety valve: one 4-byte read-back (the call is followed by a read-back of `counts` anyway)
            int32_t gave_up = 0;
            MM_HIP(ctx, hipMemcpyAsync(&gave_up, pw.scal + 2, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            MM_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->link_last_variant = gave_up ? 0 : 1;
            if (gave_up) {
                serial = true;      // (the two formulations share the workspace: start over with the serial one's layout)
                MM_LAUNCH(ctx, "link_canon_kernel", link_canon_kernel, dim3(n_frames), dim3(CANON_THREADS), (size_t)slots * sizeof(int32_t),
                          cap, slots, kp_count, kp_xy, w.canon);
            }
        }
      if (!serial) {
        MM_LAUNCH(ctx, "lp_classify_kernel", lp_classify_kernel, gm, b256, 0, n_match, cap, match_count, pw);
        int cur = 0;
        for (int64_t reach = 1; reach < n_frames; reach *= 2, cur ^= 1)      // chains have at most n_frames - 1 edges
            MM_LAUNCH(ctx, "lp_jump_kernel", lp_jump_kernel, gn, b256, 0, n_nodes, pw.birth, (const int32_t *)pw.jump[cur], pw.jump[cur ^ 1]);
        MM_LAUNCH(ctx, "lp_final_kernel", lp_final_kernel, gm, b256, 0, n_match, cap, match_count, pw);
        MM_LAUNCH(ctx, "lp_sort_kernel", lp_sort_kernel, dim3(n_frames), dim3(LK_THREADS), 0, cap, pw);
        MM_LAUNCH(ctx, "lp_frames_kernel", lp_frames_kernel, dim3(1), dim3(LK_THREADS), 0, n_frames, cap, pw, track_ptr, counts);
        MM_LAUNCH(ctx, "lp_ptr_kernel", lp_ptr_kernel, dim3((unsigned)((cap + 255) / 256), n_frames), b256, 0, cap, pw, track_ptr);
        MM_LAUNCH(ctx, "lp_emit_kernel", lp_emit_kernel, gm, b256, 0, n_match, cap, match_count, matches, pw, obs_frame, obs_kp);
        return MM_OK;
      }
    } else {
        ctx->link_last_variant = 0;
    }
    const int lds_mode = cap <= 4096 ? 2 : 1;      // (cap <= LK_MAX_CAP = 8192: the three tables always fit)
    const size_t lds_bytes = (size_t)(lds_mode == 2 ? 9 : 3) * cap * sizeof(int32_t);
    if (lds_bytes > 48 * 1024) {
        const void *fn = lds_mode == 2 ? reinterpret_cast<const void *>(link_kernel<2>) : reinterpret_cast<const void *>(link_kernel<1>);
        MM_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    if (lds_mode == 2)
        MM_LAUNCH(ctx, "link_kernel", link_kernel<2>, dim3(1), dim3(LK_THREADS), lds_bytes, n_frames, cap, kp_count,
                  match_count, matches, w, track_ptr, counts);
    else
        MM_LAUNCH(ctx, "link_kernel", link_kernel<1>, dim3(1), dim3(LK_THREADS), lds_bytes, n_frames, cap, kp_count,
                  match_count, matches, w, track_ptr, counts);
    const size_t max_tracks = (size_t)(n_frames - 1) * cap;
    MM_LAUNCH(ctx, "link_emit_kernel", link_emit_kernel, dim3((unsigned)((max_tracks + 255) / 256)), dim3(256), 0, w,
              (const int32_t *)track_ptr, (const int64_t *)counts, obs_frame, obs_kp);
    return MM_OK;
}

}  // extern "C"
