// The whole trust-region solve of adjustPoints behind ONE call (single GPU): mm_ba_trf.
//
// Replaces the loop scipy.optimize.least_squares runs for bundleAdjuster.adjustPoints (reference
// bundleAdjuster.py:180-192: method='trf', jac_sparsity, x_scale='jac', linear loss -> scipy/optimize/_lsq/trf.py
// trf_no_bounds).  The iteration is the one meatmodeler_amd/bundleAdjuster.py::_solve_device sequences from Python --
// same kernels, same order, same scalars, bit-identical iterates -- driven from C++: the host side of an iteration
// shrinks from ~0.45 ms of interpreter time to a few microseconds, which is what bounds the small systems of the
// sliding-window adjustment (300 unknowns per window: 0.39 ms per evaluation from Python, launch-bound here), and the
// library itself now owns the loop SURVEY.md section 8(b) calls mm_ba_lm.  One host synchronisation per trial step (a
// 16-double board copied to the host after the trial cost is known).  Everything lives in the caller's workspace; no
// allocation, no second stream unless the overlapped build is requested.
#include "mm_common.h"
#include <chrono>
#include <vector>
#include <algorithm>
#include <initializer_list>
#include <utility>
#include <cmath>
#include <cstdlib>

namespace {

__global__ __launch_bounds__(256) void vec_mul_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                      double *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

// The scalars the host needs after a trial step go through a pinned host mailbox: this kernel copies them there and then
// raises the sequence number (system scope); the host spins on that word.  Compared with hipMemcpyAsync +
// hipStreamSynchronize (interrupt-driven wake-up) the gap between two iterations shrinks from ~30 us to the launch latency.
struct HostBoard {
    double v[16];
    unsigned long long seq;
};

__global__ __launch_bounds__(64) void board_publish_kernel(const double *__restrict__ dev, int count, HostBoard *hb,
                                                          unsigned long long seq) {
    if ((int)threadIdx.x < count) hb->v[threadIdx.x] = dev[threadIdx.x];
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (threadIdx.x == 0) __hip_atomic_store(&hb->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the sharded loop (mm_ba_trf_dist): points partitioned over ranks, cameras replicated --------------------------------
// Every sum over observations / points is a partial sum on a rank.  The scalars among them travel in ONE 64-double
// exchange vector per synchronisation point (all-reduce = sum): a packing kernel collects the ranks' contributions, the
// unpacking kernel writes the totals where the next kernel reads them (the "total" column of the result rows), so that
// all ranks feed identical scalars to identical kernels and take identical decisions.
//   kind 0: a result row of a pass over the PARAMETER vector {camera part, point part, total}: the camera part is
//           replicated (and its reduction tree depends on the local vector length: only rank 0's copy enters the sum);
//   kind 1: a result row of inner products of RESIDUAL-space vectors: the local total enters;
//   kind 2: a maximum {camera part, point part, total}: every rank puts its point part into its own slot (a sum over
//           ranks then gathers them) and the maximum is taken after the exchange;
//   kind 3: a plain scalar;  kind 4: "info < 0 on this rank" -> 1 in the rank's slot, the sum of the slots comes back.
constexpr int EX_MAX_ITEMS = 12, EX_SCALARS = 64, EX_MAX_WORLD = 16;
struct ExItems {
    double *p[EX_MAX_ITEMS];
    int kind[EX_MAX_ITEMS];
    int n, rank, world;
    const int32_t *info;
};
__global__ __launch_bounds__(64) void ex_pack_kernel(ExItems it, double *__restrict__ ex) {
    const int t = threadIdx.x;
    double v = 0.0;
    if (t < it.n) {
        const double *r = it.p[t];
        const int k = it.kind[t];
        if (k == 0) v = it.rank == 0 ? r[0] + r[1] : r[1];
        else if (k == 1) v = r[2];
        else if (k == 3) v = r[0];
    }
    ex[t] = v;
    __syncthreads();
    // slot items: 16 slots each behind the plain items, in item order
    if (t == 0) {
        int slot = 16;
        for (int i = 0; i < it.n; ++i) {
            if (it.kind[i] == 2) { ex[slot + it.rank] = it.p[i][1]; slot += EX_MAX_WORLD; }
            if (it.kind[i] == 4) { ex[slot + it.rank] = it.info[0] < 0 ? 1.0 : 0.0; slot += EX_MAX_WORLD; }
        }
    }
}
__global__ __launch_bounds__(64) void ex_unpack_kernel(ExItems it, const double *__restrict__ ex) {
    const int t = threadIdx.x;
    if (t < it.n) {
        double *r = it.p[t];
        const int k = it.kind[t];
        if (k == 0 || k == 1) r[2] = ex[t];
        else if (k == 3) r[0] = ex[t];
    }
    if (t == 0) {
        int slot = 16;
        for (int i = 0; i < it.n; ++i) {
            if (it.kind[i] == 2) {
                double m = it.p[i][0];
                for (int q = 0; q < it.world; ++q) m = fmax(m, ex[slot + q]);
                it.p[i][2] = m;
                slot += EX_MAX_WORLD;
            }
            if (it.kind[i] == 4) {
                double sum = 0.0;
                for (int q = 0; q < it.world; ++q) sum += ex[slot + q];
                it.p[i][0] = sum;
                slot += EX_MAX_WORLD;
            }
        }
    }
}
// the lower band of S as a contiguous [n, hb + 1] array: entry [j, k] = S[j + k][j] (zero past the last row)
__global__ __launch_bounds__(256) void band_pack_kernel(const double *__restrict__ S, int64_t n, int hb, double *__restrict__ band) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, total = n * (hb + 1);
    if (i >= total) return;
    const int64_t j = i / (hb + 1), k = i % (hb + 1);
    band[i] = j + k < n ? S[(j + k) * n + j] : 0.0;
}
// ... and back, with the duplicates removed: every rank added the full blockdiag(Bd) to its S and the full g_c to its v
__global__ __launch_bounds__(256) void band_unpack_kernel(double *__restrict__ S, int64_t n, int hb, const double *__restrict__ band,
                                                          const double *__restrict__ Bd, double dup) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, total = n * (hb + 1);
    if (i >= total) return;
    const int64_t j = i / (hb + 1), k = i % (hb + 1), row = j + k;
    if (row >= n) return;
    double v = band[i];
    if (row / 6 == j / 6) v -= dup * Bd[(j / 6) * 36 + (row % 6) * 6 + j % 6];
    S[row * n + j] = v;
}
__global__ __launch_bounds__(256) void dedup_dense_kernel(double *__restrict__ S, int64_t n, const double *__restrict__ Bd, double dup) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one thread per entry of blockdiag(Bd): F * 36
    if (i >= n * 6) return;
    const int64_t f = i / 36, a = (i % 36) / 6, b = i % 6;
    S[(6 * f + a) * n + 6 * f + b] -= dup * Bd[i];
}
__global__ __launch_bounds__(256) void axpy_kernel(double *__restrict__ y, const double *__restrict__ x, double a, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}

struct Carve {
    char *p;
    size_t used = 0;
    explicit Carve(void *base) : p((char *)base) {}
    template <class T>
    T *take(size_t count) {
        T *r = p ? (T *)(p + used) : nullptr;
        used += mm_align_up(count * sizeof(T), 256);
        return r;
    }
};

struct TrfWs {
    double *x, *x_new, *g, *si, *gh, *ghs, *gn, *q1, *w, *q2, *s1, *s2;
    double *B, *Bd, *C, *Cd, *Cinv, *u1, *Jq2, *dp, *v, *S;
    double *r0, *r1, *r2, *r3, *d11, *bs, *damp, *board, *cost2, *rowsx;
    int32_t *info;
    void *ws_res, *ws_md, *ws_jvp, *ws_schur, *ws_chol, *ws_back;
    size_t ws_res_b, ws_md_b, ws_jvp_b, ws_schur_b, ws_chol_b, ws_back_b;
    double *exs, *ex;      // sharded loop: the 64-double scalar exchange vector, the bulk exchange buffer (B | g_c, band | v)
    size_t ex_count;
    size_t total;
};

TrfWs carve_trf(const mm_ba_problem *pb, void *base, int64_t ex_half_bw = -1) {
    TrfWs t;
    Carve c(base);
    const size_t F = pb->F, P = pb->P, O = (size_t)pb->O, n = 6 * F + 3 * P, nc = 6 * F;
    double **vecs[12] = {&t.x, &t.x_new, &t.g, &t.si, &t.gh, &t.ghs, &t.gn, &t.q1, &t.w, &t.q2, &t.s1, &t.s2};
    for (auto v : vecs) *v = c.take<double>(n + 2);
    t.B = c.take<double>(F * 36);
    t.Bd = c.take<double>(F * 36);
    t.C = c.take<double>(P * 6 + 2);
    t.Cd = c.take<double>(P * 6 + 2);
    t.Cinv = c.take<double>(P * 6 + 2);
    t.u1 = c.take<double>(O * 2 + 2);
    t.Jq2 = c.take<double>(O * 2 + 2);
    t.dp = c.take<double>(P * 3 + 2);
    t.v = c.take<double>(nc + 2);
    t.S = c.take<double>(2 * nc * nc + 2);   // (+ the zero padding the band view of the sharded path expects)
    t.r0 = c.take<double>(6);
    t.r1 = c.take<double>(9);
    t.r2 = c.take<double>(6);
    t.r3 = c.take<double>(18);
    t.d11 = c.take<double>(6);
    t.bs = c.take<double>(6);
    t.damp = c.take<double>(2);
    t.board = c.take<double>(16);
    t.cost2 = c.take<double>(1);
    t.rowsx = c.take<double>(3);
    t.info = c.take<int32_t>(1);
    t.ws_res_b = 2048 * 8;
    t.ws_res = c.take<char>(t.ws_res_b);
    t.ws_md_b = mm_multi_dot_workspace_bytes();
    t.ws_md = c.take<char>(t.ws_md_b);
    t.ws_jvp_b = mm_ba_jvp_dots_workspace_bytes(pb);
    t.ws_jvp = c.take<char>(t.ws_jvp_b);
    t.ws_schur_b = mm_ba_schur_workspace_bytes(pb);
    if (t.ws_schur_b < 256) t.ws_schur_b = 256;
    t.ws_schur = c.take<char>(t.ws_schur_b);
    t.ws_chol_b = mm_chol_workspace_bytes((int)nc);
    t.ws_chol = c.take<char>(t.ws_chol_b);
    t.ws_back_b = mm_ba_backsub_workspace_bytes(pb);
    t.ws_back = c.take<char>(t.ws_back_b);
    t.exs = nullptr;
    t.ex = nullptr;
    t.ex_count = 0;
    if (ex_half_bw >= 0) {      // (mm_ba_trf_dist)
        const size_t hb = (size_t)(ex_half_bw < (int64_t)nc - 1 ? ex_half_bw : (nc ? nc - 1 : 0));
        t.ex_count = nc * (hb + 1) + nc;
        if (t.ex_count < F * 42) t.ex_count = F * 42;
        t.exs = c.take<double>(EX_SCALARS);
        t.ex = c.take<double>(t.ex_count + 2);
    }
    t.total = c.used;
    return t;
}

// SciPy's update_tr_radius / check_termination (scipy/optimize/_lsq/common.py:222-245, 705-717)
void update_tr_radius(double &Delta, double actual, double predicted, double step_norm, bool bound_hit, double &ratio) {
    if (predicted > 0)
        ratio = actual / predicted;
    else if (predicted == 0 && actual == 0)
        ratio = 1;
    else
        ratio = 0;
    if (ratio < 0.25)
        Delta = 0.25 * step_norm;
    else if (ratio > 0.75 && bound_hit)
        Delta *= 2.0;
}
int check_termination(double dF, double F, double dx_norm, double x_norm, double ratio, double ftol, double xtol) {
    const bool ftol_ok = dF < ftol * F && ratio > 0.25;
    const bool xtol_ok = dx_norm < xtol * (xtol + x_norm);
    if (ftol_ok && xtol_ok) return 4;
    if (ftol_ok) return 2;
    if (xtol_ok) return 3;
    return -100;  // (None)
}

}  // namespace

extern "C" size_t mm_ba_trf_workspace_bytes(const mm_ba_problem *pb) {
    if (!pb || pb->F < 0 || pb->P < 0 || pb->O < 0) return 0;
    return carve_trf(pb, nullptr).total;
}

#define TRF_CALL(expr)        \
    do {                      \
        int rc_ = (expr);     \
        if (rc_) return rc_;  \
    } while (0)

extern "C" size_t mm_ba_trf_dist_workspace_bytes(const mm_ba_problem *pb, int half_bandwidth) {
    if (!pb || pb->F < 0 || pb->P < 0 || pb->O < 0 || half_bandwidth < 0) return 0;
    return carve_trf(pb, nullptr, half_bandwidth).total;
}

static int trf_run(mm_ctx *ctx, const mm_ba_problem *pb, double *cams, double *pts, const mm_trf_params *prm, mm_trf_report *rep,
                   mm_trf_row *log, int log_cap, void *ws, size_t ws_bytes, const mm_dist *dist) {
    const auto t_enter = std::chrono::steady_clock::now();
    if (!ctx) return MM_ERR_ARG;
    if (!pb || !cams || !pts || !prm || !rep || pb->F <= 0 || pb->P < 0 || pb->O < 0 || !pb->K || log_cap < 0 || (log_cap > 0 && !log))
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_trf: bad argument");
    if (dist && (dist->world < 1 || dist->world > EX_MAX_WORLD || dist->rank < 0 || dist->rank >= dist->world || !dist->allreduce ||
                 dist->half_bandwidth < 0))
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_trf_dist: bad communicator description (1 <= world <= %d)", EX_MAX_WORLD);
    const size_t need = dist ? mm_ba_trf_dist_workspace_bytes(pb, dist->half_bandwidth) : mm_ba_trf_workspace_bytes(pb);
    if (!ws || ws_bytes < need || ((uintptr_t)ws & 255))
        return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_ba_trf: workspace too small or misaligned");
    const TrfWs t = carve_trf(pb, ws, dist ? dist->half_bandwidth : -1);
    // the sweeps share one table of per-camera rotation coefficients: this loop knows when the camera vector behind a
    // pointer changes (only the trial point is ever rewritten), so the table is rebuilt once per trial point
    struct TableHold {
        mm_ctx *c;
        explicit TableHold(mm_ctx *ctx) : c(ctx) { mm_cam_table_hold(c, true); }
        ~TableHold() { mm_cam_table_hold(c, false); }
    } table_hold(ctx);
    // an abandoned single-launch factorisation (info = -1: its workgroups were not co-resident) switches this solve to
    // the launch-per-column factorisation; the context's setting is restored on the way out, its budget share returned
    struct FusedGuard {
        mm_ctx *c;
        bool saved, saved_strict, saved_no_event;
        FusedGuard(mm_ctx *ctx, bool sharded)
            : c(ctx), saved(ctx->chol_avoid_fused), saved_strict(ctx->chol_strict_budget), saved_no_event(ctx->fused_no_event) {
            c->fused_no_event = true;      // (the share goes back at every read-back of this loop)
            // sharded: no rank-local choice of the factorisation path (a full budget reports info = -1 instead, and the
            // "a rank's factorisation was abandoned" flag of the trial-cost exchange moves all ranks together)
            if (sharded) c->chol_strict_budget = true;
        }
        ~FusedGuard() {
            c->chol_avoid_fused = saved;
            c->chol_strict_budget = saved_strict;
            c->fused_no_event = saved_no_event;
            mm_chol_release_budget(c);
        }
    } fused_guard(ctx, dist != nullptr);
    const int F = pb->F, P = pb->P;
    const int64_t nc = 6 * (int64_t)F, n = nc + 3 * (int64_t)P;
    hipStream_t st = ctx->stream;
    auto cams_of = [&](double *v) { return v; };
    auto pts_of = [&](double *v) { return v + nc; };
    // the reduction workspaces count arrivals: zero once
    MM_HIP(ctx, hipMemsetAsync(t.ws_md, 0, t.ws_md_b, st));
    MM_HIP(ctx, hipMemsetAsync(t.ws_jvp, 0, t.ws_jvp_b, st));
    MM_HIP(ctx, hipMemsetAsync(t.S, 0, (size_t)(2 * nc * nc) * sizeof(double), st));
    double *x = t.x, *x_new = t.x_new;
    MM_HIP(ctx, hipMemcpyAsync(x, cams, (size_t)nc * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (P) MM_HIP(ctx, hipMemcpyAsync(x + nc, pts, (size_t)3 * P * sizeof(double), hipMemcpyDeviceToDevice, st));
    double host[16];
    rep->chol_fallbacks = 0;
    rep->collectives = 0;
    // ---- sharded: the exchange points (see ExItems) ----
    auto ar = [&](double *buf, int64_t count) -> int {
        ++rep->collectives;
        const int rc = dist->allreduce(dist->user, buf, count);
        return rc ? mm_fail(ctx, MM_ERR_HIP, "mm_ba_trf_dist: the all-reduce callback failed (%d)", rc) : MM_OK;
    };
    auto exchange = [&](std::initializer_list<std::pair<double *, int>> items) -> int {
        if (!dist) return MM_OK;
        ExItems it = {};
        it.rank = dist->rank;
        it.world = dist->world;
        it.info = t.info;
        for (auto &pr : items) {
            it.p[it.n] = pr.first;
            it.kind[it.n] = pr.second;
            ++it.n;
        }
        hipLaunchKernelGGL(ex_pack_kernel, dim3(1), dim3(64), 0, st, it, t.exs);
        TRF_CALL(ar(t.exs, EX_SCALARS));
        hipLaunchKernelGGL(ex_unpack_kernel, dim3(1), dim3(64), 0, st, it, (const double *)t.exs);
        MM_LAUNCH_CHECK(ctx, "ex_unpack_kernel");
        return MM_OK;
    };
    // block normal equations at v: B and g_c are sums over ALL observations
    auto normal_eq = [&](double *v) -> int {
        TRF_CALL(mm_ba_normal_eq(ctx, pb, cams_of(v), pts_of(v), t.B, cams_of(t.g), t.C, pts_of(t.g)));
        if (!dist) return MM_OK;
        MM_HIP(ctx, hipMemcpyAsync(t.ex, t.B, (size_t)F * 36 * sizeof(double), hipMemcpyDeviceToDevice, st));
        MM_HIP(ctx, hipMemcpyAsync(t.ex + (size_t)F * 36, cams_of(t.g), (size_t)nc * sizeof(double), hipMemcpyDeviceToDevice, st));
        TRF_CALL(ar(t.ex, (int64_t)F * 42));
        MM_HIP(ctx, hipMemcpyAsync(t.B, t.ex, (size_t)F * 36 * sizeof(double), hipMemcpyDeviceToDevice, st));
        MM_HIP(ctx, hipMemcpyAsync(cams_of(t.g), t.ex + (size_t)F * 36, (size_t)nc * sizeof(double), hipMemcpyDeviceToDevice, st));
        return MM_OK;
    };
    static const bool spin = !(getenv("MM_TRF_SPIN") && getenv("MM_TRF_SPIN")[0] == '0');
    if (spin && !ctx->host_board) {
        MM_HIP(ctx, hipHostMalloc(&ctx->host_board, sizeof(HostBoard), hipHostMallocDefault));
        memset(ctx->host_board, 0, sizeof(HostBoard));
        ctx->host_board_seq = 0;
    }
    auto wait_board = [&](unsigned long long seq, int count) -> int {
        HostBoard *hb = (HostBoard *)ctx->host_board;
        const auto t0 = std::chrono::steady_clock::now();
        unsigned long spins = 0;
        while (__atomic_load_n(&hb->seq, __ATOMIC_ACQUIRE) != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
                // never seen: a failed launch upstream -- let the runtime report it
                MM_HIP(ctx, hipStreamSynchronize(st));
                if (__atomic_load_n(&hb->seq, __ATOMIC_ACQUIRE) != seq)
                    return mm_fail(ctx, MM_ERR_HIP, "mm_ba_trf: the trial-step scalars never arrived");
            }
        }
        for (int i = 0; i < count; ++i) host[i] = ((volatile double *)hb->v)[i];
        mm_chol_release_budget(ctx);      // the stream has passed everything enqueued before the publishing kernel
        return MM_OK;
    };
    auto read_board = [&](const double *dev, int count) -> int {
        if (!spin) {
            MM_HIP(ctx, hipMemcpyAsync(host, dev, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, st));
            MM_HIP(ctx, hipStreamSynchronize(st));
            mm_chol_release_budget(ctx);
            return MM_OK;
        }
        HostBoard *hb = (HostBoard *)ctx->host_board;
        const unsigned long long seq = ++ctx->host_board_seq;
        hipLaunchKernelGGL(board_publish_kernel, dim3(1), dim3(64), 0, st, dev, count, hb, seq);
        MM_LAUNCH_CHECK(ctx, "board_publish_kernel");
        return wait_board(seq, count);
    };
    // initial cost
    TRF_CALL(mm_ba_residual(ctx, pb, cams_of(x), pts_of(x), nullptr, t.cost2, t.ws_res, t.ws_res_b));
    if (dist) {
        // "this rank's context is set to avoid the single-launch factorisation" rides with the initial cost: if any rank
        // is, all are -- the two paths order their sums differently and the replicated cameras must stay bit-identical
        const double av = ctx->chol_avoid_fused ? 1.0 : 0.0;
        MM_HIP(ctx, hipMemcpyAsync(t.board + 15, &av, sizeof(double), hipMemcpyHostToDevice, st));
        MM_HIP(ctx, hipStreamSynchronize(st));
        TRF_CALL(exchange({{t.cost2, 3}, {t.board + 15, 3}}));
        TRF_CALL(read_board(t.board + 15, 1));
        if (host[0] > 0) ctx->chol_avoid_fused = true;
    }
    TRF_CALL(read_board(t.cost2, 1));
    double cost = 0.5 * host[0];
    rep->cost0 = cost;
    if (!std::isfinite(cost)) {
        rep->status = -2;
        return mm_fail(ctx, MM_ERR_NUMERIC, "mm_ba_trf: residuals are not finite in the initial point");
    }
    const int half_bw = dist ? dist->half_bandwidth : 6 * pb->cam_span + 5;
    int nfev = 1, njev = 1;
    TRF_CALL(normal_eq(x));
    TRF_CALL(mm_ba_scale_update(ctx, F, P, t.B, t.C, t.si, 1));
    {   // Delta0 = |x * scale_inv|  (trf.py:428)
        hipLaunchKernelGGL(vec_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double *)x, (const double *)t.si, t.w, n);
        const double *pa[1] = {t.w}, *pbv[1] = {t.w};
        TRF_CALL(mm_multi_dot(ctx, 1, pa, pbv, n, nc, t.rowsx, t.ws_md, t.ws_md_b));
        TRF_CALL(exchange({{t.rowsx, 0}}));
        TRF_CALL(read_board(t.rowsx, 3));
    }
    double Delta = std::sqrt(host[2]);
    if (Delta == 0) Delta = 1.0;
    const long max_nfev = prm->max_nfev > 0 ? prm->max_nfev : (long)n * 100;
    double min_damping = prm->min_damping > 0 ? prm->min_damping : 1e-9;
    double alpha = 0.0;
    (void)alpha;
    int termination = -100, iteration = 0, n_log = 0;
    double step_norm = NAN, actual = NAN, g_norm = NAN;
    auto log_row = [&](int it, int nf, double c, double red, double stepn, double opt) {
        if (n_log < log_cap) log[n_log] = mm_trf_row{it, nf, c, red, stepn, opt};
        ++n_log;
    };
    static const bool timing = getenv("MM_TRF_TIMING") && atoi(getenv("MM_TRF_TIMING")) > 0;      // wall clock per iteration on stderr
    const auto t_loop = std::chrono::steady_clock::now();
    auto t_iter = t_loop;
    if (timing) fprintf(stderr, "mm_ba_trf timing: prologue %.1f us\n", std::chrono::duration<double, std::micro>(t_loop - t_enter).count());
    for (;;) {
        if (timing && iteration > 0) {
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "mm_ba_trf timing: iteration %d %.1f us (nfev %d)\n", iteration - 1,
                    std::chrono::duration<double, std::micro>(now - t_iter).count(), nfev);
            t_iter = now;
        }
        if (!dist && iteration > 0) {      // (the running maximum of the Jacobian scaling rides with the pass: idempotent)
            TRF_CALL(mm_trf_fused0_scaled(ctx, t.g, t.si, t.B, t.C, t.gh, t.ghs, n, nc, t.r0, t.ws_md, t.ws_md_b));
        } else {
            const double *in[2] = {t.g, t.si};
            double *outv[2] = {t.gh, t.ghs};
            TRF_CALL(mm_trf_fused(ctx, 0, in, outv, nullptr, 0, 0, n, nc, t.r0, t.ws_md, t.ws_md_b));
        }
        TRF_CALL(mm_ba_jvp_dots(ctx, pb, cams_of(x), pts_of(x), cams_of(t.ghs), pts_of(t.ghs), t.u1, nullptr, t.d11, t.ws_jvp, t.ws_jvp_b));
        TRF_CALL(exchange({{t.r0, 0}, {t.r0 + 3, 2}, {t.d11, 1}, {t.d11 + 3, 1}}));      // |g_h|^2, |g|_inf, |J d g_h|^2
        if (termination != -100 || nfev == max_nfev) {
            TRF_CALL(read_board(t.r0, 6));
            g_norm = host[5];
            log_row(iteration, nfev, cost, actual, step_norm, g_norm);
            break;
        }
        // (the damping of the iteration is computed by the first damped-blocks sweep itself: mm_trf_damping's formula, one
        // launch less; a retry with raised damping reads damp[1] as the host left it)
        double *reg_eff = t.damp + 1;
        bool solved = false, rows_pending = false;
        auto trial = [&](double Delta_) -> int {
            if (rows_pending) {      // first trial of the attempt: the sums of J s2 are still per-workgroup partials
                rows_pending = false;
                TRF_CALL(mm_trf_rows_step2d(ctx, ctx->jvp_partial, ctx->jvp_n_wg, t.bs, t.r0, t.d11, t.r1, t.r2, t.r3, reg_eff, t.info, Delta_,
                                            t.board));
            } else {
                TRF_CALL(mm_trf_step2d(ctx, t.r0, t.d11, t.r1, t.r2, t.r3, t.bs, reg_eff, t.info, Delta_, t.board));
            }
            if (F > 0 && !dist) {      // x_new and its cameras' rotation coefficients in one launch
                void *ctab = nullptr;
                TRF_CALL(mm_cam_table_adopt(ctx, cams_of(x_new), (int)F, &ctab));
                TRF_CALL(mm_trf_fused5_coef(ctx, x, t.s1, t.s2, x_new, t.board, n, nc, ctab, (int)F));
            } else {
                const double *in[3] = {x, t.s1, t.s2};
                double *outv[1] = {x_new};
                const double *sc[1] = {t.board};
                TRF_CALL(mm_trf_fused(ctx, 5, in, outv, sc, 0, 0, n, nc, nullptr, t.ws_md, t.ws_md_b));
                mm_cam_table_invalidate(ctx);      // x_new has new contents
            }
            if (spin && !dist) {      // the residual's final sum and the hand-over to the host mailbox are one launch
                const unsigned long long seq = ++ctx->host_board_seq;
                TRF_CALL(mm_ba_residual_publish(ctx, pb, cams_of(x_new), pts_of(x_new), t.ws_res, t.ws_res_b, t.board, 14, 16,
                                                ctx->host_board, seq));
                return wait_board(seq, 16);   // ---- the host sync of a trial step ----
            }
            TRF_CALL(mm_ba_residual(ctx, pb, cams_of(x_new), pts_of(x_new), nullptr, t.board + 14, t.ws_res, t.ws_res_b));
            // sharded: the trial cost is a sum over all ranks; and whether ANY rank's factorisation was abandoned (a
            // rank-local event) comes back in board[15], so that all ranks fall back together
            TRF_CALL(exchange({{t.board + 14, 3}, {t.board + 15, 4}}));
            return read_board(t.board, 16);
        };
        bool damping_known = false;
        for (int attempt = 0; attempt < 6 && !solved; ++attempt) {
            if (!dist) {
                // damping, reduced system, factorisation and both substitutions: four launches (schur.hip)
                mm_damp_spec dmp = {t.B, t.C, t.si, nullptr, nullptr, Delta, min_damping, t.damp, reg_eff};
                if (!damping_known) {
                    dmp.gh2 = t.r0 + 2;
                    dmp.d11 = t.d11 + 2;
                    damping_known = true;
                }
                TRF_CALL(mm_ba_schur_solve_damped(ctx, pb, cams_of(x), pts_of(x), &dmp, t.Bd, t.Cd, cams_of(t.g), pts_of(t.g), t.S, t.v, t.Cinv,
                                                  half_bw, t.info, t.ws_schur, t.ws_schur_b, t.ws_chol, t.ws_chol_b));
            } else {
                if (!damping_known) {
                    TRF_CALL(mm_ba_damp_damping(ctx, F, P, t.B, t.C, t.si, t.r0 + 2, t.d11 + 2, Delta, min_damping, t.damp, t.Bd, t.Cd));
                    damping_known = true;
                } else {
                    TRF_CALL(mm_ba_damp(ctx, F, P, t.B, t.C, t.si, reg_eff, t.Bd, t.Cd));
                }
                // every rank's S / v hold its points' share plus the FULL blockdiag(Bd) / g_c: sum, then remove the
                // duplicates.  Band exchange (decided by the caller from global quantities): n (hb + 1) + n doubles in one
                // collective instead of the dense matrix; only the lower band then holds the sum.
                TRF_CALL(mm_ba_schur(ctx, pb, cams_of(x), pts_of(x), t.Bd, t.Cd, cams_of(t.g), pts_of(t.g), t.S, t.v, t.Cinv, t.ws_schur,
                                     t.ws_schur_b));
                const double dup = (double)(dist->world - 1);
                if (dist->band_exchange) {
                    const int hb = (int)(half_bw < nc - 1 ? half_bw : nc - 1);
                    const int64_t cnt = nc * (hb + 1);
                    hipLaunchKernelGGL(band_pack_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, (const double *)t.S, nc, hb, t.ex);
                    MM_HIP(ctx, hipMemcpyAsync(t.ex + cnt, t.v, (size_t)nc * sizeof(double), hipMemcpyDeviceToDevice, st));
                    TRF_CALL(ar(t.ex, cnt + nc));
                    hipLaunchKernelGGL(band_unpack_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, t.S, nc, hb,
                                       (const double *)t.ex, (const double *)t.Bd, dup);
                    MM_HIP(ctx, hipMemcpyAsync(t.v, t.ex + cnt, (size_t)nc * sizeof(double), hipMemcpyDeviceToDevice, st));
                } else {
                    // (v rides behind the matrix: the buffer of S has room for it, one collective instead of two)
                    MM_HIP(ctx, hipMemcpyAsync(t.S + nc * nc, t.v, (size_t)nc * sizeof(double), hipMemcpyDeviceToDevice, st));
                    TRF_CALL(ar(t.S, nc * nc + nc));
                    MM_HIP(ctx, hipMemcpyAsync(t.v, t.S + nc * nc, (size_t)nc * sizeof(double), hipMemcpyDeviceToDevice, st));
                    hipLaunchKernelGGL(dedup_dense_kernel, dim3((unsigned)((nc * 6 + 255) / 256)), dim3(256), 0, st, t.S, nc,
                                       (const double *)t.Bd, dup);
                }
                hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, t.v, (const double *)cams_of(t.g), -dup, nc);
                MM_LAUNCH_CHECK(ctx, "axpy_kernel");
                TRF_CALL(mm_chol_solve_sym(ctx, t.S, (int)nc, t.v, half_bw, dist->band_exchange ? 0 : 1, t.info, t.ws_chol, t.ws_chol_b));
            }
            TRF_CALL(mm_ba_backsub(ctx, pb, cams_of(x), pts_of(x), t.Cinv, pts_of(t.g), t.v, t.dp, t.ws_back, t.ws_back_b));
            {
                const double *in[4] = {t.v, t.dp, t.si, t.gh};
                double *outv[2] = {t.gn, t.q1};
                const double *sc[1] = {t.r0 + 2};
                TRF_CALL(mm_trf_fused(ctx, 1, in, outv, sc, 0, 0, n, nc, t.r1, t.ws_md, t.ws_md_b));
                TRF_CALL(exchange({{t.r1, 0}, {t.r1 + 3, 0}}));
            }
            {
                const double *in[2] = {t.gn, t.q1};
                double *outv[1] = {t.w};
                const double *sc[1] = {t.r1 + 2};
                TRF_CALL(mm_trf_fused(ctx, 2, in, outv, sc, 0, 0, n, nc, t.r2, t.ws_md, t.ws_md_b));
                TRF_CALL(exchange({{t.r2, 0}}));
            }
            {
                const double *in[5] = {t.w, t.q1, t.si, t.gh, x};
                double *outv[3] = {t.q2, t.s1, t.s2};
                const double *sc[1] = {t.r2 + 2};
                TRF_CALL(mm_trf_fused(ctx, 3, in, outv, sc, 0, 0, n, nc, t.r3, t.ws_md, t.ws_md_b));
            }
            ctx->jvp_rows_deferred = !dist && P > 0 && pb->O > 0;      // (sharded: the sums are exchanged first)
            {
                const int rc_ = mm_ba_jvp_dots(ctx, pb, cams_of(x), pts_of(x), cams_of(t.s2), pts_of(t.s2), t.Jq2, t.u1, t.bs, t.ws_jvp, t.ws_jvp_b);
                rows_pending = ctx->jvp_rows_deferred;
                ctx->jvp_rows_deferred = false;
                if (rc_) return rc_;
            }
            // (the step inner products and the two of J s2 travel together: the Jacobian product only needs s2)
            TRF_CALL(exchange({{t.r3, 0}, {t.r3 + 3, 0}, {t.r3 + 6, 0}, {t.r3 + 9, 0}, {t.r3 + 12, 0}, {t.bs, 1}, {t.bs + 3, 1}}));
            TRF_CALL(trial(Delta));   // enqueued before the host knows whether the factorisation succeeded
            const int inf = dist && host[15] > 0 ? -1 : (int)host[6];
            if (inf == 0) {
                solved = true;
                break;
            }
            if (inf < 0) {
                // the single-launch factorisation gave up waiting (another tenant on the GPU, a profiler serialising
                // kernels): repeat this attempt -- same damping -- with the launch-per-column factorisation and stay there
                if (ctx->chol_avoid_fused) return mm_fail(ctx, MM_ERR_HIP, "mm_ba_trf: the banded factorisation reported info = -1");
                ctx->chol_avoid_fused = true;
                ++rep->chol_fallbacks;
                --attempt;
                continue;
            }
            if (host[13] <= min_damping * (1.0 + 1e-12)) min_damping *= 100.0;   // failed AT the floor: the floor was too low
            // reg_eff *= 100 on the device: damp[1] is a plain double
            double r100 = host[13] * 100.0;
            MM_HIP(ctx, hipMemcpyAsync(t.damp + 1, &r100, sizeof(double), hipMemcpyHostToDevice, st));
            MM_HIP(ctx, hipStreamSynchronize(st));
        }
        if (!solved) return mm_fail(ctx, MM_ERR_NUMERIC, "mm_ba_trf: reduced camera system is not positive definite (pivot %d)", (int)host[6]);
        g_norm = host[10];
        const double xx = host[9];
        if (g_norm < prm->gtol) termination = 1;   // (checked before the step is used, as trf.py:443 does)
        log_row(iteration, nfev, cost, actual, step_norm, g_norm);
        if (termination != -100) break;
        const double x_norm = std::sqrt(xx);
        actual = -1.0;
        bool have = true;
        double cost_new = cost;
        while (actual <= 0 && nfev < max_nfev) {
            if (!have) TRF_CALL(trial(Delta));
            have = false;
            const double predicted = host[2], step_h_norm = host[3], step_norm_dev = host[4];
            cost_new = 0.5 * host[14];
            ++nfev;
            if (!std::isfinite(cost_new)) {
                Delta = 0.25 * step_h_norm;
                continue;
            }
            actual = cost - cost_new;
            double Delta_new = Delta, ratio;
            update_tr_radius(Delta_new, actual, predicted, step_h_norm, step_h_norm > 0.95 * Delta, ratio);
            step_norm = step_norm_dev;
            termination = check_termination(actual, cost, step_norm, x_norm, ratio, prm->ftol, prm->xtol);
            if (termination != -100) break;
            alpha *= Delta / Delta_new;
            Delta = Delta_new;
        }
        if (actual > 0) {
            double *tmp = x;
            x = x_new;
            x_new = tmp;
            cost = cost_new;
            TRF_CALL(normal_eq(x));
            ++njev;
            if (dist) TRF_CALL(mm_ba_scale_update(ctx, F, P, t.B, t.C, t.si, 0));      // (else: with the next pass 0)
        } else {
            step_norm = 0;
            actual = 0;
        }
        ++iteration;
    }
    if (termination == -100) termination = 0;
    MM_HIP(ctx, hipMemcpyAsync(cams, x, (size_t)nc * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (P) MM_HIP(ctx, hipMemcpyAsync(pts, x + nc, (size_t)3 * P * sizeof(double), hipMemcpyDeviceToDevice, st));
    MM_HIP(ctx, hipStreamSynchronize(st));
    if (timing) fprintf(stderr, "mm_ba_trf timing: total %.1f us, %d iterations\n",
                        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count(), iteration);
    rep->cost = cost;
    rep->optimality = g_norm;
    rep->nfev = nfev;
    rep->njev = njev;
    rep->status = termination;
    rep->iterations = iteration;
    rep->log_rows = n_log;
    rep->min_damping = min_damping;
    return MM_OK;
}

extern "C" int mm_ba_trf(mm_ctx *ctx, const mm_ba_problem *pb, double *cams, double *pts, const mm_trf_params *prm,
                         mm_trf_report *rep, mm_trf_row *log, int log_cap, void *ws, size_t ws_bytes) {
    return trf_run(ctx, pb, cams, pts, prm, rep, log, log_cap, ws, ws_bytes, nullptr);
}

extern "C" int mm_ba_trf_dist(mm_ctx *ctx, const mm_ba_problem *pb, double *cams, double *pts, const mm_trf_params *prm,
                              mm_trf_report *rep, mm_trf_row *log, int log_cap, void *ws, size_t ws_bytes, const mm_dist *dist) {
    if (!dist) return mm_fail(ctx, MM_ERR_ARG, "mm_ba_trf_dist: null communicator description");
    return trf_run(ctx, pb, cams, pts, prm, rep, log, log_cap, ws, ws_bytes, dist);
}


// ---- several independent problems in lock-step: mm_ba_trf_batched -------------------------------------------------------------
// The sliding-window adjustment (hook processor.py:395-408) solves dozens of small problems -- 50
// cameras, ~75 k points -- whose kernels keep a fraction of the chip busy for a few microseconds each: one such solve is
// bound by the latency of its ~28 launches per evaluation (0.29 ms), and running them on 8 streams gains 3x at most (more
// streams or hardware queues make it worse: measured).  Here ALL problems advance together: every kernel of the loop is
// launched once per round with blockIdx.y = problem (bodies and per-problem grids identical to the single-problem kernels:
// bit-identical iterates), the host reads one mailbox per problem and round and takes each problem's accept / reject /
// terminate decision exactly as mm_ba_trf does.  A round costs about what one evaluation of one problem costs.
// A reduced system that is not positive definite is retried with 100x the damping inside the batch, as mm_ba_trf does (on
// outlier-laden matches that is an everyday event).  A problem whose factorisation was abandoned, or that stays indefinite
// after six raises, is taken out of the batch and solved by mm_ba_trf itself afterwards, from its initial point: every
// result equals the one mm_ba_trf alone would have produced.
namespace {
struct BState {
    int phase;      // 0 body + trial, 1 trial only, 2 final gradient norm, 3 done, 4 solve alone afterwards,
                    // 5 reduced solve again with 100x the damping (+ trial)
    double Delta, cost, cost0, x_norm, step_norm, actual, g_norm, alpha, min_damping, reg;
    int nfev, njev, iteration, termination, attempt;
    long max_nfev;
    unsigned long long seq;
};
struct BHostBoard {
    double v[16];
    unsigned long long seq;
    unsigned long long pad[7];      // 192 bytes: mailboxes of different problems never share a cache line
};
size_t batch_tables_bytes(const mm_ba_problem *pb) { return 2 * mm_align_up((size_t)pb->F * 5 * sizeof(double), 256); }
}  // namespace

extern "C" size_t mm_ba_trf_batched_workspace_bytes(const mm_ba_problem *pb) {
    if (!pb || pb->F < 0 || pb->P < 0 || pb->O < 0) return 0;
    return carve_trf(pb, nullptr).total + batch_tables_bytes(pb);
}

extern "C" int mm_ba_trf_batched(mm_ctx *ctx, int n_prob, const mm_ba_problem *const *pbs, double *const *cams, double *const *pts,
                                 const mm_trf_params *prm, mm_trf_report *reports, void *const *ws, const size_t *ws_bytes,
                                 int32_t *solved_alone) {
    if (!ctx) return MM_ERR_ARG;
    if (n_prob < 0 || (n_prob > 0 && (!pbs || !cams || !pts || !prm || !reports || !ws || !ws_bytes)))
        return mm_fail(ctx, MM_ERR_ARG, "mm_ba_trf_batched: bad argument");
    if (n_prob == 0) return MM_OK;
    hipStream_t st = ctx->stream;
    for (int p = 0; p < n_prob; ++p) {
        const mm_ba_problem *pb = pbs[p];
        if (!pb || !cams[p] || !pts[p] || pb->F <= 0 || pb->P < 0 || pb->O < 0 || !pb->K)
            return mm_fail(ctx, MM_ERR_ARG, "mm_ba_trf_batched: bad problem %d", p);
        if (!ws[p] || ws_bytes[p] < mm_ba_trf_batched_workspace_bytes(pb) || ((uintptr_t)ws[p] & 255))
            return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_ba_trf_batched: workspace of problem %d too small or misaligned", p);
        if (solved_alone) solved_alone[p] = 0;
    }
    auto solve_alone = [&](int p) -> int {
        if (solved_alone) solved_alone[p] = 1;
        return trf_run(ctx, pbs[p], cams[p], pts[p], prm, &reports[p], nullptr, 0, ws[p], ws_bytes[p], nullptr);
    };
    // ---- the batch records; a problem the batched kernels cannot take sends the whole call down the one-by-one road ----
    std::vector<mm_batch_prob> tab((size_t)n_prob);
    std::vector<TrfWs> tws((size_t)n_prob);
    bool batchable = n_prob > 1;
    for (int p = 0; p < n_prob && batchable; ++p) {
        const mm_ba_problem *pb = pbs[p];
        const TrfWs t = carve_trf(pb, ws[p]);
        tws[p] = t;
        mm_batch_prob &b = tab[p];
        memset(&b, 0, sizeof(b));
        b.pb = *pb;
        b.nc = 6 * (int64_t)pb->F;
        b.n = b.nc + 3 * (int64_t)pb->P;
        b.x = t.x; b.x_new = t.x_new; b.g = t.g; b.si = t.si; b.gh = t.gh; b.ghs = t.ghs; b.gn = t.gn; b.q1 = t.q1; b.w = t.w;
        b.q2 = t.q2; b.s1 = t.s1; b.s2 = t.s2;
        b.B = t.B; b.Bd = t.Bd; b.C = t.C; b.Cd = t.Cd; b.Cinv = t.Cinv; b.u1 = t.u1; b.Jq2 = t.Jq2; b.dp = t.dp; b.v = t.v; b.S = t.S;
        b.r0 = t.r0; b.r1 = t.r1; b.r2 = t.r2; b.r3 = t.r3; b.d11 = t.d11; b.bs = t.bs; b.damp = t.damp; b.board = t.board;
        b.info = t.info;
        char *extra = (char *)ws[p] + t.total;
        b.ctab_x = extra;
        b.ctab_new = extra + mm_align_up((size_t)pb->F * 5 * sizeof(double), 256);
        b.res_partial = (double *)t.ws_res;
        b.md_counter = (unsigned *)t.ws_md;
        b.md_partial = (double *)((char *)t.ws_md + 256);
        b.jvp_partial = (double *)((char *)t.ws_jvp + 256);
        b.backsub_T = (double *)t.ws_back;
        b.half_bw = 6 * pb->cam_span + 5;
        b.g_vec = mm_batch_fused_grid(b.n);
        b.g_scale = (uint32_t)((b.n + 255) / 256);
        b.g_damp = (uint32_t)(((int64_t)pb->F * 36 + (int64_t)pb->P * 6 + 255) / 256);
        mm_batch_ba_setup(&b);
        if (pb->O <= 0 || pb->P <= 0 || mm_batch_schur_setup(ctx, &b, t.ws_schur, t.ws_schur_b) != MM_OK ||
            mm_batch_chol_setup(ctx, &b, t.ws_chol, t.ws_chol_b) != MM_OK)
            batchable = false;
    }
    ctx->batch_last = batchable ? n_prob : 0;
    if (!batchable) {
        if (getenv("MM_BATCH_DEBUG")) fprintf(stderr, "mm_ba_trf_batched: %d problems one by one (%s)\n", n_prob, n_prob > 1 ? ctx->err : "a single problem");
        for (int p = 0; p < n_prob; ++p) TRF_CALL(solve_alone(p));
        return MM_OK;
    }
    // ---- device / pinned staging (kept by the context) ----
    const size_t tab_b = mm_align_up((size_t)n_prob * sizeof(mm_batch_prob), 256), dyn_b = mm_align_up((size_t)n_prob * sizeof(mm_batch_dyn), 256);
    const size_t list_b = mm_align_up((size_t)n_prob * sizeof(int32_t), 256);
    constexpr int N_LISTS = 7;      // fused0 | final | body (new iteration) | trial | body (all) | retry | accept
    const size_t dev_need = tab_b + dyn_b + N_LISTS * list_b;
    const size_t host_need = dyn_b + N_LISTS * list_b + (size_t)n_prob * sizeof(BHostBoard) + (size_t)n_prob * 4 * sizeof(double);
    if (ctx->batch_dev_cap < dev_need) {
        if (ctx->batch_dev) (void)hipFree(ctx->batch_dev);
        ctx->batch_dev = nullptr;
        ctx->batch_dev_cap = 0;
        MM_HIP(ctx, hipMalloc(&ctx->batch_dev, dev_need));
        ctx->batch_dev_cap = dev_need;
    }
    if (ctx->batch_host_cap < host_need) {
        if (ctx->batch_host) (void)hipHostFree(ctx->batch_host);
        ctx->batch_host = nullptr;
        ctx->batch_host_cap = 0;
        MM_HIP(ctx, hipHostMalloc(&ctx->batch_host, host_need, hipHostMallocDefault));
        ctx->batch_host_cap = host_need;
    }
    char *dbase = (char *)ctx->batch_dev, *hbase = (char *)ctx->batch_host;
    mm_batch_prob *d_tab = (mm_batch_prob *)dbase;
    mm_batch_dyn *d_dyn = (mm_batch_dyn *)(dbase + tab_b);
    int32_t *d_list[N_LISTS];
    for (int q = 0; q < N_LISTS; ++q) d_list[q] = (int32_t *)(dbase + tab_b + dyn_b + q * list_b);
    mm_batch_dyn *h_dyn = (mm_batch_dyn *)hbase;
    int32_t *h_list[N_LISTS];
    for (int q = 0; q < N_LISTS; ++q) h_list[q] = (int32_t *)(hbase + dyn_b + q * list_b);
    BHostBoard *mail = (BHostBoard *)(hbase + dyn_b + N_LISTS * list_b);
    double *h_init = (double *)(mail + n_prob);      // per problem: cost2, |x si|^2 (3 doubles)
    memset(mail, 0, (size_t)n_prob * sizeof(BHostBoard));
    for (int p = 0; p < n_prob; ++p) tab[p].mailbox = &mail[p];
    MM_HIP(ctx, hipMemcpyAsync(d_tab, tab.data(), (size_t)n_prob * sizeof(mm_batch_prob), hipMemcpyHostToDevice, st));
    // ---- prologue, problem by problem with the single-problem calls (initial cost, normal equations, scale, Delta0) ----
    for (int p = 0; p < n_prob; ++p) {
        const mm_ba_problem *pb = pbs[p];
        const TrfWs &t = tws[p];
        const int64_t nc = tab[p].nc, n = tab[p].n;
        MM_HIP(ctx, hipMemsetAsync(t.ws_md, 0, t.ws_md_b, st));
        MM_HIP(ctx, hipMemsetAsync(t.ws_jvp, 0, t.ws_jvp_b, st));
        MM_HIP(ctx, hipMemcpyAsync(t.x, cams[p], (size_t)nc * sizeof(double), hipMemcpyDeviceToDevice, st));
        MM_HIP(ctx, hipMemcpyAsync(t.x + nc, pts[p], (size_t)3 * pb->P * sizeof(double), hipMemcpyDeviceToDevice, st));
        TRF_CALL(mm_ba_residual(ctx, pb, t.x, t.x + nc, nullptr, t.cost2, t.ws_res, t.ws_res_b));
        TRF_CALL(mm_ba_normal_eq(ctx, pb, t.x, t.x + nc, t.B, t.g, t.C, t.g + nc));
        TRF_CALL(mm_ba_scale_update(ctx, pb->F, pb->P, t.B, t.C, t.si, 1));
        hipLaunchKernelGGL(vec_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double *)t.x, (const double *)t.si, t.w, n);
        const double *pa[1] = {t.w}, *pbv[1] = {t.w};
        TRF_CALL(mm_multi_dot(ctx, 1, pa, pbv, n, nc, t.rowsx, t.ws_md, t.ws_md_b));
        MM_HIP(ctx, hipMemcpyAsync(h_init + 4 * p, t.cost2, sizeof(double), hipMemcpyDeviceToHost, st));
        MM_HIP(ctx, hipMemcpyAsync(h_init + 4 * p + 1, t.rowsx, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    MM_HIP(ctx, hipStreamSynchronize(st));
    std::vector<BState> S((size_t)n_prob);
    for (int p = 0; p < n_prob; ++p) {
        BState &s_ = S[p];
        s_ = BState{};
        s_.cost = s_.cost0 = 0.5 * h_init[4 * p];
        if (!std::isfinite(s_.cost)) {
            reports[p].status = -2;
            return mm_fail(ctx, MM_ERR_NUMERIC, "mm_ba_trf_batched: residuals of problem %d are not finite in the initial point", p);
        }
        s_.Delta = std::sqrt(h_init[4 * p + 3]);
        if (s_.Delta == 0) s_.Delta = 1.0;
        s_.max_nfev = prm->max_nfev > 0 ? prm->max_nfev : (long)tab[p].n * 100;
        s_.nfev = s_.njev = 1;
        s_.termination = -100;
        s_.step_norm = s_.actual = s_.g_norm = NAN;
        s_.phase = 0;
        s_.min_damping = prm->min_damping > 0 ? prm->min_damping : 1e-9;
    }
    // every problem's rotation coefficients at its starting point
    auto upload_lists = [&]() -> int {
        MM_HIP(ctx, hipMemcpyAsync(d_dyn, h_dyn, (size_t)n_prob * sizeof(mm_batch_dyn), hipMemcpyHostToDevice, st));
        MM_HIP(ctx, hipMemcpyAsync(d_list[0], h_list[0], (N_LISTS - 1) * list_b, hipMemcpyHostToDevice, st));
        return MM_OK;
    };
    auto max_of = [&](const int32_t *list, int cnt, uint32_t mm_batch_prob::*field) {
        uint32_t m = 1;
        for (int q = 0; q < cnt; ++q) m = std::max(m, tab[list[q]].*field);
        return m;
    };
    {
        for (int p = 0; p < n_prob; ++p) h_list[2][p] = p;
        TRF_CALL(upload_lists());
        TRF_CALL(mm_batch_cam_coef(ctx, d_tab, d_list[2], n_prob, max_of(h_list[2], n_prob, &mm_batch_prob::g_coef), 0));
    }
    auto wait_mail = [&](int p, unsigned long long seq) -> int {
        BHostBoard *hb = &mail[p];
        const auto t0 = std::chrono::steady_clock::now();
        unsigned long spins = 0;
        while (__atomic_load_n(&hb->seq, __ATOMIC_ACQUIRE) != seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
                MM_HIP(ctx, hipStreamSynchronize(st));
                if (__atomic_load_n(&hb->seq, __ATOMIC_ACQUIRE) != seq)
                    return mm_fail(ctx, MM_ERR_HIP, "mm_ba_trf_batched: the scalars of problem %d never arrived", p);
            }
        }
        return MM_OK;
    };
    // ---- rounds ----
    for (;;) {
        int n_f0 = 0, n_final = 0, n_body = 0, n_trial = 0, n_all = 0, n_retry = 0;
        for (int p = 0; p < n_prob; ++p) {
            BState &s_ = S[p];
            if (s_.phase == 3 || s_.phase == 4) continue;
            h_dyn[p].Delta = s_.Delta;
            h_dyn[p].seq = ++s_.seq;
            h_dyn[p].min_damping = s_.min_damping;
            h_dyn[p].reg = s_.reg;
            if (s_.phase == 0 || s_.phase == 2) h_list[0][n_f0++] = p;
            if (s_.phase == 2) h_list[1][n_final++] = p;
            if (s_.phase == 0) h_list[2][n_body++] = p;
            if (s_.phase != 2) h_list[3][n_trial++] = p;
            if (s_.phase == 0 || s_.phase == 5) h_list[4][n_all++] = p;
            if (s_.phase == 5) h_list[5][n_retry++] = p;
        }
        if (n_f0 + n_trial == 0) break;
        TRF_CALL(upload_lists());
        TRF_CALL(mm_batch_fused(ctx, 0, d_tab, d_list[0], n_f0, max_of(h_list[0], n_f0, &mm_batch_prob::g_vec)));
        TRF_CALL(mm_batch_publish_rows(ctx, d_tab, d_list[1], n_final, d_dyn));
        if (n_body) {
            TRF_CALL(mm_batch_jvp_dots(ctx, d_tab, d_list[2], n_body, max_of(h_list[2], n_body, &mm_batch_prob::g_jvp), 0));
            TRF_CALL(mm_batch_damping(ctx, d_tab, d_list[2], n_body, d_dyn));
        }
        TRF_CALL(mm_batch_set_reg(ctx, d_tab, d_list[5], n_retry, d_dyn));
        if (n_all) {
            const int32_t *L = d_list[4];
            const int32_t *H = h_list[4];
            const uint32_t gv = max_of(H, n_all, &mm_batch_prob::g_vec);
            TRF_CALL(mm_batch_damp(ctx, d_tab, L, n_all, max_of(H, n_all, &mm_batch_prob::g_damp)));
            TRF_CALL(mm_batch_schur(ctx, d_tab, L, n_all, max_of(H, n_all, &mm_batch_prob::g_zero), max_of(H, n_all, &mm_batch_prob::g_prep),
                                    max_of(H, n_all, &mm_batch_prob::g_pairs)));
            TRF_CALL(mm_batch_chol(ctx, d_tab, L, n_all, max_of(H, n_all, &mm_batch_prob::g_chol), max_of(H, n_all, &mm_batch_prob::g_bwd)));
            TRF_CALL(mm_batch_backsub(ctx, d_tab, L, n_all, max_of(H, n_all, &mm_batch_prob::g_obs), max_of(H, n_all, &mm_batch_prob::g_pts)));
            TRF_CALL(mm_batch_fused(ctx, 1, d_tab, L, n_all, gv));
            TRF_CALL(mm_batch_fused(ctx, 2, d_tab, L, n_all, gv));
            TRF_CALL(mm_batch_fused(ctx, 3, d_tab, L, n_all, gv));
            TRF_CALL(mm_batch_jvp_dots(ctx, d_tab, L, n_all, max_of(H, n_all, &mm_batch_prob::g_jvp), 1));
        }
        if (n_trial) {
            const int32_t *L = d_list[3];
            TRF_CALL(mm_batch_step2d(ctx, d_tab, L, n_trial, d_dyn));
            TRF_CALL(mm_batch_fused(ctx, 5, d_tab, L, n_trial, max_of(h_list[3], n_trial, &mm_batch_prob::g_vec)));
            TRF_CALL(mm_batch_cam_coef(ctx, d_tab, L, n_trial, max_of(h_list[3], n_trial, &mm_batch_prob::g_coef), 1));
            TRF_CALL(mm_batch_residual_publish(ctx, d_tab, L, n_trial, max_of(h_list[3], n_trial, &mm_batch_prob::g_res), d_dyn));
        }
        // ---- the host side of the round: every listed problem's decision, exactly as trf_run takes it ----
        int n_accept = 0;
        for (int q = 0; q < n_final; ++q) {
            const int p = h_list[1][q];
            BState &s_ = S[p];
            // (a final problem is also in no other mailbox-writing list this round: its publish carries this round's seq)
            TRF_CALL(wait_mail(p, s_.seq));
            s_.g_norm = ((volatile double *)mail[p].v)[5];
            s_.phase = 3;
        }
        for (int q = 0; q < n_trial; ++q) {
            const int p = h_list[3][q];
            BState &s_ = S[p];
            TRF_CALL(wait_mail(p, s_.seq));
            double host[16];
            for (int i = 0; i < 16; ++i) host[i] = ((volatile double *)mail[p].v)[i];
            if (s_.phase == 0 || s_.phase == 5) {      // first trial point of an iteration (or of a retry with raised damping)
                const int inf = (int)host[6];
                if (inf < 0) {      // the factorisation was abandoned: off the common road
                    s_.phase = 4;
                    continue;
                }
                if (inf > 0) {      // not positive definite at this damping: 100x, as mm_ba_trf (the trial point is dropped)
                    if (++s_.attempt >= 6) {
                        s_.phase = 4;      // (mm_ba_trf will report the failure)
                        continue;
                    }
                    if (host[13] <= s_.min_damping * (1.0 + 1e-12)) s_.min_damping *= 100.0;
                    s_.reg = host[13] * 100.0;
                    s_.phase = 5;
                    continue;
                }
                s_.attempt = 0;
                s_.g_norm = host[10];
                if (s_.g_norm < prm->gtol) {      // (checked before the step is used; the trial point is dropped)
                    s_.termination = 1;
                    s_.phase = 3;
                    continue;
                }
                s_.x_norm = std::sqrt(host[9]);
                s_.actual = -1.0;
            }
            const double predicted = host[2], step_h_norm = host[3], step_norm_dev = host[4];
            const double cost_new = 0.5 * host[14];
            ++s_.nfev;
            bool again = false;      // another trial step of the same iteration
            if (!std::isfinite(cost_new)) {
                s_.Delta = 0.25 * step_h_norm;
                again = s_.nfev < s_.max_nfev;
            } else {
                s_.actual = s_.cost - cost_new;
                double Delta_new = s_.Delta, ratio;
                update_tr_radius(Delta_new, s_.actual, predicted, step_h_norm, step_h_norm > 0.95 * s_.Delta, ratio);
                s_.step_norm = step_norm_dev;
                s_.termination = check_termination(s_.actual, s_.cost, s_.step_norm, s_.x_norm, ratio, prm->ftol, prm->xtol);
                if (s_.termination == -100) {
                    s_.alpha *= s_.Delta / Delta_new;
                    s_.Delta = Delta_new;
                    again = s_.actual <= 0 && s_.nfev < s_.max_nfev;
                }
            }
            if (again) {
                s_.phase = 1;
                continue;
            }
            if (s_.actual > 0) {
                s_.cost = cost_new;
                ++s_.njev;
                h_list[6][n_accept++] = p;
            } else {
                s_.step_norm = 0;
                s_.actual = 0;
            }
            ++s_.iteration;
            s_.phase = (s_.termination != -100 || s_.nfev == s_.max_nfev) ? 2 : 0;
        }
        if (n_accept) {
            MM_HIP(ctx, hipMemcpyAsync(d_list[6], h_list[6], (size_t)n_accept * sizeof(int32_t), hipMemcpyHostToDevice, st));
            const int32_t *L = d_list[6];
            TRF_CALL(mm_batch_accept(ctx, d_tab, L, n_accept, max_of(h_list[6], n_accept, &mm_batch_prob::g_vec)));
            TRF_CALL(mm_batch_normal_eq(ctx, d_tab, L, n_accept, max_of(h_list[6], n_accept, &mm_batch_prob::g_pblk),
                                        [&] { int m = 1; for (int q = 0; q < n_accept; ++q) m = std::max(m, tab[h_list[6][q]].pb.F); return (unsigned)m; }()));
            TRF_CALL(mm_batch_scale_update(ctx, d_tab, L, n_accept, max_of(h_list[6], n_accept, &mm_batch_prob::g_scale)));
        }
    }
    // ---- results ----
    for (int p = 0; p < n_prob; ++p) {
        BState &s_ = S[p];
        if (s_.phase != 3) continue;
        const int64_t nc = tab[p].nc;
        MM_HIP(ctx, hipMemcpyAsync(cams[p], tws[p].x, (size_t)nc * sizeof(double), hipMemcpyDeviceToDevice, st));
        MM_HIP(ctx, hipMemcpyAsync(pts[p], tws[p].x + nc, (size_t)3 * pbs[p]->P * sizeof(double), hipMemcpyDeviceToDevice, st));
        mm_trf_report &r = reports[p];
        r.cost0 = s_.cost0;
        r.cost = s_.cost;
        r.optimality = s_.g_norm;
        r.nfev = s_.nfev;
        r.njev = s_.njev;
        r.status = s_.termination == -100 ? 0 : s_.termination;
        r.iterations = s_.iteration;
        r.log_rows = 0;
        r.min_damping = s_.min_damping;
        r.chol_fallbacks = 0;
        r.collectives = 0;
    }
    MM_HIP(ctx, hipStreamSynchronize(st));
    for (int p = 0; p < n_prob; ++p)
        if (S[p].phase == 4) {
            --ctx->batch_last;
            TRF_CALL(solve_alone(p));
        }
    return MM_OK;
}
