// Internal helpers shared by the translation units of libmeatmodeler_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/meatmodeler.h"

struct mm_prof_rec {
    const char *name;
    hipEvent_t a, b;
};

struct mm_ctx {
    int device;
    hipStream_t stream;
    char err[512];
    // optional per-launch HIP-event profiling (mm_profile_*): bench.py's live roofline measurement
    int prof = 0;  // 0 off, 1 every launch, 2 only launches of >= 64 workgroups (micro-launch chains stay untouched)
    std::vector<mm_prof_rec> recs;
    std::vector<hipEvent_t> pool;
    // second stream + fork/join events for overlapping the reduced-system build with its factorisation (lazily created)
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // one-time kernel attributes (dynamic LDS opt-in) are per device: remembered per context, not per process
    bool attr_chol_fused = false, attr_chol_bwd = false;
    // Single-launch banded factorisation (chol.hip): its workgroups wait for each other, so ALL of them have to be
    // resident -- one per CU (512 registers per lane).  cu_count = CUs of the device; fused_wgs = workgroups this context
    // has reserved out of the process-wide budget (mm_fused_budget) for its factorisation in flight, released at the
    // context's next host synchronisation; chol_avoid_fused: take the launch-per-column path (set by mm_ba_trf after an
    // abandoned factorisation); chol_last_path: 1 single launch, 0 per column, -1 none yet; debug_abandon: test hook.
    int cu_count = 0, fused_wgs = 0, chol_last_path = -1, debug_abandon = 0;
    bool chol_avoid_fused = false;
    // mm_ba_trf_dist: the choice between the two factorisation paths must be the same on every rank (they order their
    // sums differently and the replicated cameras must stay bit-identical).  With this set, a launch that finds no room
    // in the budget does NOT quietly take the per-column path: it reports info = -1 like an abandoned factorisation, the
    // flag travels with the trial cost, and all ranks switch together.
    bool chol_strict_budget = false;
    // recorded behind the last kernel of a single-launch solve: once it has completed the reservation can be returned
    // without a host synchronisation (by the owner's next reservation or by another context that finds the budget spent)
    hipEvent_t fused_ev = nullptr;
    bool fused_ev_pending = false;
    // link.hip: formulation the last mm_link_tracks_device took (1 parallel, 0 serial -- asked for, or the safety valve), -1 none
    int link_last_variant = -1;
    // rotation coefficients of the cameras the BA sweeps were last called with (ba.hip: mm_cam_coef_table)
    void *cam_tab = nullptr;
    int cam_tab_cap = 0, cam_tab_F = 0;
    const double *cam_tab_for = nullptr;
    bool cam_tab_hold = false;
    // pinned, device-visible host mailbox of mm_ba_trf (trf.hip): the trial-step scalars are written into it by a kernel
    // and the host spins on its sequence number instead of paying a copy + stream synchronisation per trial step
    void *host_board = nullptr;
    unsigned long long host_board_seq = 0;
};

// launches of the enclosed scope go to another stream of the context
struct mm_stream_swap {
    mm_ctx *c;
    hipStream_t saved;
    mm_stream_swap(mm_ctx *ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { ctx->stream = s; }
    ~mm_stream_swap() { c->stream = saved; }
};

// ba.hip: per-camera rotation coefficients shared by the sweeps (see there)
int mm_cam_coef_table(mm_ctx *ctx, const double *cams, int F, const void **tab_out);
void mm_cam_table_hold(mm_ctx *ctx, bool on);       // on: the table stays valid for the same camera pointer until ...
void mm_cam_table_invalidate(mm_ctx *ctx);          // ... the caller says the vector behind it changed

// ba.hip: the residual sweep whose final sum also delivers mm_ba_trf's scalar board to the pinned host mailbox
int mm_ba_residual_publish(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, void *ws,
                           size_t ws_bytes, double *board, int cost_slot, int count, void *host_board, unsigned long long seq);

// chol.hip: give back the workgroups a context has reserved for a single-launch factorisation (call after a host sync)
void mm_chol_release_budget(mm_ctx *ctx);
// chol.hip internals used by the overlapped Schur + solve entry point (schur.hip)
bool mm_chol_fused_eligible(int n, int half_bandwidth);
int mm_chol_solve_gated(mm_ctx *ctx, double *A, int n, double *b, int nrhs, int half_bandwidth, int32_t *info, void *ws,
                        size_t ws_bytes, const int32_t *slab_ready, int cams_per_slab, int n_cams, int sym_mode);

static inline hipEvent_t mm_prof_event(mm_ctx *c) {
    hipEvent_t e = nullptr;
    if (!c->pool.empty()) {
        e = c->pool.back();
        c->pool.pop_back();
    } else {
        (void)hipEventCreate(&e);
    }
    return e;
}

// Launch on the context's stream; when profiling is on, bracket the launch with two events on that stream.
#define MM_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                                          \
    do {                                                                                               \
        hipEvent_t ea_ = nullptr, eb_ = nullptr;                                                       \
        const dim3 g_ = (grid);                                                                        \
        const bool p_ = (ctx)->prof == 1 || ((ctx)->prof == 2 && (size_t)g_.x * g_.y * g_.z >= 64);    \
        if (p_) {                                                                                      \
            ea_ = mm_prof_event(ctx);                                                                  \
            eb_ = mm_prof_event(ctx);                                                                  \
            (void)hipEventRecord(ea_, (ctx)->stream);                                                  \
        }                                                                                              \
        hipLaunchKernelGGL(kernel, g_, block, shmem, (ctx)->stream, __VA_ARGS__);                      \
        if (p_) {                                                                                      \
            (void)hipEventRecord(eb_, (ctx)->stream);                                                  \
            (ctx)->recs.push_back(mm_prof_rec{name, ea_, eb_});                                        \
        }                                                                                              \
        MM_LAUNCH_CHECK(ctx, name);                                                                    \
    } while (0)

static inline int mm_fail(mm_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define MM_HIP(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return mm_fail(ctx, MM_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),    \
                           __FILE__, __LINE__);                                                        \
    } while (0)

#define MM_LAUNCH_CHECK(ctx, name)                                                                     \
    do {                                                                                               \
        hipError_t e_ = hipGetLastError();                                                             \
        if (e_ != hipSuccess)                                                                          \
            return mm_fail(ctx, MM_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_));   \
    } while (0)

static inline size_t mm_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
