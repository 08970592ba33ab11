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
    int prof = 0;  // 0 off, 1 every launch, 2 only launches of >= 64 workgroups (micro-launch chains stay untouched),
                   // 3 only the kernel named by mm_profile_select (the timed region of bench.py: one kernel's events, not all)
    char prof_only[64] = {0};
    std::vector<mm_prof_rec> recs;
    std::vector<hipEvent_t> pool;
    // second stream + fork/join events for overlapping the reduced-system build with its factorisation (lazily created)
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // one-time kernel attributes (dynamic LDS opt-in) are per device: remembered per context, not per process
    bool attr_chol_fused = false, attr_chol_bwd = false, attr_chol_batch = false;
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
    // mm_ba_trf returns the share itself at every read-back: no event needed there (recording one costs a ~6 us bubble
    // behind every solve: measured in the kernel trace)
    bool fused_no_event = false;
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
    // mm_ba_trf_batched: device tables (batch records, per-round lists) and pinned staging + mailboxes, grown on demand
    void *batch_dev = nullptr, *batch_host = nullptr;
    size_t batch_dev_cap = 0, batch_host_cap = 0;
    bool jvp_rows_deferred = false;      // mm_ba_jvp_dots leaves the final sum of its partials to the caller's next kernel ...
    const double *jvp_partial = nullptr;      // ... which finds them here
    unsigned jvp_n_wg = 0;
    const void *chol_init_done = nullptr;      // chol workspace whose fills the caller's own kernel has done (mm_chol_init_done)
    int chol_init_sides = 0;
    int batch_last = -1;      // problems the last mm_ba_trf_batched advanced in lock-step (0: all one by one), -1 none yet
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

// vec.hip: pass 0 of the loop (gh = g / si, ...) preceded by mm_ba_scale_update(first = 0) on the same elements, and pass 5
// (x_new = x + h0 s1 + h1 s2) which also leaves the cameras' rotation coefficients at x_new in `ctab` (mm_cam_table_adopt)
int mm_trf_fused0_scaled(mm_ctx *ctx, const double *g, double *si, const double *B, const double *C, double *gh, double *ghs, int64_t n,
                         int64_t split, double *out, void *ws, size_t ws_bytes);
int mm_trf_fused5_coef(mm_ctx *ctx, const double *x, const double *s1, const double *s2, double *x_new, const double *h01, int64_t n,
                       int64_t split, void *ctab, int F);
// vec.hip: mm_trf_step2d preceded by the final sum of the partials of the Jacobian product before it (what jvp_rows_kernel does,
// same tree; rows receives them as mm_ba_jvp_dots would have written them)
int mm_trf_rows_step2d(mm_ctx *ctx, const double *partial, unsigned n_wg, double *rows, const double *r0, const double *d11,
                       const double *r1, const double *r2, const double *r3, const double *reg, const int32_t *info, double Delta,
                       double *board);
// ba.hip: the context's coefficient table buffer for F cameras, to be filled by the caller's own kernel for the cameras at
// `cams` (then held like a table mm_cam_coef_table computed itself; needs mm_cam_table_hold(on))
int mm_cam_table_adopt(mm_ctx *ctx, const double *cams, int F, void **tab_out);
// vec.hip: mm_trf_damping + mm_ba_damp in one launch
int mm_ba_damp_damping(mm_ctx *ctx, int F, int P, const double *B, const double *C, const double *scale_inv, const double *gh2,
                       const double *d11, double Delta, double min_damping, double *damp_out, double *Bd, double *Cd);
// chol.hip: give back the workgroups a context has reserved for a single-launch factorisation (call after a host sync)
void mm_chol_release_budget(mm_ctx *ctx);
// chol.hip internals used by the overlapped Schur + solve entry point (schur.hip)
bool mm_chol_fused_eligible(int n, int half_bandwidth);
// what chol_init_kernel fills before a single-launch factorisation (chol_init.h has the device function)
struct mm_chol_init_args {
    int32_t *info, *flags;
    size_t nflags;
    unsigned long long *sentinel_buf;
    size_t nsent;
    unsigned long long *lpub;
    size_t nlpub;
    unsigned long long *Linv;
    size_t nblk;
};
// The arguments mm_chol_solve_sym(both triangles, one right-hand side) would hand to chol_init_kernel for this workspace;
// false: that solve would not take the single-launch path.  A caller whose own kernel then runs mm_chol_init_body says so
// with mm_chol_init_done right before the solve, which skips its launch (good for one solve on that workspace; if the
// solve ends up on the launch-per-column path after all -- no budget -- the fills were wasted, nothing more).
bool mm_chol_init_plan(mm_ctx *ctx, int n, int half_bandwidth, int32_t *info, void *ws, size_t ws_bytes, mm_chol_init_args *out, int *sides, int *bwb);
void mm_chol_init_done(mm_ctx *ctx, const void *ws, int sides);
// vec.hip / schur.hip: the damped blocks formed where they are consumed.  Bd = B + reg diag(si_c^2) is stored (the pair kernel
// reads it), the points' C + reg diag(si_p^2) only ever exists in the registers that invert it.  gh2 != nullptr: first
// attempt of an iteration, reg = max(trf_damping_value(gh2[0], d11[0], Delta), min_damping), recorded as {value, reg} in
// damp_out; else reg = reg[0].
struct mm_damp_spec {
    const double *B, *C, *si;
    const double *gh2, *d11;
    double Delta, min_damping;
    double *damp_out;
    const double *reg;
};
// schur.hip: mm_ba_damp(_damping) + mm_ba_schur_solve(no slabs) with the damping, the zero fill of the band of S and the
// factorisation's fills inside schur_prepare_kernel -- same arithmetic, three launches and a 72 MB fill less per solve at
// the bench shape.  S must be zero outside the band tiles on entry and stays so.  Falls back to the separate calls (which
// need Cd) when the problem has no pair list or the band is too wide for the single-launch factorisation.
int mm_ba_schur_solve_damped(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const mm_damp_spec *dmp,
                             double *Bd, double *Cd, const double *gc, const double *gp, double *S, double *v, double *Cinv,
                             int half_bandwidth, int32_t *info, void *ws_schur, size_t ws_schur_bytes, void *ws_chol,
                             size_t ws_chol_bytes);
int mm_chol_solve_gated(mm_ctx *ctx, double *A, int n, double *b, int nrhs, int half_bandwidth, int32_t *info, void *ws,
                        size_t ws_bytes, const int32_t *slab_ready, int cams_per_slab, int n_cams, int sym_mode);

// ---- lock-step solves of several independent problems (mm_ba_trf_batched, trf.hip) ------------------------------------------
// ONE record per problem in device memory: every buffer of its solve (the layout of mm_ba_trf's workspace) plus the launch
// geometry each kernel would be given if the problem were solved alone.  A batched launch is grid (max workgroups over the
// listed problems, number of listed problems): blockIdx.y picks the problem out of a list of indices, a workgroup beyond
// the problem's own grid leaves at once, and the body is the single-problem kernel's body with (blockIdx.x, gridDim.x)
// replaced by (blockIdx.x, the problem's own grid) -- same slices, same reduction trees, bit-identical results.
struct mm_batch_prob {
    mm_ba_problem pb;
    int64_t n, nc;      // 6 F + 3 P, 6 F
    double *x, *x_new, *g, *si, *gh, *ghs, *gn, *q1, *w, *q2, *s1, *s2;
    double *B, *Bd, *C, *Cd, *Cinv, *u1, *Jq2, *dp, *v, *S;
    double *r0, *r1, *r2, *r3, *d11, *bs, *damp, *board;
    int32_t *info;
    void *ctab_x, *ctab_new;      // rotation coefficients of the cameras in x / x_new (ba.hip's CamCoef rows)
    // reductions
    double *res_partial;          // residual: RES_BLOCKS partial sums
    unsigned *md_counter;         // fused vector passes: arrival counter + per-workgroup partials
    double *md_partial;
    double *jvp_partial;
    double *backsub_T;
    // reduced camera system (schur.hip) and its factorisation (chol.hip)
    double *schur_partial, *schur_camtab;
    int32_t *schur_seg_done, *schur_desc;
    double *chol_Linv, *chol_ytmp, *chol_contrib, *chol_contrib_bwd, *chol_lpub, *chol_spub;
    int32_t *chol_flags;
    int32_t chol_n, chol_nblk, chol_bwb, chol_a, chol_m, chol_b, chol_pad;      // (TwGeom of chol.hip)
    uint64_t chol_nflags, chol_nsent, chol_nlpub;
    int32_t half_bw, chol_sym_mirror;
    // per-kernel grids (what the single-problem launch would use)
    uint32_t g_vec, g_res, g_jvp, g_pblk, g_obs, g_pts, g_scale, g_damp, g_prep, g_pairs, g_chol, g_bwd, g_coef, g_zero;
    // host mailbox of this problem (pinned, device visible): 16 doubles + sequence number
    void *mailbox;
};
// what changes from round to round: one record per problem, copied to the device before each round
struct mm_batch_dyn {
    double Delta;
    unsigned long long seq;
    double min_damping;      // this problem's damping floor (raised x100 when the reduced system failed AT the floor)
    double reg;              // a retry of the reduced solve with 100x the damping: the value that replaces damp[1]
};

// batched launches of the loop's kernels (each in the translation unit of its single-problem kernel)
int mm_batch_fused(mm_ctx *ctx, int op, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g);
unsigned mm_batch_fused_grid(int64_t n);
int mm_batch_scale_update(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g);
int mm_batch_damp(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g);
int mm_batch_accept(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g);
int mm_batch_step2d(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, const mm_batch_dyn *dyn);
// ba.hip
void mm_batch_ba_setup(mm_batch_prob *bp);      // grids of the sweeps from bp->pb
int mm_batch_cam_coef(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g, int which /*0: x, 1: x_new*/);
int mm_batch_jvp_dots(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g, int second);
int mm_batch_damping(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, const mm_batch_dyn *dyn);
int mm_batch_set_reg(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, const mm_batch_dyn *dyn);
int mm_batch_backsub(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g_obs, unsigned max_g_pts);
int mm_batch_residual_publish(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g,
                              const mm_batch_dyn *dyn);
int mm_batch_normal_eq(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g_pblk, unsigned max_F);
int mm_batch_publish_rows(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, const mm_batch_dyn *dyn);
// schur.hip / chol.hip
int mm_batch_schur_setup(mm_ctx *ctx, mm_batch_prob *bp, void *ws_schur, size_t ws_schur_bytes);
int mm_batch_schur(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g_zero, unsigned max_g_prep,
                   unsigned max_g_pairs);
int mm_batch_chol_setup(mm_ctx *ctx, mm_batch_prob *bp, void *ws_chol, size_t ws_chol_bytes);
int mm_batch_chol(mm_ctx *ctx, const mm_batch_prob *tab, const int32_t *list, int n_list, unsigned max_g_chol, unsigned max_g_bwd);

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
        const bool p_ = (ctx)->prof == 1 || ((ctx)->prof == 2 && (size_t)g_.x * g_.y * g_.z >= 64) || \
                        ((ctx)->prof == 3 && strcmp(name, (ctx)->prof_only) == 0);                      \
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
