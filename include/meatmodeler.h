/*
 * meatmodeler.h — C ABI of libmeatmodeler_hip.so (MI355X / gfx950 only).
 *
 * The drop-in boundary for the structure-from-motion hot path of skyepurchase/MeatModeler.
 * The reference has no FFI of its own (it is pure Python calling cv2 / scipy); each entry point below
 * names the reference call site it replaces (paths relative to the reference checkout) — the Python
 * façade in meatmodeler_amd/{processor,bundleAdjuster,track}.py binds them with ctypes
 * (see INTEGRATION.md for the stub a maintainer would add).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / C++ types.
 *  - Pointers marked [dev] are HIP device pointers on the context's device; [host] are host pointers.
 *  - Every call is asynchronous on the context's stream unless marked (sync); no call allocates or
 *    frees device memory: the caller passes workspaces sized by the *_workspace_bytes functions.
 *  - Return value: 0 = ok, negative = error (text via mm_last_error).  Nothing throws across the ABI.
 */
#ifndef MEATMODELER_H
#define MEATMODELER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_OK 0
#define MM_ERR_ARG (-1)
#define MM_ERR_HIP (-2)
#define MM_ERR_WORKSPACE (-3)
#define MM_ERR_NUMERIC (-4)

typedef struct mm_ctx mm_ctx;

/* ---- context ------------------------------------------------------------------------------- */
/* 3 since round 4: mm_trf_report is 64 bytes (chol_fallbacks, collectives), mm_ctx_control / mm_flatten_* /
 * mm_ba_trf_dist / mm_ba_trf_batched exist.  A caller built against an older header must refuse to run. */
int mm_abi_version(void);
/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL for the default stream. */
int mm_ctx_create(int device, void *hip_stream, mm_ctx **out);
void mm_ctx_destroy(mm_ctx *ctx);
const char *mm_last_error(mm_ctx *ctx);
int mm_ctx_sync(mm_ctx *ctx); /* (sync) hipStreamSynchronize on the context stream */
/* Knobs and queries of a context (tests, diagnostics).  Returns the queried value, MM_OK, or a negative error.
 *   MM_CTL_CHOL_FORCE_ABANDON: the next `value` single-launch banded factorisations on this context give up as if one of
 *     their workgroups had not become resident (info = -1) -- exercises the fall-back to the launch-per-column path;
 *   MM_CTL_CHOL_LAST_PATH: path of the last mm_chol_solve* on this context: 1 single launch, 0 per column, -1 none yet;
 *   MM_CTL_CHOL_RESERVED: workgroups this context currently holds of the per-process budget of co-resident
 *     factorisation workgroups (one compute unit each; given back at the context's next synchronisation, at every
 *     read-back inside mm_ba_trf, or -- once the event recorded behind the solve has completed -- at this context's next
 *     reservation or when another context finds the budget spent);
 *   MM_CTL_CU_COUNT: compute units of the device;
 *   MM_CTL_CHOL_AVOID_FUSED: value != 0: this context takes the launch-per-column factorisation from now on (what a
 *     caller that sequences the solver itself does after it has seen info = -1); 0: back to the default; < 0: query.
 *     Returns the setting as it was BEFORE the call (0 / 1), so a caller can restore it. */
#define MM_CTL_CHOL_FORCE_ABANDON 1
#define MM_CTL_CHOL_LAST_PATH 2
#define MM_CTL_CHOL_RESERVED 3
#define MM_CTL_CU_COUNT 4
#define MM_CTL_CHOL_AVOID_FUSED 5
/*   MM_CTL_LINK_LAST_VARIANT: formulation the last mm_link_tracks_device on this context took: 1 parallel (pointer
 *     doubling), 0 serial (asked for with MM_LINK_VARIANT=serial, or the safety valve: the parallel formulation's
 *     fixed-point pass over key points with coincident coordinates exceeded its work budget), -1 none yet. */
#define MM_CTL_LINK_LAST_VARIANT 6
/*   MM_CTL_BATCH_LAST: problems the last mm_ba_trf_batched on this context advanced in lock-step (0: all were solved one by
 *     one), -1 none yet. */
#define MM_CTL_BATCH_LAST 7
long long mm_ctx_control(mm_ctx *ctx, int what, long long value);
/* HIP-event timing on the context stream (bench.py's roofline leg). */
int mm_timer_create(mm_ctx *ctx, void **timer_out);
int mm_timer_start(mm_ctx *ctx, void *timer);
int mm_timer_stop(mm_ctx *ctx, void *timer);
int mm_timer_elapsed_ms(mm_ctx *ctx, void *timer, float *ms_out); /* (sync) */
void mm_timer_destroy(mm_ctx *ctx, void *timer);
/* Per-launch profiling: on = 1 brackets every kernel launched through this context with two HIP events on the
 * context's stream (~2.5 us of device time and ~5 us of host time per launch); on = 2 only launches of >= 64
 * workgroups, leaving the micro-launch chains of the Cholesky untouched; on = 3 only launches of the kernel named by
 * mm_profile_select (the name mm_profile_report prints; what bench.py's timed steps use: every bracketed launch costs the
 * stream a bubble, ~10 % of a BA iteration with on = 2); on = 0 stops collecting.  mm_profile_report (sync) writes one line
 * per kernel name, "name launches total_ms\n", and returns the number of lines.  Enabling clears earlier records. */
int mm_profile_enable(mm_ctx *ctx, int on);
int mm_profile_select(mm_ctx *ctx, const char *name);
int mm_profile_report(mm_ctx *ctx, char *buf, size_t buf_len);

/* ---- a-2: brute-force Hamming 2-NN + ratio test ----------------------------------------------
 * Replaces cv2.FlannBasedMatcher(...).knnMatch(prev_desc, new_desc, k=2) and the Lowe filter,
 * processor.py:132-137 (exact form of the approximate FLANN search; ties -> lowest train index).
 *
 * Batched form: pair p matches query set  q + p*q_set_stride  (nq[p] rows, or nq_cap if nq == NULL)
 * against train set t + p*t_set_stride.  Descriptors are 32 bytes, rows contiguous, 16-byte aligned.
 *   idx  [n_pairs, nq_cap, 2] int32 : train index of the nearest / second nearest (-1 if absent)
 *   dist [n_pairs, nq_cap, 2] int32 : their Hamming distances (-1 if absent)
 * Rows >= nq[p] are left untouched.
 * The distances are computed on the matrix cores (descriptors expanded to +1 / -1 FP4 values: dot product = 256 - 2 dist,
 * exact); the workspace holds the expanded train sets, 128 bytes per train descriptor (mm_bf_workspace_bytes; 16-byte
 * aligned).  Train sets of fewer than 64 or of 65536 and more descriptors take the xor / popcount kernel instead;
 * MM_BF_VARIANT=114 (xor / popcount), 200 (int8 MFMA), 300 (FP4 MFMA, default) in the environment selects one -- all
 * return identical results.
 */
size_t mm_bf_workspace_bytes(int n_pairs, int nq_cap, int nt_cap);
int mm_bf_knn2_batched(mm_ctx *ctx, const uint8_t *q /*dev*/, const int32_t *nq /*dev|NULL*/, int nq_cap,
                       size_t q_set_stride, const uint8_t *t /*dev*/, const int32_t *nt /*dev|NULL*/, int nt_cap,
                       size_t t_set_stride, int n_pairs, int32_t *idx /*dev*/, int32_t *dist /*dev*/,
                       void *ws /*dev*/, size_t ws_bytes);
int mm_bf_knn2_hamming(mm_ctx *ctx, const uint8_t *q /*dev*/, int nq, const uint8_t *t /*dev*/, int nt,
                       int32_t *idx /*dev [nq,2]*/, int32_t *dist /*dev [nq,2]*/, void *ws, size_t ws_bytes);
/* Lowe ratio test, order preserving: keeps query i iff it has two neighbours and
 * (double)d0 < threshold * (double)d1  (processor.py:136-137).  pairs[p, j] = (queryIdx, trainIdx) of the
 * j-th kept match in query order; m_out[p] = number kept. */
int mm_ratio_filter_batched(mm_ctx *ctx, const int32_t *idx /*dev*/, const int32_t *dist /*dev*/,
                            const int32_t *nq /*dev|NULL*/, int nq_cap, int n_pairs, double threshold,
                            int32_t *pairs /*dev [n_pairs,nq_cap,2]*/, int32_t *m_out /*dev [n_pairs]*/);

/* ---- a-1: ORB detect + describe ----------------------------------------------------------------
 * Replaces orb.detectAndCompute(img, None), processor.py:129,328 with cv2.ORB_create(nfeatures=...) defaults
 * (processor.py:308): 8 levels x1.2, FAST-9/16 t=20 + 3x3 NMS, border 31, 2n by FAST score then n by Harris,
 * intensity-centroid orientation, 7x7 sigma=2 blur, 256-bit steered BRIEF.  Integer-exact definition in DESIGN.md.
 */
typedef struct mm_orb_params {
    int32_t nfeatures;      /* processor.py:308 uses 20000; BASELINE configs 2000 / 4000 / 8000 */
    int32_t nlevels;        /* 8  */
    int32_t edge_threshold; /* 31 */
    int32_t fast_threshold; /* 20 */
    float scale_factor;     /* 1.2f */
    int32_t reserved;
} mm_orb_params;
size_t mm_orb_workspace_bytes(int batch, int height, int width, const mm_orb_params *prm);
/* imgs [batch, height, stride] u8 grey.  Outputs are per frame with capacity cap = nfeatures:
 *   kp_xy   [batch, cap, 2] f32 : level-0 coordinates (cv2.KeyPoint.pt)
 *   kp_meta [batch, cap, 4] i32 : level, x_level, y_level, harris25 (low 32 bits; debugging / parity)
 *   kp_resp [batch, cap]    f32 : Harris response (cv2.KeyPoint.response)
 *   kp_mom  [batch, cap, 2] i32 : intensity-centroid moments m10, m01 (angle = atan2(m01, m10))
 *   desc    [batch, cap, 32] u8
 *   n_out   [batch] i32
 * pattern [256*4] int8 (x0,y0,x1,y1 per bit) device pointer — the rBRIEF sampling pattern. */
int mm_orb_detect_compute(mm_ctx *ctx, const uint8_t *imgs /*dev*/, int batch, int height, int width, int stride,
                          const mm_orb_params *prm, const int8_t *pattern /*dev*/, void *ws /*dev*/, size_t ws_bytes,
                          float *kp_xy, int32_t *kp_meta, float *kp_resp, int32_t *kp_mom, uint8_t *desc,
                          int32_t *n_out);
/* Level geometry shared with the host side (and the oracle checks it): widths/heights of the nlevels levels. */
int mm_orb_level_sizes(int height, int width, const mm_orb_params *prm, int32_t *lvl_w /*host[nlevels]*/,
                       int32_t *lvl_h /*host*/, int32_t *lvl_n /*host: features per level*/,
                       float *lvl_scale /*host*/);

/* ---- a-3 / a-5: track linking and flattening (host logic, exact coordinate-equality semantics) -----
 * Replaces processor.pointTracking (processor.py:190-243) run over a whole clip plus
 * `popped_tracks += tracks` (processor.py:418) and managePoints (processor.py:264-291).
 * All pointers are HOST pointers.  kp_xy [n_frames, cap, 2] f32, matches [n_frames-1, mcap, 2] (queryIdx, trainIdx).
 * Output: CSR over tracks in the reference's final order; obs_frame / obs_kp give each observation.
 * Returns the number of tracks (>= 0) or a negative error; *n_obs_out receives the observation count.
 * If the output capacity is too small returns MM_ERR_WORKSPACE with the required sizes in *n_obs_out / return. */
int64_t mm_link_tracks_clip(int n_frames, int cap, const int32_t *kp_count, const float *kp_xy, int mcap,
                            const int32_t *match_count, const int32_t *matches, int64_t max_tracks,
                            int64_t max_obs, int64_t *track_ptr /*[max_tracks+1]*/, int32_t *obs_frame,
                            int32_t *obs_kp, int64_t *n_obs_out);
/* The same linking ON THE DEVICE (one resident workgroup walks the clip; scatter-min / scatter-max / prefix sums per
 * keyframe pair): all pointers are DEVICE pointers, nothing is copied to the host, no synchronisation.
 * kp_xy [n_frames, cap, 2] f32, matches [n_frames-1, cap, 2] i32 (queryIdx, trainIdx), cap <= 8192.
 * Outputs sized for the worst case: track_ptr [(n_frames-1)*cap + 1] i32, obs_frame / obs_kp [2*(n_frames-1)*cap] i32;
 * counts [3] i64 = (n_tracks, n_obs, 1 if a malformed match was skipped).  Same result as mm_link_tracks_clip. */
size_t mm_link_workspace_bytes(int n_frames, int cap);
int mm_link_tracks_device(mm_ctx *ctx, int n_frames, int cap, const int32_t *kp_count, const float *kp_xy,
                          const int32_t *match_count, const int32_t *matches, void *ws, size_t ws_bytes,
                          int32_t *track_ptr, int32_t *obs_frame, int32_t *obs_kp, int64_t *counts /*dev[3]*/);
/* managePoints ON THE DEVICE (processor.py:264-291): the flat observation arrays of a selection of tracks, point-major,
 * insertion order inside a track -- what adjustPoints takes (bundleAdjuster.py:160-166).  All pointers are DEVICE pointers.
 *   selection: sel == NULL: the n_sel tracks t_lo .. t_lo + n_sel - 1 (everything, or one rank's shard);
 *              sel [n_sel] i32: these tracks in this order (a sliding window), with out_ptr [n_sel + 1] i64 = the exclusive
 *              scan of their lengths from mm_flatten_offsets (out_ptr[n_sel] = number of observations: read it back to size
 *              the outputs);
 *   coords [n_obs, 2] f64 = kp_xy[obs_frame, obs_kp] (kp_xy [F, cap, 2] f32); frame_indices [n_obs] i32 = obs_frame -
 *   frame_offset; point_indices [n_obs] i32 = position of the observation's track in the selection. */
int mm_flatten_offsets(mm_ctx *ctx, const int32_t *track_ptr, const int32_t *sel, int64_t n_sel, int64_t *out_ptr);
int mm_flatten_tracks(mm_ctx *ctx, const int32_t *track_ptr, const int32_t *obs_frame, const int32_t *obs_kp, const float *kp_xy,
                      int cap, const int32_t *sel /*|NULL*/, int t_lo, int64_t n_sel, const int64_t *out_ptr /*|NULL*/, int64_t n_obs,
                      int frame_offset, double *coords, int32_t *frame_indices, int32_t *point_indices);
/* Host: co-observation pairs (o, o2) of one point with camera(o2) <= camera(o), grouped by block segment
 * camera(o) * (span + 1) + camera(o) - camera(o2) in a fixed canonical order.  seg_ptr [F*(span+1)+1].  Call with
 * pair_o == NULL to get the pair count.  Returns the count or a negative error (MM_ERR_ARG if span is too small). */
int64_t mm_ba_build_pairs(int F, int P, int64_t O, const int32_t *fi, const int32_t *pi, const int32_t *pt_ptr,
                          const int32_t *pt_obs, const int32_t *cam_ptr, const int32_t *cam_obs, int span,
                          int64_t *seg_ptr, int32_t *pair_o, int32_t *pair_o2, int64_t max_pairs);
/* Host index build for BA: CSR of observations by point and by camera (stable). */
int mm_ba_build_index(int F, int P, int64_t O, const int32_t *fi, const int32_t *pi, int32_t *pt_ptr /*[P+1]*/,
                      int32_t *pt_obs /*[O]*/, int32_t *cam_ptr /*[F+1]*/, int32_t *cam_obs /*[O]*/);

/* ---- a-4: two-view DLT triangulation ------------------------------------------------------------
 * Replaces the per-track cv2.triangulatePoints + dehomogenise loop, processor.py:254-261.
 * proj [F,3,4] f64; track i uses views f0[i], f1[i] with pixels x0[i], x1[i]; X [n,3]. */
int mm_triangulate_dlt(mm_ctx *ctx, const double *proj /*dev*/, const int32_t *f0 /*dev*/, const int32_t *f1 /*dev*/,
                       const double *x0 /*dev [n,2]*/, const double *x1 /*dev [n,2]*/, int64_t n,
                       double *X /*dev [n,3]*/);

/* ---- a-7..a-9: bundle adjustment sweeps -----------------------------------------------------------
 * Cost model of bundleAdjuster.py:7-52,81-102 (Rodrigues rotate, translate, full 3x3 K, divide, minus obs),
 * f64 throughout.  The LM/trust-region driver (the reference's scipy least_squares TRF loop,
 * bundleAdjuster.py:180-192) lives in meatmodeler_amd/bundleAdjuster.py and calls these sweeps. */
typedef struct mm_ba_problem {
    int32_t F, P;
    int64_t O;
    const double *K;         /* dev [9] row-major 3x3 */
    const int32_t *fi, *pi;  /* dev [O] camera / point index of each observation */
    const double *obs;       /* dev [O,2] */
    const int32_t *pt_ptr, *pt_obs;   /* dev CSR by point  (mm_ba_build_index) */
    const int32_t *cam_ptr, *cam_obs; /* dev CSR by camera */
    /* optional co-observation pair list (mm_ba_build_pairs) for the banded, bitwise reproducible Schur kernel;
     * n_seg == 0 / NULL pointers select the general kernel.  Segments are cut into chunks of at most 256 pairs: one
     * wave sums a chunk, a second pass adds the chunks of a segment in order. */
    int32_t cam_span;                 /* max over points of (largest - smallest observing camera index) */
    int32_t reserved;
    int64_t n_seg;                    /* number of NON-EMPTY block segments */
    const int32_t *seg_ids;           /* dev [n_seg]   segment id = camera * (cam_span + 1) + (camera - camera2) */
    const int32_t *seg_chunk_ptr;     /* dev [n_seg+1] first chunk of each segment */
    int64_t n_chunks;
    const int32_t *chunk_seg;         /* dev [n_chunks] index into seg_ids */
    const int32_t *chunk_begin, *chunk_end; /* dev [n_chunks] pair range */
    const int32_t *pair_o, *pair_o2;  /* dev [n_pairs] */
    const int32_t *pair_p;            /* dev [n_pairs] point index of each pair (= pi[pair_o]); NULL: looked up */
} mm_ba_problem;

/* res [O,2] (may be NULL) ; cost2 [1] receives sum of squared residuals (caller halves it). */
int mm_ba_residual(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams /*dev [F,6]*/,
                   const double *pts /*dev [P,3]*/, double *res /*dev|NULL*/, double *cost2 /*dev [1]*/,
                   void *ws, size_t ws_bytes);
/* Analytic Jacobian blocks per observation: Jc [O,2,6], Jp [O,2,3] (parity / debugging surface). */
int mm_ba_jacobian(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, double *Jc,
                   double *Jp);
/* Block normal equations: B [F,6,6] = sum Jc^T Jc, gc [F,6] = sum Jc^T r, C [P,6] = upper triangle of sum Jp^T Jp
 * (xx,xy,xz,yy,yz,zz), gp [P,3] = sum Jp^T r.  Deterministic (segmented, atomic-free).  gc/B may be NULL when
 * cameras are fixed; C/gp may be NULL when points are fixed (pose-only, bundleAdjuster.py:206-243). */
int mm_ba_normal_eq(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, double *B,
                    double *gc, double *C, double *gp);
/* out [O,2] = Jc * wc[fi] + Jp * wp[pi]  (either w may be NULL = zero). */
int mm_ba_jvp(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const double *wc,
              const double *wp, double *out);
/* mm_ba_jvp with its inner products fused in: rows [2,3] dev receives {0, s, s} for s = <out, other> (other == NULL:
 * <out, out>) and for s = <out, out> (the layout of mm_multi_dot's rows for residual-space vectors).  Deterministic. */
size_t mm_ba_jvp_dots_workspace_bytes(const mm_ba_problem *pb);
int mm_ba_jvp_dots(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const double *wc,
                   const double *wp, double *out, const double *other /*dev [O,2] | NULL*/, double *rows /*dev [2,3]*/,
                   void *ws, size_t ws_bytes);
/* Reduced camera system.  Cd [P,6] = damped point blocks (C + reg*diag), Bd [F,6,6] damped camera blocks.
 * S [6F,6F] (row-major) = blockdiag(Bd) - sum_p E_p Cd_p^-1 E_p^T ;  v [6F] = gc - E Cd^-1 gp.
 * Cinv [P,6] receives Cd^-1 (upper triangle).  S and v are overwritten.
 * If the problem carries a co-observation pair list (mm_ba_build_pairs) the LOWER block band |i - j| <= cam_span of S
 * is computed by an atomic-free, bitwise reproducible kernel and the rest of S is zero; otherwise a general kernel
 * fills all of S (LDS f64 atomics, last bits vary from run to run). */
int mm_ba_schur(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const double *Bd,
                const double *Cd, const double *gc, const double *gp, double *S, double *v, double *Cinv, void *ws,
                size_t ws_bytes);
size_t mm_ba_schur_workspace_bytes(const mm_ba_problem *pb); /* 42 doubles per chunk (0 without a pair list) */
/* Device-side construction of the pair list (needs fi, pi and the CSR by point in *pb):
 *   count: cnt[o] = number of observations o2 of o's point with camera(o2) <= camera(o); span_out[0] = widest camera
 *          distance inside a point (= cam_span);
 *   emit : after an exclusive scan of cnt (offsets), writes key = camera(o)*(span+1) + camera(o) - camera(o2) and the
 *          pair (o, o2) at offsets[o]...; a STABLE sort by key then yields the segments in a fixed order. */
int mm_ba_pairs_count(mm_ctx *ctx, const mm_ba_problem *pb, int32_t *cnt /*dev [O]*/, int32_t *span_out /*dev [1]*/);
int mm_ba_pairs_emit(mm_ctx *ctx, const mm_ba_problem *pb, const int64_t *offsets /*dev [O]*/, int span,
                     int32_t *key /*dev [n]*/, int32_t *pair_o /*dev [n]*/, int32_t *pair_o2 /*dev [n]*/);
/* k <= 8 inner products <a_q, b_q> over vectors of length n in one launch (the trust-region driver's reductions, SciPy
 * trf.py via bundleAdjuster.py:180-192).  a, b: HOST arrays of k device pointers.  out [k,3] dev = {sum over i < split,
 * sum over i >= split, total}; deterministic.  The workspace must be zero-filled once before its first use. */
size_t mm_multi_dot_workspace_bytes(void);
int mm_multi_dot(mm_ctx *ctx, int k, const double *const *a, const double *const *b, int64_t n, int64_t split,
                 double *out, void *ws, size_t ws_bytes);
/* Fused element-wise passes of the 2-D subspace trust-region step (SciPy trf.py:478-494 via bundleAdjuster.py:180-192),
 * each accumulating the inner products the next pass needs (deterministic, reported like mm_multi_dot as rows of
 * {i < split, i >= split, total}; one more row holds maxima).  in / outv / scalars: HOST arrays of device pointers.
 *   op 0: in {g, si}                 outv {gh = g/si, ghs = gh/si}            out rows: gh.gh | max |g|
 *   op 1: in {v[split], dp[n-split], si, gh}  scalars {gh2}  outv {gn = [v;dp]*si, q1 = gh/sqrt(gh2)}   rows: q1.gn, gn.gn | -
 *   op 2: in {gn, q1}  scalars {sc}   outv {w = gn - sc q1}                    rows: w.w | -
 *   op 3: in {w, q1, si, gh, x}  scalars {wn2}  outv {q2 = w/sqrt(wn2), s1 = q1/si, s2 = q2/si}
 *                                                                              rows: s1.s1, s1.s2, s2.s2, q2.gh, x.x | -
 *   op 4: in {x, s1, s2}  h0, h1      outv {x + h0 s1 + h1 s2}                 (no result rows)
 *   op 5: in {x, s1, s2}  scalars {p[2]}  outv {x + p[0] s1 + p[1] s2}  with p in device memory (mm_trf_step2d);
 *         p[1] == 0 skips s2 (one-dimensional subspace)                        (no result rows)
 * out [(K+1), 3] dev; workspace as for mm_multi_dot (zero-filled once). */
int mm_trf_fused(mm_ctx *ctx, int op, const double *const *in, double *const *outv, const double *const *scalars,
                 double h0, double h1, int64_t n, int64_t split, double *out, void *ws, size_t ws_bytes);
/* Regulariser of the trust-region sub-problem on the device (SciPy trf.py:473-477, reached through
 * bundleAdjuster.py:180-192): gh2 = |g_h|^2, d11 = |J_h g_h|^2 (device scalars), Delta the radius.
 * out [2] dev = {reg, max(reg, min_damping)}. */
int mm_trf_damping(mm_ctx *ctx, const double *gh2, const double *d11, double Delta, double min_damping, double *out);
/* dp [P,3] = Cinv (gp - E^T dc).  With a workspace (mm_ba_backsub_workspace_bytes: 24 bytes per observation) the work is
 * spread over the observations and added per point in a second pass -- a launch with one thread per point waits for the
 * longest tracks; ws == NULL keeps that one-launch form.  Both are deterministic. */
size_t mm_ba_backsub_workspace_bytes(const mm_ba_problem *pb);
int mm_ba_backsub(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const double *Cinv,
                  const double *gp, const double *dc /*dev [F,6]*/, double *dp, void *ws /*dev|NULL*/, size_t ws_bytes);
/* mm_ba_schur followed by mm_chol_solve(S, 6F, v, 1, half_bandwidth), overlapped: S is built in n_slabs ascending camera
 * slabs (slab s = cameras [s, s+1) * cams_per_slab; slab_seg_ptr / slab_chunk_ptr: HOST arrays [n_slabs+1] with the first
 * segment / chunk of each slab) on the context's stream while the single-launch banded factorisation runs on a second
 * stream and consumes block rows as their slabs complete.  On return (stream-ordered) v holds the solution and info the
 * factorisation status.  Without a pair list / slabs, or for wide bands, the two steps simply run one after the other. */
int mm_ba_schur_solve(mm_ctx *ctx, const mm_ba_problem *pb, const double *cams, const double *pts, const double *Bd,
                      const double *Cd, const double *gc, const double *gp, double *S, double *v, double *Cinv,
                      int half_bandwidth, int32_t *info, void *ws_schur, size_t ws_schur_bytes, void *ws_chol,
                      size_t ws_chol_bytes, int n_slabs, int cams_per_slab, const int64_t *slab_seg_ptr,
                      const int64_t *slab_chunk_ptr);
/* Block glue of the trust-region driver (one launch each instead of a handful of tensor-library kernels):
 * scale_update: scale_inv [6F+3P] = sqrt of the diagonal of J^T J taken from B [F,6,6] and C [P,6] (packed upper
 *   triangle); first != 0: zeros become 1 (SciPy compute_jac_scale), else the running maximum with the stored value;
 * damp: Bd = B + reg diag(scale_inv_c^2), Cd = C + reg diag(scale_inv_p^2), reg [1] in device memory. */
int mm_ba_scale_update(mm_ctx *ctx, int F, int P, const double *B, const double *C, double *scale_inv /*in/out*/, int first);
int mm_ba_damp(mm_ctx *ctx, int F, int P, const double *B, const double *C, const double *scale_inv, const double *reg /*dev [1]*/,
               double *Bd, double *Cd);
/* The 2-D trust-region subproblem of a TRF iteration solved on the device (SciPy trf.py:481-494, common.py:171-219) from
 * the results of the fused passes, all still in device memory: r0 [2,3] (op 0), d11 [1,3] = <J g_hs, J g_hs>, r1 [3,3]
 * (op 1), r2 [2,3] (op 2), r3 [6,3] (op 3), bs [2,3] = {<u1, J s2>, <J s2, J s2>} (the "total" column of each is used),
 * reg [1] the damping in use, info [1] the factorisation status.  board [14] receives p0, p1 (step in the orthonormal
 * basis; feed them to mm_trf_fused op 5), predicted reduction, |p|, unscaled step norm, degenerate flag, info, and the
 * scalars the host-side bookkeeping needs (wn2, gn2, |x|^2, |g|_inf, |g_h|^2, d11, reg): one read-back per trial step. */
int mm_trf_step2d(mm_ctx *ctx, const double *r0, const double *d11, const double *r1, const double *r2, const double *r3,
                  const double *bs, const double *reg, const int32_t *info, double Delta, double *board /*dev [14]*/);
/* The whole trust-region solve of adjustPoints in ONE call (single GPU; replaces the loop scipy.optimize.least_squares
 * runs for bundleAdjuster.py:180-192: method='trf', x_scale='jac', linear loss, exact Schur-complement step instead of
 * LSMR).  Sequences exactly the calls above the way meatmodeler_amd/bundleAdjuster.py does -- bit-identical iterates --
 * with the host side of an iteration in C++: one 16-double read-back per trial step, no allocation.
 * cams [F,6] / pts [P,3] dev: initial values in, solution out.  log (HOST, may be NULL with log_cap 0) receives one row
 * per line of SciPy's verbose=2 table; report->log_rows counts the rows produced (may exceed log_cap).
 * status: SciPy's (0 max_nfev, 1 gtol, 2 ftol, 3 xtol, 4 both).  Returns MM_ERR_NUMERIC if the residuals are not finite
 * at the start or the reduced system stays indefinite after six 100x increases of the damping. */
typedef struct mm_trf_params {
    double ftol, xtol, gtol;
    double min_damping;   /* floor of the damping relative to the unit diagonal of the scaled system (<= 0: 1e-9) */
    int64_t max_nfev;     /* <= 0: 100 * (6F + 3P) */
} mm_trf_params;
typedef struct mm_trf_row {
    int32_t iteration, nfev;
    double cost, reduction, step_norm, optimality;   /* reduction / step_norm are NaN on the first row */
} mm_trf_row;
typedef struct mm_trf_report {
    double cost0, cost, optimality, min_damping /* as raised during the solve */;
    int32_t nfev, njev, status, iterations, log_rows;
    int32_t chol_fallbacks;   /* times the solve switched to the launch-per-column factorisation (info = -1 seen): 0 or 1 */
    int32_t collectives;      /* mm_ba_trf_dist: all-reduce calls made (7 per trust-region iteration + 3 at the start) */
    int32_t reserved;
} mm_trf_report;
size_t mm_ba_trf_workspace_bytes(const mm_ba_problem *pb);
int mm_ba_trf(mm_ctx *ctx, const mm_ba_problem *pb, double *cams /*dev, in/out*/, double *pts /*dev, in/out*/,
              const mm_trf_params *prm, mm_trf_report *report /*host*/, mm_trf_row *log /*host|NULL*/, int log_cap,
              void *ws /*dev, 256-byte aligned*/, size_t ws_bytes);
/* The same loop SHARDED over the GPUs of a node (one process per GPU): `pb` holds this rank's points
 * with all their observations (point indices local), the cameras are replicated, `pts` is the rank's shard.  Every sum
 * over observations / points is completed by the caller's all-reduce -- the library does not link a communication
 * library: `allreduce(user, buf, count)` must sum `count` doubles at the DEVICE pointer `buf` (inside the workspace) over
 * the ranks, in place, ordered after the work already on the context's stream and before what follows (RCCL:
 * ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, stream); torch.distributed: all_reduce on a tensor view of the
 * workspace), and return 0.  Per trust-region iteration: {B, g_c} | {|g_h|^2, |g|_inf, |J d g_h|^2} | {band of S, v} |
 * {<q1, gn_h>, |gn_h|^2} | {|w|^2} | {the five step inner products, <u1, J s2>, |J s2|^2} | {trial cost, "a rank's
 * factorisation was abandoned"} -- seven collectives, six of them one 64-double vector (the chain of dependent scalars of
 * SciPy's 2-D subspace step does not get shorter than the four groups it has).
 *   half_bandwidth: of the reduced camera system, from the GLOBAL camera span (6 * max over ranks of cam_span + 5);
 *   band_exchange != 0: every rank's S is confined to that band (all shards have a pair list): n (hb + 1) + n doubles are
 *     exchanged instead of n^2 + n.  Both must be the same on every rank (decide them from all-reduced quantities).
 * All ranks see identical scalars and take identical decisions; the replicated cameras come out bit-identical: that
 * includes the choice between the two factorisation paths (a rank whose context is set to avoid the single launch, finds
 * no room in the co-residency budget, or sees its factorisation abandoned tells the others inside the exchanges above
 * and all switch together).
 * FATAL ERRORS ARE GROUP-WIDE: a non-zero return of the callback, or any error on one rank, makes THAT rank return; its
 * peers are then waiting inside their next all-reduce.  The caller must treat a failed mm_ba_trf_dist as fatal for the
 * whole group (abort / destroy the communicator, which releases the peers with an error, or rely on its timeout). */
typedef int (*mm_allreduce_fn)(void *user, double *buf /*dev*/, int64_t count);
typedef struct mm_dist {
    int32_t rank, world;   /* world <= 16 */
    int32_t half_bandwidth, band_exchange;
    mm_allreduce_fn allreduce;
    void *user;
} mm_dist;
size_t mm_ba_trf_dist_workspace_bytes(const mm_ba_problem *pb, int half_bandwidth);
int mm_ba_trf_dist(mm_ctx *ctx, const mm_ba_problem *pb, double *cams, double *pts, const mm_trf_params *prm,
                   mm_trf_report *report, mm_trf_row *log, int log_cap, void *ws, size_t ws_bytes, const mm_dist *dist);
/* SEVERAL INDEPENDENT PROBLEMS IN LOCK-STEP (the sliding-window adjustment; hook at reference
 * processor.py:395-408: the windows of one colour of the wavefront schedule share no camera).  Every kernel of mm_ba_trf's loop
 * is launched once per round for all problems (blockIdx.y = problem; per-problem grids and bodies identical to the
 * single-problem kernels), the host reads one 16-double mailbox per problem and round and takes each problem's accept /
 * reject / terminate decision exactly as mm_ba_trf does: cams[p] / pts[p] / reports[p] come out BIT-IDENTICAL to n_prob calls
 * of mm_ba_trf.  A reduced system that is not positive definite is retried with 100x the damping inside the batch, as
 * mm_ba_trf does; a problem whose factorisation was abandoned (info = -1) or that stays indefinite after six raises is
 * solved by mm_ba_trf itself after the batch, from its initial point (solved_alone[p] = 1; may be NULL);
 * if a problem cannot be batched at all (no co-observation pair list, reduced system outside the single-launch
 * factorisation, n_prob == 1) the call solves them one after the other.
 * pbs / cams / pts / ws / ws_bytes: HOST arrays of n_prob entries; ws[p] >= mm_ba_trf_batched_workspace_bytes(pbs[p]),
 * 256-byte aligned.  Problems must not share buffers.  (sync) */
size_t mm_ba_trf_batched_workspace_bytes(const mm_ba_problem *pb);
int mm_ba_trf_batched(mm_ctx *ctx, int n_prob, const mm_ba_problem *const *pbs, double *const *cams /*dev, in/out*/,
                      double *const *pts /*dev, in/out*/, const mm_trf_params *prm, mm_trf_report *reports /*host [n_prob]*/,
                      void *const *ws /*dev*/, const size_t *ws_bytes, int32_t *solved_alone /*host [n_prob] | NULL*/);
/* SPD solve A x = b by blocked Cholesky (f64 MFMA trailing updates).  A [n,n] row-major, lower triangle is
 * overwritten by L; b [nrhs,n] is overwritten by x.  half_bandwidth: A[i][j] == 0 whenever i - j > half_bandwidth
 * (pass n for a dense matrix); the factorisation and the substitutions skip blocks outside the band.
 * info [1] dev int32: 0 ok, k>0 = non-positive pivot at column k, -1 = the single-launch banded factorisation gave up
 * waiting for a block (its spin loops are bounded so that a scheduling surprise cannot hang the GPU; mm_ba_trf then
 * repeats the solve on the launch-per-column path).
 * Narrow bands (<= 15 blocks of 64) are factored by ONE data-flow scheduled launch, everything else by three launches
 * per block column; MM_CHOL_FUSED=0 in the environment forces the latter.  The workgroups of the single launch wait for
 * each other, so all of them must be resident (one per compute unit): every launch reserves its grid out of a
 * per-process budget of the device's compute units, held until the context's next synchronisation (mm_ctx_sync, the
 * end of mm_ba_trf); a launch that does not fit -- several contexts solving at once -- takes the per-column path. */
size_t mm_chol_workspace_bytes(int n);
/* Solution only: A x = b for ONE right-hand side, A destroyed (the layout of the factor it is overwritten with is
 * unspecified).  both_triangles != 0: both triangles of the band hold A on input; 0: only the lower one (the band of the
 * upper triangle is then filled from it first).  This freedom lets a narrow band be eliminated from BOTH ends at once
 * (chain of ~(nblk + bwb) / 2 dependent block columns instead of nblk): what the bundle adjustment calls per
 * trust-region iteration.  MM_CHOL_TWISTED=0 in the environment forces the one-ended elimination. */
int mm_chol_solve_sym(mm_ctx *ctx, double *A /*dev*/, int n, double *b /*dev [n]*/, int half_bandwidth, int both_triangles,
                      int32_t *info /*dev*/, void *ws, size_t ws_bytes);
int mm_chol_solve(mm_ctx *ctx, double *A /*dev*/, int n, double *b /*dev*/, int nrhs, int half_bandwidth,
                  int32_t *info /*dev*/, void *ws, size_t ws_bytes);

/* ---- keyframe gating front end (reference processor.py:61-110 keyframeTracking) ------------------------------------------
 * The arithmetic of the three calls is defined by oracle/frame_oracle.c (OpenCV's published algorithms made integer exact;
 * OpenCV itself is absent offline: parity unpinned) and reproduced bit for bit.
 * mm_pyr_down: one pyramid level, 5-tap [1 4 6 4 1]/16, reflect-101: dst [(h+1)/2, (w+1)/2].
 * mm_lk_track: cv2.calcOpticalFlowPyrLK(prev, next, pts, None, winSize, maxLevel, criteria) (processor.py:79) on pyramids
 *   built with mm_pyr_down.  prev_levels / next_levels / w / h / pitch: HOST arrays of `levels` (= maxLevel + 1) device
 *   pointers / sizes; pts [n,2] f32 dev.  next_pts [n,2] f32, status [n] u8, err [n] f32 (dev).  window 3..41, <= 8 levels.
 * mm_min_eig: Shi-Tomasi minimum eigenvalue map [h,w] f64 (cornerMinEigenVal scaling for 8-bit input), block size 1..15.
 * mm_corner_candidates: pixels above quality * max that survive the 3x3 non-maximum suppression -> value bit patterns
 *   [cap] i64 and flat positions [cap] i32 (unordered), count [1]; the caller sorts (value desc, position asc) and runs
 * mm_gftt_select (host): greedy minimum-distance selection of cv2.goodFeaturesToTrack (processor.py:104). */
int mm_pyr_down(mm_ctx *ctx, const uint8_t *src /*dev*/, int w, int h, int src_pitch, uint8_t *dst /*dev*/, int dst_pitch);
int mm_lk_track(mm_ctx *ctx, const uint8_t *const *prev_levels, const uint8_t *const *next_levels, const int *w, const int *h,
                const int *pitch, int levels, const float *pts, int n, int win_w, int win_h, int max_count,
                double epsilon_sq, float *next_pts, uint8_t *status, float *err);
int mm_min_eig(mm_ctx *ctx, const uint8_t *img /*dev*/, int w, int h, int pitch, int block_size, double *eig /*dev*/);
int mm_corner_candidates(mm_ctx *ctx, const double *eig, int w, int h, double quality, unsigned long long *max_bits /*dev [1]*/,
                         long long *val_bits /*dev [cap]*/, int32_t *pos /*dev [cap]*/, int cap, int32_t *count /*dev [1]*/);
int mm_gftt_select(const int32_t *pos /*host [n], sorted*/, int64_t n, int w, int h, int max_corners, double min_distance,
                   float *out /*host [cap,2]*/, int cap);
/* ---- increaseContrast + grey (reference processor.py:12-26, :357) and the PLY export (processor.py:480-485) ---------------
 * mm_increase_contrast: bgr [batch,h,w,3] u8 -> out (same shape), optionally grey [batch,h,w] of the result: fixed-point
 *   BGR <-> L*a*b* with the tables of meatmodeler_amd/frame_tables.py (gamma [256] u16, cbrt_tab [4096] u16, gamma_inv
 *   [4096] u8, device), CLAHE(clip_limit, tiles) on L.  mm_bgr_to_grey: (1868 B + 9617 G + 4899 R + 8192) >> 14.
 * mm_write_ply (host): binary little-endian PLY, double x / y / z per vertex. */
size_t mm_contrast_workspace_bytes(int batch, int w, int h, int tiles_x, int tiles_y);
int mm_increase_contrast(mm_ctx *ctx, const uint8_t *bgr, int batch, int w, int h, const uint16_t *gamma,
                         const uint16_t *cbrt_tab, const uint8_t *gamma_inv, double clip_limit, int tiles_x, int tiles_y,
                         uint8_t *out, uint8_t *grey /*dev|NULL*/, void *ws, size_t ws_bytes);
int mm_bgr_to_grey(mm_ctx *ctx, const uint8_t *bgr /*dev [n,3]*/, size_t n_pixels, uint8_t *grey /*dev [n]*/);
int mm_write_ply(const char *path, const double *xyz /*host [n,3]*/, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* MEATMODELER_H */
