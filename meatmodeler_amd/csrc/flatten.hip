// managePoints on the device (gfx950): the flat observation arrays the bundle adjustment takes, from the track CSR.
//
// Replaces processor.managePoints (reference processor.py:264-291): for every track in list order its 3-D point once,
// and for every (frame, coordinate) entry of the track -- in insertion order -- the 2-D coordinate, the frame index and
// the index of the track's point.  With the tracks as a CSR over observations (mm_link_tracks_device) that is a gather:
//   coords[o]        = kp_xy[obs_frame[o'], obs_kp[o']]   (f32 key point coordinates widened to f64, as np.array does)
//   frame_indices[o] = obs_frame[o'] - frame_offset
//   point_indices[o] = position of the track in the selection
// for a SELECTION of tracks (all of them, a contiguous range = a rank's shard, or an index list = a sliding window).
// HBM-bound streaming: 16 B read + 24 B written per observation; no host round trip, no framework kernels.
#include "mm_common.h"

namespace {

constexpr int FL_THREADS = 1024;
constexpr int FL_IPT = 8;

// out_ptr[i] = sum_{j < i} len(track sel[j]), i = 0 .. n_sel (64-bit); one workgroup, chunked scan with a running base
__global__ __launch_bounds__(FL_THREADS) void flatten_offsets_kernel(const int32_t *__restrict__ track_ptr,
                                                                     const int32_t *__restrict__ sel, int64_t n_sel,
                                                                     int64_t *__restrict__ out_ptr) {
    __shared__ long long s_wave[FL_THREADS / 64];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < n_sel; c0 += (int64_t)FL_THREADS * FL_IPT) {
        long long len[FL_IPT], local = 0;
#pragma unroll
        for (int q = 0; q < FL_IPT; ++q) {
            const int64_t i = c0 + (int64_t)tid * FL_IPT + q;
            long long l = 0;
            if (i < n_sel) {
                const int t = sel[i];
                l = (long long)track_ptr[t + 1] - track_ptr[t];
            }
            len[q] = local;      // exclusive inside the thread
            local += l;
        }
        long long incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave[w] = incl;
        __syncthreads();
        long long wave_base = 0;
        for (int i = 0; i < w; ++i) wave_base += s_wave[i];
        const long long base = s_base + wave_base + incl - local;
#pragma unroll
        for (int q = 0; q < FL_IPT; ++q) {
            const int64_t i = c0 + (int64_t)tid * FL_IPT + q;
            if (i < n_sel) out_ptr[i] = base + len[q];
        }
        __syncthreads();
        if (tid == FL_THREADS - 1) s_base = base + local;
        __syncthreads();
    }
    if (tid == 0) out_ptr[n_sel] = s_base;
}

// one thread per output observation; the track it belongs to by binary search in the selection's offsets
__global__ __launch_bounds__(256) void flatten_fill_kernel(const int32_t *__restrict__ track_ptr, const int32_t *__restrict__ obs_frame,
                                                           const int32_t *__restrict__ obs_kp, const float *__restrict__ kp_xy,
                                                           int cap, const int32_t *__restrict__ sel, int t_lo, int64_t n_sel,
                                                           const int64_t *__restrict__ out_ptr, int64_t n_obs, int frame_offset,
                                                           double *__restrict__ coords, int32_t *__restrict__ fi,
                                                           int32_t *__restrict__ pi) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_obs) return;
    int64_t src, p;
    if (sel) {
        int64_t lo = 0, hi = n_sel;      // largest p with out_ptr[p] <= o
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (out_ptr[mid] <= o) lo = mid;
            else hi = mid;
        }
        p = lo;
        src = (int64_t)track_ptr[sel[p]] + (o - out_ptr[p]);
    } else {      // contiguous range of tracks: the observations are contiguous too
        const int64_t first = track_ptr[t_lo];
        src = first + o;
        int64_t lo = 0, hi = n_sel;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)track_ptr[t_lo + mid] - first <= o) lo = mid;
            else hi = mid;
        }
        p = lo;
    }
    const int f = obs_frame[src], k = obs_kp[src];
    const float2 xy = reinterpret_cast<const float2 *>(kp_xy)[(size_t)f * cap + k];
    reinterpret_cast<double2 *>(coords)[o] = make_double2((double)xy.x, (double)xy.y);
    fi[o] = f - frame_offset;
    pi[o] = (int32_t)p;
}

}  // namespace

extern "C" {

int mm_flatten_offsets(mm_ctx *ctx, const int32_t *track_ptr, const int32_t *sel, int64_t n_sel, int64_t *out_ptr) {
    if (!ctx) return MM_ERR_ARG;
    if (n_sel < 0 || !track_ptr || !out_ptr || (n_sel > 0 && !sel)) return mm_fail(ctx, MM_ERR_ARG, "mm_flatten_offsets: bad argument");
    MM_LAUNCH(ctx, "flatten_offsets_kernel", flatten_offsets_kernel, dim3(1), dim3(FL_THREADS), 0, track_ptr, sel, n_sel, out_ptr);
    return MM_OK;
}

int mm_flatten_tracks(mm_ctx *ctx, const int32_t *track_ptr, const int32_t *obs_frame, const int32_t *obs_kp, const float *kp_xy,
                      int cap, const int32_t *sel, int t_lo, int64_t n_sel, const int64_t *out_ptr, int64_t n_obs, int frame_offset,
                      double *coords, int32_t *frame_indices, int32_t *point_indices) {
    if (!ctx) return MM_ERR_ARG;
    if (n_sel < 0 || n_obs < 0 || cap <= 0 || t_lo < 0) return mm_fail(ctx, MM_ERR_ARG, "mm_flatten_tracks: bad argument");
    if (n_obs == 0) return MM_OK;
    if (!track_ptr || !obs_frame || !obs_kp || !kp_xy || !coords || !frame_indices || !point_indices || (sel && !out_ptr) || n_sel == 0)
        return mm_fail(ctx, MM_ERR_ARG, "mm_flatten_tracks: null pointer");
    if (((uintptr_t)coords & 15) || ((uintptr_t)kp_xy & 7)) return mm_fail(ctx, MM_ERR_ARG, "mm_flatten_tracks: alignment");
    MM_LAUNCH(ctx, "flatten_fill_kernel", flatten_fill_kernel, dim3((unsigned)((n_obs + 255) / 256)), dim3(256), 0, track_ptr, obs_frame,
              obs_kp, kp_xy, cap, sel, t_lo, n_sel, out_ptr, n_obs, frame_offset, coords, frame_indices, point_indices);
    return MM_OK;
}

}  // extern "C"
