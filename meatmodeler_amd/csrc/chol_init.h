// chol_init.h -- the fills a single-launch factorisation needs before it starts, as a device function.
// chol.hip's chol_init_kernel runs it; schur_prepare_kernel (schur.hip) runs it as well when mm_ba_trf hands the whole
// reduced solve to one entry point (mm_ba_schur_solve_damped): one launch less per solve.  Arguments: mm_chol_init_plan.
#pragma once
#include "mm_common.h"

constexpr int MM_CHOL_NB = 64;                            // block size of the factorisation (chol.hip's NB)
constexpr unsigned long long MM_CHOL_SENTINEL = ~0ull;    // the NaN every polled word starts as

__device__ __forceinline__ void mm_chol_init_body(const mm_chol_init_args &a, const unsigned bx, const unsigned gx) {
    // (lpub covers both hand-over buffers: the streamed pieces of the diagonal blocks and the sub-diagonal blocks)
    const size_t i = (size_t)bx * 256 + threadIdx.x, stride = (size_t)gx * 256;
    if (i == 0) a.info[0] = 0;
    for (size_t k = i; k < a.nflags; k += stride) a.flags[k] = 0;
    // (16 bytes per store: both buffers are 256-byte aligned and hold an even number of words -- 24 MB at the bench shape)
    const ulonglong2 s2 = make_ulonglong2(MM_CHOL_SENTINEL, MM_CHOL_SENTINEL);
    for (size_t k = i; k < a.nsent / 2; k += stride) reinterpret_cast<ulonglong2 *>(a.sentinel_buf)[k] = s2;
    if ((a.nsent & 1) && i == 0) a.sentinel_buf[a.nsent - 1] = MM_CHOL_SENTINEL;
    for (size_t k = i; k < a.nlpub / 2; k += stride) reinterpret_cast<ulonglong2 *>(a.lpub)[k] = s2;
    if ((a.nlpub & 1) && i == 0) a.lpub[a.nlpub - 1] = MM_CHOL_SENTINEL;
    for (size_t k = i; k < a.nblk * 1024; k += stride) {
        const size_t b = k >> 10, d = (k >> 8) & 3, e = k & 255;
        a.Linv[b * MM_CHOL_NB * MM_CHOL_NB + (16 * d + (e >> 4)) * MM_CHOL_NB + 16 * d + (e & 15)] = MM_CHOL_SENTINEL;
    }
}
