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
    for (size_t k = i; k < a.nsent; k += stride) a.sentinel_buf[k] = MM_CHOL_SENTINEL;
    for (size_t k = i; k < a.nlpub; k += stride) a.lpub[k] = MM_CHOL_SENTINEL;
    for (size_t k = i; k < a.nblk * 1024; k += stride) {
        const size_t b = k >> 10, d = (k >> 8) & 3, e = k & 255;
        a.Linv[b * MM_CHOL_NB * MM_CHOL_NB + (16 * d + (e >> 4)) * MM_CHOL_NB + 16 * d + (e & 15)] = MM_CHOL_SENTINEL;
    }
}
