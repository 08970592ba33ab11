// Ping-pong between two workgroups through global memory: hop latency by cache-scope bits (sc0 / sc1) and by placement
// (same XCD: workgroup ids congruent mod 8; different XCDs).  hipcc --offload-arch=gfx950 -O3 -o xcd_pingpong xcd_pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LD(BITS) static __device__ __forceinline__ unsigned long long ld_##BITS
template <int MODE>
__device__ __forceinline__ unsigned long long ld(const unsigned long long *p) {
    unsigned long long v;
    if (MODE == 0) asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 5) asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE>
__device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) {
    if (MODE == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (MODE == 4) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (MODE == 5) asm volatile("global_store_dwordx2 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
}

template <int LM, int SM>
__global__ void pingpong(unsigned long long *w, int a, int b, int rounds, long long *out, int *xcc) {
    const int bx = blockIdx.x;
    if (threadIdx.x == 0) xcc[bx] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID [3:0]
    if (bx != a && bx != b) return;
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    int fail = 0;
    for (int i = 1; i <= rounds && !fail; ++i) {
        if (bx == a) {
            st<SM>(w, (unsigned long long)i);
            long spin = 0;
            while (ld<LM>(w + 32) != (unsigned long long)i)
                if (++spin > 2000000) { fail = 1; break; }
        } else {
            long spin = 0;
            while (ld<LM>(w) != (unsigned long long)i)
                if (++spin > 2000000) { fail = 1; break; }
            st<SM>(w + 32, (unsigned long long)i);
        }
    }
    if (bx == a) {
        out[0] = wall_clock64() - t0;
        out[1] = fail;
    } else {
        out[2] = fail;
    }
}

template <int LM, int SM>
void run(const char *name, int a, int b, unsigned long long *w, long long *out, int *xcc) {
    const int rounds = 2000;
    hipMemset(w, 0, 4096);
    hipMemset(out, 0, 64);
    hipLaunchKernelGGL((pingpong<LM, SM>), dim3(64), dim3(64), 0, 0, w, a, b, rounds, out, xcc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    long long h[3];
    int hx[64];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
    // wall_clock64: 100 MHz
    printf("%-28s wg %2d (xcc %d) <-> wg %2d (xcc %d): %7.3f us per hop%s\n", name, a, hx[a], b, hx[b], h[0] * 10.0 / 1000.0 / rounds / 2.0,
           (h[1] || h[2]) ? "   ** NEVER SEEN (stale) **" : "");
}

int main() {
    unsigned long long *w;
    long long *out;
    int *xcc;
    hipMalloc(&w, 4096);
    hipMalloc(&out, 64);
    hipMalloc(&xcc, 64 * sizeof(int));
    for (int rep = 0; rep < 2; ++rep) {
        const int a = 0, b = rep == 0 ? 8 : 1;
        printf("--- %s ---\n", rep == 0 ? "same XCD expected (0, 8)" : "different XCDs expected (0, 1)");
        run<2, 2>("ld sc1 / st sc1", a, b, w, out, xcc);
        run<1, 1>("ld sc0 / st sc0", a, b, w, out, xcc);
        run<1, 0>("ld sc0 / st plain", a, b, w, out, xcc);
        run<1, 2>("ld sc0 / st sc1", a, b, w, out, xcc);
        run<2, 1>("ld sc1 / st sc0", a, b, w, out, xcc);
        run<3, 3>("ld sc0 sc1 / st sc0 sc1", a, b, w, out, xcc);
        run<5, 1>("ld sc0 nt / st sc0", a, b, w, out, xcc);
        run<0, 0>("ld plain / st plain", a, b, w, out, xcc);
    }
    return 0;
}
