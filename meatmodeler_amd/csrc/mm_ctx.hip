// Context, error text and HIP-event timers of the C ABI (include/meatmodeler.h).
#include "mm_common.h"

extern "C" {

int mm_abi_version(void) { return 3; }

int mm_ctx_create(int device, void *hip_stream, mm_ctx **out) {
    if (!out) return MM_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return MM_ERR_HIP;
    mm_ctx *c = new mm_ctx();
    c->device = device;
    c->stream = (hipStream_t)hip_stream;
    c->err[0] = 0;
    if (hipSetDevice(device) != hipSuccess) {
        delete c;
        return MM_ERR_HIP;
    }
    if (hipDeviceGetAttribute(&c->cu_count, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) c->cu_count = 0;
    *out = c;
    return MM_OK;
}

void mm_ctx_destroy(mm_ctx *ctx) {
    if (!ctx) return;
    mm_chol_release_budget(ctx);
    for (auto &r : ctx->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : ctx->pool) (void)hipEventDestroy(e);
    if (ctx->cam_tab) (void)hipFree(ctx->cam_tab);
    if (ctx->host_board) (void)hipHostFree(ctx->host_board);
    if (ctx->batch_dev) (void)hipFree(ctx->batch_dev);
    if (ctx->batch_host) (void)hipHostFree(ctx->batch_host);
    if (ctx->fused_ev) (void)hipEventDestroy(ctx->fused_ev);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    delete ctx;
}

// ---- per-launch profiling ---------------------------------------------------------------------------------------------
int mm_profile_enable(mm_ctx *ctx, int on) {
    if (!ctx) return MM_ERR_ARG;
    if (on) {
        for (auto &r : ctx->recs) {
            ctx->pool.push_back(r.a);
            ctx->pool.push_back(r.b);
        }
        ctx->recs.clear();
    }
    ctx->prof = on < 0 ? 0 : on;
    return MM_OK;
}

int mm_profile_select(mm_ctx *ctx, const char *name) {
    if (!ctx || !name || strlen(name) >= sizeof(ctx->prof_only)) return MM_ERR_ARG;
    strcpy(ctx->prof_only, name);
    return MM_OK;
}

int mm_profile_report(mm_ctx *ctx, char *buf, size_t buf_len) {
    if (!ctx || !buf || buf_len < 64) return MM_ERR_ARG;
    MM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    struct Agg {
        const char *name;
        long n;
        double ms;
    };
    std::vector<Agg> agg;
    for (auto &r : ctx->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        bool found = false;
        for (auto &a : agg)
            if (strcmp(a.name, r.name) == 0) {
                a.n++;
                a.ms += ms;
                found = true;
                break;
            }
        if (!found) agg.push_back(Agg{r.name, 1, (double)ms});
    }
    size_t off = 0;
    buf[0] = 0;
    for (auto &a : agg) {
        int w = snprintf(buf + off, buf_len - off, "%s %ld %.6f\n", a.name, a.n, a.ms);
        if (w < 0 || (size_t)w >= buf_len - off) return mm_fail(ctx, MM_ERR_WORKSPACE, "mm_profile_report: buffer too small");
        off += (size_t)w;
    }
    return (int)agg.size();
}

const char *mm_last_error(mm_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int mm_ctx_sync(mm_ctx *ctx) {
    if (!ctx) return MM_ERR_ARG;
    MM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    mm_chol_release_budget(ctx);      // nothing of this context is in flight any more
    return MM_OK;
}

long long mm_ctx_control(mm_ctx *ctx, int what, long long value) {
    if (!ctx) return MM_ERR_ARG;
    switch (what) {
        case MM_CTL_CHOL_FORCE_ABANDON: ctx->debug_abandon = value < 0 ? 0 : (int)value; return MM_OK;
        case MM_CTL_CHOL_LAST_PATH: return ctx->chol_last_path;
        case MM_CTL_CHOL_RESERVED: return ctx->fused_wgs;
        case MM_CTL_CU_COUNT: return ctx->cu_count;
        case MM_CTL_CHOL_AVOID_FUSED: {      // value < 0: query only; returns the PREVIOUS setting (0 / 1)
            const long long prev = ctx->chol_avoid_fused ? 1 : 0;
            if (value >= 0) ctx->chol_avoid_fused = value != 0;
            return prev;
        }
        case MM_CTL_LINK_LAST_VARIANT: return ctx->link_last_variant;
        case MM_CTL_BATCH_LAST: return ctx->batch_last;
        default: return mm_fail(ctx, MM_ERR_ARG, "mm_ctx_control: unknown request %d", what);
    }
}

struct mm_timer {
    hipEvent_t a, b;
};

int mm_timer_create(mm_ctx *ctx, void **timer_out) {
    if (!ctx || !timer_out) return MM_ERR_ARG;
    mm_timer *t = new mm_timer();
    MM_HIP(ctx, hipEventCreate(&t->a));
    MM_HIP(ctx, hipEventCreate(&t->b));
    *timer_out = t;
    return MM_OK;
}
int mm_timer_start(mm_ctx *ctx, void *timer) {
    if (!ctx || !timer) return MM_ERR_ARG;
    MM_HIP(ctx, hipEventRecord(((mm_timer *)timer)->a, ctx->stream));
    return MM_OK;
}
int mm_timer_stop(mm_ctx *ctx, void *timer) {
    if (!ctx || !timer) return MM_ERR_ARG;
    MM_HIP(ctx, hipEventRecord(((mm_timer *)timer)->b, ctx->stream));
    return MM_OK;
}
int mm_timer_elapsed_ms(mm_ctx *ctx, void *timer, float *ms_out) {
    if (!ctx || !timer || !ms_out) return MM_ERR_ARG;
    mm_timer *t = (mm_timer *)timer;
    MM_HIP(ctx, hipEventSynchronize(t->b));
    MM_HIP(ctx, hipEventElapsedTime(ms_out, t->a, t->b));
    return MM_OK;
}
void mm_timer_destroy(mm_ctx *ctx, void *timer) {
    (void)ctx;
    if (!timer) return;
    mm_timer *t = (mm_timer *)timer;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
}

}  // extern "C"
