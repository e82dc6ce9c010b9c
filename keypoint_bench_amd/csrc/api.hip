// api.hip -- context lifetime and error reporting of libkpb.so (include/kpb.h).
#include "kpb_common.h"

char g_kpb_err[512] = {0};

extern "C" __attribute__((visibility("default"))) int kpb_version(void) { return KPB_VERSION; }

extern "C" __attribute__((visibility("default"))) const char* kpb_last_error(const kpb_ctx* ctx) { return ctx ? ctx->err : g_kpb_err; }

extern "C" __attribute__((visibility("default"))) int kpb_ctx_create(int device, void* stream, kpb_ctx** out)
{
    if (!out) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_ctx_create: null out pointer");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return kpb_fail(nullptr, KPB_E_HIP, "kpb_ctx_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= count)
        return kpb_fail(nullptr, KPB_E_INVALID, "kpb_ctx_create: device %d out of range (0..%d)", device, count - 1);
    hipDeviceProp_t prop;
    KPB_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return kpb_fail(nullptr, KPB_E_HIP, "kpb_ctx_create: device %d is %s; libkpb.so is built for gfx950 only", device,
                        prop.gcnArchName);
    KPB_HIP(nullptr, hipSetDevice(device));
    kpb_ctx* c = new kpb_ctx();
    c->device = device;
    c->stream = static_cast<hipStream_t>(stream);   // NULL = the device's default stream
    *out = c;
    return KPB_OK;
}

extern "C" __attribute__((visibility("default"))) void kpb_ctx_destroy(kpb_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (kpb_buf* b : {&ctx->ws_nms_state, &ctx->ws_nms_map, &ctx->ws_nms_list, &ctx->ws_cand, &ctx->ws_match, &ctx->ws_misc, &ctx->ws_sel})
        if (b->p) (void)hipFree(b->p);
    if (ctx->wait_ev) (void)hipEventDestroy(ctx->wait_ev);
    if (ctx->host_det) (void)hipHostFree(ctx->host_det);
    if (ctx->host_match) (void)hipHostFree(ctx->host_match);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->det_state && ctx->det_state_free) ctx->det_state_free(ctx->det_state);
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    delete ctx;
}

extern "C" __attribute__((visibility("default"))) int kpb_ctx_set_stream(kpb_ctx* ctx, void* stream)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_ctx_set_stream: null context");
    KPB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = static_cast<hipStream_t>(stream);
    return KPB_OK;
}

extern "C" __attribute__((visibility("default"))) int kpb_ctx_set_option(kpb_ctx* ctx, int option, int64_t value)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_ctx_set_option: null context");
    if (value < 0) return kpb_fail(ctx, KPB_E_INVALID, "kpb_ctx_set_option: negative value");
    switch (option) {
    case KPB_OPT_COVIS_STORE_BYTES: ctx->covis_store_bytes = (size_t)value; return KPB_OK;
    default: return kpb_fail(ctx, KPB_E_INVALID, "kpb_ctx_set_option: unknown option %d", option);
    }
}

extern "C" __attribute__((visibility("default"))) int kpb_sync(kpb_ctx* ctx)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_sync: null context");
    KPB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KPB_OK;
}

extern "C" __attribute__((visibility("default"))) int kpb_prof_enable(kpb_ctx* ctx, int on)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_prof_enable: null context");
    ctx->prof = on != 0;
    return KPB_OK;
}

// Writes one line per kernel: "<name> <calls> <total_ms>\n"; clears the records.  Synchronises the stream.
extern "C" __attribute__((visibility("default"))) int kpb_prof_report(kpb_ctx* ctx, char* buf, size_t cap)
{
    if (!ctx || !buf || cap == 0) return kpb_fail(ctx, KPB_E_INVALID, "kpb_prof_report: bad argument");
    KPB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<std::string> names;
    std::vector<double> ms;
    std::vector<long> calls;
    for (auto& r : ctx->prof_recs) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.e0, r.e1);
        size_t i = 0;
        for (; i < names.size(); ++i) if (names[i] == r.name) break;
        if (i == names.size()) { names.push_back(r.name); ms.push_back(0.0); calls.push_back(0); }
        ms[i] += t; calls[i] += 1;
        ctx->prof_pool.push_back(r.e0);
        ctx->prof_pool.push_back(r.e1);
    }
    ctx->prof_recs.clear();
    size_t off = 0;
    buf[0] = 0;
    for (size_t i = 0; i < names.size(); ++i) {
        int n = snprintf(buf + off, cap - off, "%s %ld %.6f\n", names[i].c_str(), calls[i], ms[i]);
        if (n < 0 || (size_t)n >= cap - off) break;
        off += (size_t)n;
    }
    return KPB_OK;
}
