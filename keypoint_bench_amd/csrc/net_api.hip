// net_api.hip -- kpb_net_* entry points of include/kpb.h: parse the weight container, dispatch on the architecture.
#include "net.h"

#define KPB_API extern "C" __attribute__((visibility("default")))

KPB_API int kpb_net_create(kpb_ctx* ctx, int arch, const void* blob, size_t len, kpb_net** out)
{
    if (!ctx || !out || !blob) return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_create: null argument");
    *out = nullptr;
    KpbwBlob bl;
    if (!bl.parse(blob, len) || (int)bl.arch != arch) return kpb_fail(ctx, KPB_E_WEIGHTS, "kpb_net_create: malformed .kpbw blob");
    switch (arch) {
    case KPB_ARCH_ALIKE: return alike_create(ctx, bl, out);
    case KPB_ARCH_SUPERPOINT: return superpoint_create(ctx, bl, out);
    case KPB_ARCH_XFEAT: return xfeat_create(ctx, bl, out);
    case KPB_ARCH_DISK: return disk_create(ctx, bl, out);
    default: return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_create: unknown arch %d", arch);
    }
}

KPB_API void kpb_net_destroy(kpb_net* net)
{
    if (!net) return;
    (void)hipSetDevice(net->ctx->device);
    (void)hipStreamSynchronize(net->ctx->stream);
    if (net->wdev) (void)hipFree(net->wdev);
    if (net->act.p) (void)hipFree(net->act.p);
    delete net;
}

KPB_API int kpb_net_desc_dim(const kpb_net* net) { return net ? net->dim : 0; }

KPB_API int kpb_net_desc_div(const kpb_net* net) { return net ? net->desc_div : 0; }

KPB_API int kpb_net_forward(kpb_net* net, const float* img_dev, int batch, int H, int W, float* score_out_dev,
                            float* desc_out_dev)
{
    if (!net) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_net_forward: null net");
    kpb_ctx* ctx = net->ctx;
    if (!img_dev || !score_out_dev || batch <= 0 || H <= 0 || W <= 0)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_forward: bad argument");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    return net->forward(img_dev, batch, H, W, score_out_dev, desc_out_dev);
}

KPB_API int kpb_net_desc_at(kpb_net* net, const float* pts_dev, int pts_cols, int max_n, const int32_t* n_dev,
                            float* out_dev)
{
    if (!net) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_net_desc_at: null net");
    kpb_ctx* ctx = net->ctx;
    if (max_n == 0) return KPB_OK;
    if (!pts_dev || !out_dev || pts_cols < 2 || max_n < 0) return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_desc_at: bad argument");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    return net->desc_at(pts_dev, pts_cols, max_n, n_dev, out_dev);
}
