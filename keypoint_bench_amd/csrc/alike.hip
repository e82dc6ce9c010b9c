// alike.hip -- placeholder entry points (replaced by the ALIKE forward kernels)
#include "kpb_common.h"
struct kpb_net { kpb_ctx* ctx; };
extern "C" __attribute__((visibility("default"))) int kpb_net_create(kpb_ctx* ctx, int, const void*, size_t, kpb_net**) { return kpb_fail(ctx, KPB_E_INVALID, "not built"); }
extern "C" __attribute__((visibility("default"))) void kpb_net_destroy(kpb_net*) {}
extern "C" __attribute__((visibility("default"))) int kpb_net_desc_dim(const kpb_net*) { return 0; }
extern "C" __attribute__((visibility("default"))) int kpb_net_forward(kpb_net* n, const float*, int, int, int, float*, float*) { return kpb_fail(n ? n->ctx : nullptr, KPB_E_INVALID, "not built"); }
extern "C" __attribute__((visibility("default"))) int kpb_net_desc_at(kpb_net* n, const float*, int, int, const int32_t*, float*) { return kpb_fail(n ? n->ctx : nullptr, KPB_E_INVALID, "not built"); }
