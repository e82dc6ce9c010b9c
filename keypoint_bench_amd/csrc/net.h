// net.h -- extractor-network interface behind kpb_net_* and the .kpbw weight container parser.
#pragma once
#include "kpb_common.h"

#include <map>

struct kpb_net {
    kpb_ctx* ctx = nullptr;
    int arch = 0;
    int dim = 0;        // descriptor channels
    int desc_div = 1;   // descriptor map is (H/desc_div) x (W/desc_div)
    float* wdev = nullptr;                 // all repacked weights
    std::map<std::string, size_t> off;     // name -> float offset in wdev
    std::map<std::string, float> wscale;   // name -> power-of-two scale a split-f16 pack was made with (conv_mfma_h)
    kpb_buf act;                           // activations of the last forward
    int B = 0, H = 0, W = 0;
    virtual ~kpb_net() {}
    virtual int forward(const float* img, int batch, int H, int W, float* score_out, float* desc_out) = 0;
    virtual int desc_at(const float* pts, int pts_cols, int max_n, const int32_t* n_dev, float* out)
    {
        (void)pts; (void)pts_cols; (void)max_n; (void)n_dev; (void)out;
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_desc_at: this network materialises its descriptor map; use kpb_sample");
    }
    float* wp(const char* n) { return wdev + off.at(n); }
};

struct KpbwRec { char name[40]; uint32_t ndim; uint32_t dims[4]; uint32_t off; };

struct KpbwBlob {
    std::map<std::string, std::pair<const float*, std::vector<uint32_t>>> t;
    uint32_t arch = 0;
    bool parse(const void* blob, size_t len)
    {
        const unsigned char* p = static_cast<const unsigned char*>(blob);
        if (len < 16 || memcmp(p, "KPBWGT1\0", 8) != 0) return false;
        uint32_t n;
        memcpy(&arch, p + 8, 4);
        memcpy(&n, p + 12, 4);
        const size_t base = 16 + (size_t)n * sizeof(KpbwRec);
        if (base > len) return false;
        for (uint32_t i = 0; i < n; ++i) {
            KpbwRec r;
            memcpy(&r, p + 16 + (size_t)i * sizeof(KpbwRec), sizeof(KpbwRec));
            if (r.ndim > 4) return false;
            size_t cnt = 1;
            std::vector<uint32_t> d(r.dims, r.dims + r.ndim);
            for (uint32_t v : d) cnt *= v;
            if (base + 4 * ((size_t)r.off + cnt) > len) return false;
            char nm[41];                      // names may fill all 40 bytes of the field (no terminator then)
            memcpy(nm, r.name, 40);
            nm[40] = 0;
            t[nm] = {reinterpret_cast<const float*>(p + base + 4 * (size_t)r.off), d};
        }
        return true;
    }
    const float* get(const char* name, std::vector<uint32_t> dims) const
    {
        auto it = t.find(name);
        if (it == t.end() || it->second.second != dims) return nullptr;
        return it->second.first;
    }
};

// host-side staging of repacked tensors; upload() copies them into one device allocation
struct WeightStage {
    std::vector<float> host;
    std::map<std::string, size_t> off;
    std::map<std::string, float> wscale;
    void put(const std::string& name, const std::vector<float>& v)
    {
        while (host.size() % 64) host.push_back(0.0f);   // 256-byte alignment for scalar/vector loads
        off[name] = host.size();
        host.insert(host.end(), v.begin(), v.end());
    }
    void put_raw(const std::string& name, const float* p, size_t n) { put(name, std::vector<float>(p, p + n)); }
    int upload(kpb_net* net)
    {
        kpb_ctx* ctx = net->ctx;
        if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc(&net->wdev, host.size() * sizeof(float)) != hipSuccess)
            return kpb_fail(ctx, KPB_E_NOMEM, "kpb_net_create: weight allocation failed");
        if (hipMemcpy(net->wdev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return kpb_fail(ctx, KPB_E_HIP, "kpb_net_create: weight upload failed");
        net->off = off;
        net->wscale = wscale;
        return KPB_OK;
    }
};

// factories, one per architecture
int alike_create(kpb_ctx* ctx, const KpbwBlob& bl, kpb_net** out);
int superpoint_create(kpb_ctx* ctx, const KpbwBlob& bl, kpb_net** out);
int xfeat_create(kpb_ctx* ctx, const KpbwBlob& bl, kpb_net** out);
int disk_create(kpb_ctx* ctx, const KpbwBlob& bl, kpb_net** out);
