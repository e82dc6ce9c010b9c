// preprocess.hip -- the per-image input transform of the reference's datasets, after decoding:
//   kpb_preprocess   datasets/hpatches.py:47-69: BGR -> RGB (59-60), astype(float32) / 255 (59-60),
//                    cv2.resize(img, (size, size)) = INTER_LINEAR (66-67), HWC -> CHW (74-75);
//                    datasets/megadepth.py:312-313 (transforms.ToTensor: / 255, HWC -> CHW) is the no-resize case.
// One thread per output pixel, all three channels: 12 bytes read from the decoded image, 12 written, batched over
// equally sized images.  cv2 is not in this image, so the resize follows OpenCV's documented INTER_LINEAR arithmetic
// for float32 images (half-pixel centres, clamped taps, horizontal then vertical blend) and is "parity unpinned".
#include "kpb_common.h"

namespace {

struct PrepArgs {
    const uint8_t* src; float* out;
    int Hs, Ws, Hd, Wd, swap_rb;
    double scale_x, scale_y;
};

// source index and weight of destination index d (cv::resize, INTER_LINEAR): f = (d + 0.5) * scale - 0.5 in double,
// rounded to float; s = floor(f); f -= s; clamped at both ends
__device__ __forceinline__ void lin_coord(int d, double scale, int n_src, int& s, float& f)
{
    f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.0f; }
    if (s >= n_src - 1) { s = n_src - 1; f = 0.0f; }
}

__global__ __launch_bounds__(256) void preprocess(PrepArgs a)
{
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.Wd || y >= a.Hd) return;
    int sx, sy; float fx, fy;
    lin_coord(x, a.scale_x, a.Ws, sx, fx);
    lin_coord(y, a.scale_y, a.Hs, sy, fy);
    const int sx1 = min(sx + 1, a.Ws - 1), sy1 = min(sy + 1, a.Hs - 1);
    const uint8_t* img = a.src + (size_t)b * a.Hs * a.Ws * 3;
    const uint8_t* p00 = img + ((size_t)sy * a.Ws + sx) * 3;
    const uint8_t* p01 = img + ((size_t)sy * a.Ws + sx1) * 3;
    const uint8_t* p10 = img + ((size_t)sy1 * a.Ws + sx) * 3;
    const uint8_t* p11 = img + ((size_t)sy1 * a.Ws + sx1) * 3;
    const float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
    const size_t P = (size_t)a.Hd * a.Wd;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int cs = a.swap_rb ? 2 - c : c;
        const float v00 = (float)p00[cs] / 255.0f, v01 = (float)p01[cs] / 255.0f;       // hpatches.py:59-60
        const float v10 = (float)p10[cs] / 255.0f, v11 = (float)p11[cs] / 255.0f;
        const float r0 = v00 * a0 + v01 * a1, r1 = v10 * a0 + v11 * a1;                 // horizontal pass
        a.out[((size_t)b * 3 + c) * P + (size_t)y * a.Wd + x] = r0 * b0 + r1 * b1;      // vertical pass
    }
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int kpb_preprocess(kpb_ctx* ctx, const uint8_t* src_dev, int batch, int Hs, int Ws,
                                                                      int swap_rb, int Hd, int Wd, float* out_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_preprocess: null context");
    if (!src_dev || !out_dev || batch <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_preprocess: bad argument");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    PrepArgs a{src_dev, out_dev, Hs, Ws, Hd, Wd, swap_rb ? 1 : 0, (double)Ws / (double)Wd, (double)Hs / (double)Hd};
    KPB_LAUNCH(ctx, "preprocess", preprocess, dim3(cdiv(Wd, 64), cdiv(Hd, 4), batch), dim3(256), 0, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}
