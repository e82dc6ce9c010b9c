// lk.hip -- the tensor Lucas-Kanade tracker of the reference:
//   kpb_lk_track   utils/matcher.py:7-142  OpticalFlow(params)(img1, img2, pts1, pts2)
//
// The reference unfolds six (win*win*C)-channel patch maps per pyramid level -- 1.6 GB each at 480x640 with the
// configured 21x21 window -- and grid_samples them at the keypoints, 40 times.  Sampling an unfolded map at (px, py)
// is, for window cell (ky, kx),
//     sum over the bilinear taps (xt, yt) of (px, py) inside the image   w_t * img[c][yt + ky - r][xt + kx - r]
// (zero outside the image), so nothing is materialised here: one wave per keypoint, the window's cells strided over
// the lanes, every cell read straight from the image and the two Sobel maps (L2-resident), five butterfly
// reductions per iteration.  fp32 throughout, the reference's formulas kept as they are -- including its update
// einsum 'bik,bk->bk', which sums the inverse over its second index instead of applying it.
#include "kpb_common.h"

namespace {

struct Taps { int x0, y0; float w[4]; bool ok[4]; };

// grid_sample(align_corners=True, bilinear, zeros) taps of pixel position (px, py), through the reference's normalise
// (matcher.py:109, 115) and ATen's unnormalise
__device__ __forceinline__ Taps make_taps(float px, float py, int H, int W)
{
    Taps t;
    const float gx = px / (float)(W - 1) * 2.0f - 1.0f, gy = py / (float)(H - 1) * 2.0f - 1.0f;
    const float ix = (gx + 1.0f) * ((float)(W - 1) / 2.0f), iy = (gy + 1.0f) * ((float)(H - 1) / 2.0f);
    const float fx = floorf(ix), fy = floorf(iy);
    const float wx = ix - fx, wy = iy - fy, ex = 1.0f - wx, sy = 1.0f - wy;
    t.x0 = (int)fx; t.y0 = (int)fy;
    t.w[0] = sy * ex; t.w[1] = sy * wx; t.w[2] = wy * ex; t.w[3] = wy * wx;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xt = t.x0 + (k & 1), yt = t.y0 + (k >> 1);
        t.ok[k] = xt >= 0 && xt < W && yt >= 0 && yt < H;
    }
    return t;
}

__device__ __forceinline__ float sample_cell(const float* __restrict__ img, int H, int W, const Taps& t, int oy, int ox)
{
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = t.y0 + (k >> 1) + oy, x = t.x0 + (k & 1) + ox;
        if (t.ok[k] && y >= 0 && y < H && x >= 0 && x < W) s += img[(size_t)y * W + x] * t.w[k];
    }
    return s;
}

// level images: out = avg_pool2d(img, k, k) (matcher.py:45); k = 1 is never launched
__global__ __launch_bounds__(256) void lk_avgpool(const float* __restrict__ img, float* __restrict__ out, int C, int H, int W, int k)
{
    const int Ho = H / k, Wo = W / k;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)C * Ho * Wo) return;
    const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), c = (int)(i / ((size_t)Wo * Ho));
    float s = 0.0f;
    for (int a = 0; a < k; ++a)
        for (int b = 0; b < k; ++b) s += img[(size_t)c * H * W + (size_t)(y * k + a) * W + x * k + b];
    out[i] = s / (float)(k * k);
}

// Sobel pair (matcher.py:23-24) as conv2d evaluates it: cross-correlation, zero padding 1 (87-90)
__global__ __launch_bounds__(256) void lk_sobel(const float* __restrict__ img, float* __restrict__ dx, float* __restrict__ dy, int C, int H, int W)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)C * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float* p = img + (i - (size_t)y * W - x);
    auto at = [&](int yy, int xx) { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? p[(size_t)yy * W + xx] : 0.0f; };
    float sx = 0.0f, sy = 0.0f;
    const float kx[3][3] = {{1, 0, -1}, {2, 0, -2}, {1, 0, -1}};
    const float ky[3][3] = {{1, 2, 1}, {0, 0, 0}, {-1, -2, -1}};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float v = at(y + a - 1, x + b - 1);
            sx += kx[a][b] * v; sy += ky[a][b] * v;
        }
    dx[i] = sx; dy[i] = sy;
}

// pixel positions and the randomly displaced, clamped start (matcher.py:51-61)
__global__ void lk_init(const float* pts1, const float* pts2, int stride, const float* unit, int n, int H, int W, float distance,
                        float* p1, float* p2, float* cur)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    p1[2 * i] = pts1[(size_t)i * stride] * (float)(W - 1); p1[2 * i + 1] = pts1[(size_t)i * stride + 1] * (float)(H - 1);
    const float x2 = pts2[(size_t)i * stride] * (float)(W - 1), y2 = pts2[(size_t)i * stride + 1] * (float)(H - 1);
    p2[2 * i] = x2; p2[2 * i + 1] = y2;
    cur[2 * i] = fminf(fmaxf(x2 + unit[2 * i] * distance, 10.0f), (float)(W - 10));
    cur[2 * i + 1] = fminf(fmaxf(y2 + unit[2 * i + 1] * distance, 10.0f), (float)(H - 10));
}

__device__ __forceinline__ float wave_sum(float v)
{
    v = kpb_wave_sum(v);
    return v;
}

struct LevelArgs {
    const float* img1; const float* img2; const float* dx2; const float* dy2;
    const float* p1;    // [n][2] full-resolution pixel positions in image 1
    float* cur;         // [n][2] full-resolution estimate in image 2, updated in place
    int n, C, H, W, win, iters;
    float scale;        // 2^(level index): positions are divided by it on entry and multiplied on exit (79-85)
};

// optical_flow_level (matcher.py:77-133): one wave per keypoint; the patch of image 1 stays in LDS
__global__ __launch_bounds__(256) void lk_level(LevelArgs a)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wv;
    if (i >= a.n) return;
    const int r = a.win / 2, WW = a.win * a.win, E = a.C * WW;
    const size_t P = (size_t)a.H * a.W;
    float* patch = lds + (size_t)wv * E;
    {
        const Taps t1 = make_taps(a.p1[2 * i] / a.scale, a.p1[2 * i + 1] / a.scale, a.H, a.W);
        for (int e = lane; e < E; e += 64) {
            const int c = e / WW, k = e - c * WW, ky = k / a.win, kx = k - ky * a.win;
            patch[e] = sample_cell(a.img1 + c * P, a.H, a.W, t1, ky - r, kx - r);
        }
    }
    float px = a.cur[2 * i] / a.scale, py = a.cur[2 * i + 1] / a.scale;
    for (int it = 0; it < a.iters; ++it) {
        const Taps t = make_taps(px, py, a.H, a.W);
        float g00 = 0.f, g01 = 0.f, g11 = 0.f, b0 = 0.f, b1 = 0.f;
        for (int e = lane; e < E; e += 64) {
            const int c = e / WW, k = e - c * WW, ky = k / a.win, kx = k - ky * a.win;
            const float v = sample_cell(a.img2 + c * P, a.H, a.W, t, ky - r, kx - r);
            const float jx = sample_cell(a.dx2 + c * P, a.H, a.W, t, ky - r, kx - r);
            const float jy = sample_cell(a.dy2 + c * P, a.H, a.W, t, ky - r, kx - r);
            const float dI = patch[e] - v;                                  // 117
            g00 += jx * jx; g01 += jx * jy; g11 += jy * jy;                 // 121
            b0 += dI * jx; b1 += dI * jy;                                   // 122
        }
        g00 = wave_sum(g00); g01 = wave_sum(g01); g11 = wave_sum(g11); b0 = wave_sum(b0); b1 = wave_sum(b1);
        const float det = g00 * g11 - g01 * g01;                            // 123
        if (det > 1e-6f) {
            const float i00 = g11 / det, i01 = -g01 / det, i11 = g00 / det; // 124
            px = px - (i00 + i01) * b0;                                     // 125: 'bik,bk->bk' sums the inverse over i
            py = py - (i01 + i11) * b1;
        }
    }
    if (lane == 0) { a.cur[2 * i] = px * a.scale; a.cur[2 * i + 1] = py * a.scale; }
}

__global__ void lk_finish(const float* cur, const float* p2, int n, float* out_pts, float* out_err)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float dx = cur[2 * i] - p2[2 * i], dy = cur[2 * i + 1] - p2[2 * i + 1];
    out_pts[2 * i] = cur[2 * i]; out_pts[2 * i + 1] = cur[2 * i + 1];
    out_err[i] = fminf(sqrtf(dx * dx + dy * dy), 8.0f);                     // 73
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int kpb_lk_track(
    kpb_ctx* ctx, const float* img1_dev, const float* img2_dev, int C, int H, int W, const float* pts1_dev, const float* pts2_dev,
    int pts_stride, const float* unit_dev, int n, const kpb_lk_params* prm, float* out_pts_dev, float* out_err_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_lk_track: null context");
    if (!img1_dev || !img2_dev || !prm || C <= 0 || H <= 20 || W <= 20 || n < 0 || pts_stride < 2)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_lk_track: bad argument");
    if (prm->win_size < 1 || prm->win_size > 31 || !(prm->win_size & 1) || prm->levels < 1 || prm->levels > 4 || prm->iterations < 0)
        return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_lk_track: win_size must be odd and <= 31, levels 1..4");
    if ((H >> (prm->levels - 1)) < 2 || (W >> (prm->levels - 1)) < 2)
        return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_lk_track: image too small for %d levels", prm->levels);
    if (n == 0) return KPB_OK;
    if (!pts1_dev || !pts2_dev || !unit_dev || !out_pts_dev || !out_err_dev)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_lk_track: null buffer");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)H * W, plane = (size_t)C * P;
    // workspace: p1, p2, cur [n][2]; pooled image 1 / image 2, dx, dy of image 2 (each at most one full-size plane set)
    if (int rc = kpb_reserve(ctx, ctx->ws_misc, ((size_t)6 * n + 4 * plane) * sizeof(float))) return rc;
    float* p1 = static_cast<float*>(ctx->ws_misc.p);
    float* p2 = p1 + 2 * (size_t)n;
    float* cur = p2 + 2 * (size_t)n;
    float* l1 = cur + 2 * (size_t)n;
    float* l2 = l1 + plane;
    float* dx = l2 + plane;
    float* dy = dx + plane;
    hipStream_t st = ctx->stream;
    KPB_LAUNCH(ctx, "lk_init", lk_init, dim3(cdiv(n, 256)), dim3(256), 0, st, pts1_dev, pts2_dev, pts_stride, unit_dev, n, H, W, prm->distance, p1, p2, cur);
    const size_t lds = (size_t)4 * C * prm->win_size * prm->win_size * sizeof(float);
    if (lds > 64 * 1024) return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_lk_track: window of %d x %d x %d does not fit the patch buffer", prm->win_size, prm->win_size, C);
    for (int lv = 0; lv < prm->levels; ++lv) {
        const int idx = prm->levels - lv - 1;
        const int k = idx == 0 ? 1 : 2 * idx;       // build_pyramid (45): level i > 0 is avg_pool2d(img, 2i, 2i) ...
        const int Hl = H / k, Wl = W / k;
        const float* a = img1_dev;
        const float* b = img2_dev;
        if (k > 1) {
            const unsigned g = (unsigned)(((size_t)C * Hl * Wl + 255) / 256);
            KPB_LAUNCH(ctx, "lk_avgpool", lk_avgpool, dim3(g), dim3(256), 0, st, img1_dev, l1, C, H, W, k);
            KPB_LAUNCH(ctx, "lk_avgpool", lk_avgpool, dim3(g), dim3(256), 0, st, img2_dev, l2, C, H, W, k);
            a = l1; b = l2;
        }
        KPB_LAUNCH(ctx, "lk_sobel", lk_sobel, dim3((unsigned)(((size_t)C * Hl * Wl + 255) / 256)), dim3(256), 0, st, b, dx, dy, C, Hl, Wl);
        LevelArgs la{a, b, dx, dy, p1, cur, n, C, Hl, Wl, prm->win_size, prm->iterations, (float)(1 << idx)};   // ... while positions scale by 2^i (66, 79)
        KPB_LAUNCH(ctx, "lk_level", lk_level, dim3(cdiv(n, 4)), dim3(256), lds, st, la);
    }
    KPB_LAUNCH(ctx, "lk_finish", lk_finish, dim3(cdiv(n, 256)), dim3(256), 0, st, cur, p2, n, out_pts_dev, out_err_dev);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}
