// geometry.hip -- the metric side of tasks/*.py that consumes the matched keypoints (SURVEY 8(f) rank 3 and the
// FundamentalMatrix task of BASELINE configs[3]), gfx950 only.
//
//   epipolar_error   tasks/FundamentalMatrix.py:137-161   |x1^T F x0| / |(F x0)_xy| per match + mean / ratio / count
//
// One workgroup per image pair; a pair's matches (<= top_k rows) are a few KB, so everything after the loads lives in
// registers and LDS and the launch is latency-bound by construction.
#include "kpb_common.h"

namespace {

__device__ inline double block_sum(double v, double* red)
{
    for (int o = 32; o; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

struct EpiArgs {
    const float* kps0; int cols0;
    const float* kps1; int cols1;
    int max_k; const int32_t* k_dev;
    const float* fmat; int W, H, mode1; float th;
    float* err; float* stats;
};

// FundamentalMatrix.py:137-144.  kps0 rows are normalised (x, y, ...): kps0_wh = (x (W-1), y (H-1), 1).
// kps1 rows, as the reference's three matcher branches leave them:
//   mode1 0  brute force (120-122): the matched rows themselves, normalised (x, y, score), used as a 3-vector as is;
//   mode1 1  LightGlue (132-135): (x (W-1), y (H-1), 1);
//   mode1 2  optical flow (117-119): pixel (x, y) with a 1 appended.
__global__ __launch_bounds__(256) void epipolar_error(EpiArgs a)
{
    __shared__ double red[4];
    const int b = blockIdx.x;
    const int k = a.k_dev ? min(a.k_dev[b], a.max_k) : a.max_k;
    const float* f = a.fmat + (size_t)b * 9;
    const float sx = (float)(a.W - 1), sy = (float)(a.H - 1);
    double sum = 0.0, cnt = 0.0;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const float* p0 = a.kps0 + ((size_t)b * a.max_k + i) * a.cols0;
        const float* p1 = a.kps1 + ((size_t)b * a.max_k + i) * a.cols1;
        const float x0 = p0[0] * sx, y0 = p0[1] * sy;
        float l[3];
        for (int r = 0; r < 3; ++r) l[r] = fmaf(f[3 * r + 1], y0, f[3 * r] * x0) + f[3 * r + 2];      // I = F @ kps0_wh^T (140)
        float q0, q1, q2;
        if (a.mode1 == 0) { q0 = p1[0]; q1 = p1[1]; q2 = p1[2]; }
        else if (a.mode1 == 1) { q0 = p1[0] * sx; q1 = p1[1] * sy; q2 = 1.f; }
        else { q0 = p1[0]; q1 = p1[1]; q2 = 1.f; }
        const float e = fabsf(fmaf(q2, l[2], fmaf(q1, l[1], q0 * l[0])));                              // |diag(kps1 @ I)| (141-142)
        const float nrm = fmaxf(sqrtf(fmaf(l[1], l[1], l[0] * l[0])), 1e-6f);                          // norm(I[:-1]).clamp(1e-6) (143)
        const float err = e / nrm;
        a.err[(size_t)b * a.max_k + i] = err;
        sum += (double)err;
        cnt += err < a.th ? 1.0 : 0.0;
    }
    sum = block_sum(sum, red);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        float* s = a.stats + (size_t)b * 3;
        s[0] = k ? (float)(sum / k) : nanf("");          // torch.mean of an empty tensor is nan
        s[1] = k ? (float)(cnt / k) : nanf("");          // the reference divides by error.shape[0] (159): the host raises
        s[2] = (float)cnt;
    }
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int kpb_epipolar_error(
    kpb_ctx* ctx, const float* kps0_dev, int cols0, const float* kps1_dev, int cols1, int batch, int max_k, const int32_t* k_dev,
    const float* fmat_dev, int W, int H, int mode1, float th, float* out_err_dev, float* out_stats_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_epipolar_error: null context");
    if (batch <= 0 || max_k < 0 || cols0 < 2 || mode1 < 0 || mode1 > 2 || cols1 < (mode1 == 0 ? 3 : 2) || !fmat_dev || !out_stats_dev ||
        W < 1 || H < 1)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_epipolar_error: bad argument");
    if (max_k && (!kps0_dev || !kps1_dev || !out_err_dev)) return kpb_fail(ctx, KPB_E_INVALID, "kpb_epipolar_error: null buffer");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    EpiArgs a{kps0_dev, cols0, kps1_dev, cols1, max_k, k_dev, fmat_dev, W, H, mode1, th, out_err_dev, out_stats_dev};
    KPB_LAUNCH(ctx, "epipolar_error", epipolar_error, dim3(batch), dim3(256), 0, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}
