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

// ================================================================================================ RANSAC homography
// cv2.findHomography(pts0, pts1, cv2.RANSAC) as tasks/MHA.py:45-47 calls it (defaults: threshold 3 px, 2000 iterations,
// confidence 0.995), restated from OpenCV's published algorithm -- PARITY UNPINNED (cv2 is absent and draws from its own
// RNG); the numpy restatement is oracle/geometry_ref.py, hypothesis for hypothesis the same.  One 256-thread workgroup
// per image pair:
//   * the matched rows are scaled to pixels in fp32 exactly as MHA.py:41-42 does and parked in LDS;
//   * a round = 256 hypotheses, one per thread: four distinct indices from a counter-based generator, the exact model
//     in closed form (projective basis: two 3x3 adjugates, registers only), OpenCV's degeneracy tests, then every thread
//     scores ALL matches against its own model (LDS broadcast reads);
//   * the workgroup keeps the model with strictly more inliers (ties: the lower iteration), adapts the iteration count
//     as RANSACUpdateNumIters does, and stops at the first round boundary past it;
//   * refit on the inliers: normalised inhomogeneous DLT (8x8 normal equations, 44 block-reduced sums) and up to ten
//     Levenberg-Marquardt steps on the forward reprojection error.  All estimator arithmetic is fp64.
namespace {

constexpr int RS_THREADS = 256;
constexpr double RS_EPS = 2.220446049250313e-16;

__device__ __forceinline__ uint32_t lowbias32(uint32_t h)
{
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ int sample_index(uint32_t seed, uint32_t it, uint32_t draw, int n)
{
    const uint32_t h = lowbias32(seed ^ (it * 0x9E3779B1u) ^ (draw * 0x85EBCA77u));
    return (int)(((unsigned long long)h * (unsigned long long)n) >> 32);
}

template <int M>
__device__ __forceinline__ bool draw_samples(uint32_t seed, uint32_t it, int n, int* idx)
{
    int have = 0;
    for (int d = 0; d < 4 * M; ++d) {
        const int c = sample_index(seed, it, d, n);
        bool dup = false;
#pragma unroll
        for (int j = 0; j < M; ++j) dup |= (j < have) && idx[j] == c;
        if (have < M && !dup) {
#pragma unroll
            for (int j = 0; j < M; ++j) if (j == have) idx[j] = c;
            ++have;
        }
    }
    return have == M;
}

struct M3 { double m[9]; };

__device__ __forceinline__ M3 adj3(const M3& a)
{
    M3 c;
    c.m[0] = a.m[4] * a.m[8] - a.m[5] * a.m[7];  c.m[1] = a.m[2] * a.m[7] - a.m[1] * a.m[8];  c.m[2] = a.m[1] * a.m[5] - a.m[2] * a.m[4];
    c.m[3] = a.m[5] * a.m[6] - a.m[3] * a.m[8];  c.m[4] = a.m[0] * a.m[8] - a.m[2] * a.m[6];  c.m[5] = a.m[2] * a.m[3] - a.m[0] * a.m[5];
    c.m[6] = a.m[3] * a.m[7] - a.m[4] * a.m[6];  c.m[7] = a.m[1] * a.m[6] - a.m[0] * a.m[7];  c.m[8] = a.m[0] * a.m[4] - a.m[1] * a.m[3];
    return c;
}

__device__ __forceinline__ double sgn(double v) { return (v > 0.0) - (v < 0.0); }

// src / dst: the four sample points (x0 y0 x1 y1 ...).  Returns false for a degenerate sample.
__device__ bool homography_4pt(const double* s, const double* d, double* H)
{
    M3 A{{s[0], s[2], s[4], s[1], s[3], s[5], 1.0, 1.0, 1.0}};      // columns p1 p2 p3
    M3 B{{d[0], d[2], d[4], d[1], d[3], d[5], 1.0, 1.0, 1.0}};
    const M3 aA = adj3(A), aB = adj3(B);
    double lam[3], mu[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        lam[i] = aA.m[3 * i] * s[6] + aA.m[3 * i + 1] * s[7] + aA.m[3 * i + 2];
        mu[i] = aB.m[3 * i] * d[6] + aB.m[3 * i + 1] * d[7] + aB.m[3 * i + 2];
    }
    const double detA = A.m[0] * aA.m[0] + A.m[1] * aA.m[3] + A.m[2] * aA.m[6];
    const double detB = B.m[0] * aB.m[0] + B.m[1] * aB.m[3] + B.m[2] * aB.m[6];
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) { sa = fmax(sa, fabs(s[i])); sb = fmax(sb, fabs(d[i])); }
    sa = sa * sa + 1e-300; sb = sb * sb + 1e-300;
    const double tiny = 1e-9;
    bool ok = fabs(detA) > tiny * sa && fabs(detB) > tiny * sb;
    const double pa = sgn(detA), pb = sgn(detB);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ok = ok && fabs(lam[i]) > tiny * sa && fabs(mu[i]) > tiny * sb;
        ok = ok && (sgn(lam[i]) * pa == sgn(mu[i]) * pb);           // the sample keeps its orientation (checkSubset)
    }
    const double w[3] = {lam[1] * lam[2], lam[0] * lam[2], lam[0] * lam[1]};
    double amax = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) acc += B.m[3 * i + j] * mu[j] * w[j] * aA.m[3 * j + k];
            H[3 * i + k] = acc;
            amax = fmax(amax, fabs(acc));
        }
    ok = ok && fabs(H[8]) > 1e-12 * amax;
    const double inv = 1.0 / (ok ? H[8] : 1.0);
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] *= inv;
    return ok;
}

__device__ __forceinline__ double reproj_err2(const double* H, double x, double y, double u, double v)
{
    const double w = H[6] * x + H[7] * y + H[8];
    const double wi = fabs(w) > RS_EPS ? 1.0 / w : 0.0;
    const double dx = (H[0] * x + H[1] * y + H[2]) * wi - u;
    const double dy = (H[3] * x + H[4] * y + H[5]) * wi - v;
    return dx * dx + dy * dy;
}

__device__ int update_iters(double conf, double outlier_ratio, int m, int max_iters)
{
    const double p = fmin(fmax(conf, 0.0), 1.0), ep = fmin(fmax(outlier_ratio, 0.0), 1.0);
    double num = fmax(1.0 - p, 2.2250738585072014e-308);
    double denom = 1.0 - pow(1.0 - ep, (double)m);
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num); denom = log(denom);
    return (denom >= 0 || -num >= max_iters * (-denom)) ? max_iters : (int)rint(num / denom);
}

// sums N per-thread doubles over the workgroup; every thread gets the totals back in v
template <int N>
__device__ void block_sum_n(double* v, double* scratch /* [RS_THREADS/64 + 1][N] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = v[i];
        for (int o = 32; o; o >>= 1) s += __shfl_down(s, o);
        if (lane == 0) scratch[wave * N + i] = s;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        double s = 0.0;
        for (int w = 0; w < RS_THREADS / 64; ++w) s += scratch[w * N + threadIdx.x];
        scratch[(RS_THREADS / 64) * N + threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = scratch[(RS_THREADS / 64) * N + i];
    __syncthreads();
}

// Solves the n x n system a x = b in place (partial pivoting); returns false when singular.  One thread.
__device__ bool solve_small(double* a /* [n][n+1] augmented */, int n)
{
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = fabs(a[c * (n + 1) + c]);
        for (int r = c + 1; r < n; ++r) if (fabs(a[r * (n + 1) + c]) > best) { best = fabs(a[r * (n + 1) + c]); piv = r; }
        if (!(best > 0.0) || !isfinite(best)) return false;
        if (piv != c) for (int k = 0; k <= n; ++k) { const double t = a[c * (n + 1) + k]; a[c * (n + 1) + k] = a[piv * (n + 1) + k]; a[piv * (n + 1) + k] = t; }
        const double inv = 1.0 / a[c * (n + 1) + c];
        for (int r = c + 1; r < n; ++r) {
            const double f = a[r * (n + 1) + c] * inv;
            if (f != 0.0) for (int k = c; k <= n; ++k) a[r * (n + 1) + k] -= f * a[c * (n + 1) + k];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        double s = a[r * (n + 1) + n];
        for (int k = r + 1; k < n; ++k) s -= a[r * (n + 1) + k] * a[k * (n + 1) + n];
        a[r * (n + 1) + n] = s / a[r * (n + 1) + r];
    }
    return true;
}

struct RansacArgs {
    const float* m0; int cols0; const float* m1; int cols1;
    int max_k; const int32_t* k_dev; const float* scale;      // [B][4] = (sx0, sy0, sx1, sy1): normalised -> pixels
    const uint32_t* seed_dev; uint32_t seed;
    double threshold, confidence; int max_iters, refine;
    double* H; uint8_t* mask; int32_t* info;                  // [B][9], [B][max_k], [B][4] = (found, inliers, iterations, 0)
};

// accumulates the upper triangle of J^T J (36), J^T r (8) for one 2-row block of an 8-parameter problem
__device__ __forceinline__ void acc_rows(double* acc, const double* j0, double r0, const double* j1, double r1)
{
    int t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = i; k < 8; ++k) acc[t++] += j0[i] * j0[k] + j1[i] * j1[k];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[36 + i] += j0[i] * r0 + j1[i] * r1;
}

__device__ __forceinline__ void load_system(double* sys, const double* acc, double damp)
{   // thread 0: expands the packed triangle into the augmented [8][9] system, (1 + damp) on the diagonal
    int t = 0;
    for (int i = 0; i < 8; ++i)
        for (int k = i; k < 8; ++k) { sys[i * 9 + k] = acc[t]; sys[k * 9 + i] = acc[t]; ++t; }
    for (int i = 0; i < 8; ++i) { sys[i * 9 + i] *= (1.0 + damp); sys[i * 9 + 8] = acc[36 + i]; }
}

__global__ __launch_bounds__(RS_THREADS) void ransac_homography(RansacArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* pts = reinterpret_cast<float4*>(smem);                       // [n] (x, y, u, v) pixels
    __shared__ double scratch[(RS_THREADS / 64 + 1) * 45];
    __shared__ double bestH[9], sys[72], hcur[9];
    __shared__ unsigned long long wkey[RS_THREADS / 64];
    __shared__ int s_flag;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.k_dev ? min(a.k_dev[b], a.max_k) : a.max_k;
    const uint32_t seed = a.seed_dev ? a.seed_dev[b] : a.seed;
    const float* sc = a.scale + 4 * b;
    for (int i = tid; i < n; i += RS_THREADS) {
        const float* p0 = a.m0 + ((size_t)b * a.max_k + i) * a.cols0;
        const float* p1 = a.m1 + ((size_t)b * a.max_k + i) * a.cols1;
        pts[i] = make_float4(p0[0] * sc[0], p0[1] * sc[1], p1[0] * sc[2], p1[1] * sc[3]);      // MHA.py:41-42 (fp32 products)
    }
    uint8_t* mask = a.mask + (size_t)b * a.max_k;
    for (int i = tid; i < a.max_k; i += RS_THREADS) mask[i] = 0;
    if (tid == 0) s_flag = 0;
    __syncthreads();
    int32_t* info = a.info + 4 * b;
    double* Hout = a.H + 9 * b;
    if (n < 4) {
        if (tid < 9) Hout[tid] = 0.0;
        if (tid == 0) { info[0] = 0; info[1] = 0; info[2] = 0; info[3] = 0; }
        return;
    }
    const double t2 = a.threshold * a.threshold;
    int done = 0, niters = a.max_iters, best = 0;
    if (n == 4) {                       // the minimal set: the model itself, every point an inlier (OpenCV skips RANSAC)
        if (tid == 0) {
            double s[8], d[8], H[9];
            for (int j = 0; j < 4; ++j) { s[2 * j] = pts[j].x; s[2 * j + 1] = pts[j].y; d[2 * j] = pts[j].z; d[2 * j + 1] = pts[j].w; }
            const bool ok = homography_4pt(s, d, H);
            for (int i = 0; i < 9; ++i) Hout[i] = ok ? H[i] : 0.0;
            for (int i = 0; i < 4; ++i) mask[i] = ok;
            info[0] = ok; info[1] = ok ? 4 : 0; info[2] = 0; info[3] = 0;
        }
        return;
    }
    while (done < niters) {
        const uint32_t it = (uint32_t)(done + tid);
        int idx[4] = {0, 0, 0, 0};
        bool ok = draw_samples<4>(seed, it, n, idx) && (int)it < a.max_iters;
        double s[8], d[8], H[9];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 p = pts[idx[j]]; s[2 * j] = p.x; s[2 * j + 1] = p.y; d[2 * j] = p.z; d[2 * j + 1] = p.w; }
        ok = homography_4pt(s, d, H) && ok;
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            const float4 p = pts[i];
            cnt += reproj_err2(H, p.x, p.y, p.z, p.w) <= t2;
        }
        if (!ok) cnt = 0;
        // workgroup arg-max: more inliers first, then the lower iteration
        unsigned long long key = ((unsigned long long)(unsigned)cnt << 32) | (unsigned long long)(0xFFFFFFFFu - it);
        unsigned long long kmax = key;
        for (int o = 32; o; o >>= 1) { const unsigned long long t = __shfl_down(kmax, o); kmax = t > kmax ? t : kmax; }
        if ((tid & 63) == 0) wkey[tid >> 6] = kmax;
        __syncthreads();
        kmax = wkey[0];
        for (int w = 1; w < RS_THREADS / 64; ++w) kmax = wkey[w] > kmax ? wkey[w] : kmax;
        const int top = (int)(kmax >> 32);
        if (top > max(best, 3)) {
            best = top;
            if (key == kmax) for (int i = 0; i < 9; ++i) bestH[i] = H[i];
        }
        __syncthreads();
        done += RS_THREADS;
        niters = best ? update_iters(a.confidence, (double)(n - best) / n, 4, a.max_iters) : a.max_iters;
    }
    if (best == 0) {
        if (tid < 9) Hout[tid] = 0.0;
        if (tid == 0) { info[0] = 0; info[1] = 0; info[2] = done; info[3] = 0; }
        return;
    }
    // ---- inliers of the best model
    double Hb[9];
    for (int i = 0; i < 9; ++i) Hb[i] = bestH[i];
    for (int i = tid; i < n; i += RS_THREADS) {
        const float4 p = pts[i];
        mask[i] = reproj_err2(Hb, p.x, p.y, p.z, p.w) <= t2;
    }
    __syncthreads();
    if (a.refine) {
        // ---- normalisation (OpenCV runKernel): centroid and mean absolute deviation per axis, inliers only
        double st[5] = {0, 0, 0, 0, 0};
        for (int i = tid; i < n; i += RS_THREADS) if (mask[i]) { const float4 p = pts[i]; st[0] += p.x; st[1] += p.y; st[2] += p.z; st[3] += p.w; st[4] += 1.0; }
        block_sum_n<5>(st, scratch);
        const double cnt = st[4];
        const double c0x = st[0] / cnt, c0y = st[1] / cnt, c1x = st[2] / cnt, c1y = st[3] / cnt;
        double dv[4] = {0, 0, 0, 0};
        for (int i = tid; i < n; i += RS_THREADS) if (mask[i]) {
            const float4 p = pts[i];
            dv[0] += fabs(p.x - c0x); dv[1] += fabs(p.y - c0y); dv[2] += fabs(p.z - c1x); dv[3] += fabs(p.w - c1y);
        }
        block_sum_n<4>(dv, scratch);
        double sn[4];
        for (int i = 0; i < 4; ++i) { const double m = dv[i] / cnt; sn[i] = m > RS_EPS ? 1.0 / m : 1.0; }
        // ---- inhomogeneous DLT in the normalised frame: rows (x y 1 0 0 0 -ux -uy | u), (0 0 0 x y 1 -vx -vy | v)
        double acc[44];
        for (int i = 0; i < 44; ++i) acc[i] = 0.0;
        for (int i = tid; i < n; i += RS_THREADS) if (mask[i]) {
            const float4 p = pts[i];
            const double x = (p.x - c0x) * sn[0], y = (p.y - c0y) * sn[1], u = (p.z - c1x) * sn[2], v = (p.w - c1y) * sn[3];
            const double j0[8] = {x, y, 1.0, 0, 0, 0, -u * x, -u * y}, j1[8] = {0, 0, 0, x, y, 1.0, -v * x, -v * y};
            acc_rows(acc, j0, u, j1, v);
        }
        block_sum_n<44>(acc, scratch);
        if (tid == 0) {
            load_system(sys, acc, 0.0);
            bool ok = solve_small(sys, 8);
            double Hn[9], T[9];
            for (int i = 0; i < 8; ++i) { Hn[i] = sys[i * 9 + 8]; ok = ok && isfinite(Hn[i]); }
            Hn[8] = 1.0;
            // H = Tb^-1 Hn Ta, Ta = [sx 0 -cx sx; 0 sy -cy sy; 0 0 1], Tb^-1 = [1/su 0 cu; 0 1/sv cv; 0 0 1]
            const double ta[9] = {sn[0], 0, -c0x * sn[0], 0, sn[1], -c0y * sn[1], 0, 0, 1.0};
            const double tb[9] = {1.0 / sn[2], 0, c1x, 0, 1.0 / sn[3], c1y, 0, 0, 1.0};
            for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { double s = 0; for (int j = 0; j < 3; ++j) s += Hn[3 * i + j] * ta[3 * j + k]; T[3 * i + k] = s; }
            for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { double s = 0; for (int j = 0; j < 3; ++j) s += tb[3 * i + j] * T[3 * j + k]; hcur[3 * i + k] = s; }
            ok = ok && fabs(hcur[8]) > 1e-300;
            if (ok) { const double inv = 1.0 / hcur[8]; for (int i = 0; i < 9; ++i) hcur[i] *= inv; }
            s_flag = ok;
        }
        __syncthreads();
        // keep whichever of (DLT refit, best sample model) has the smaller inlier error as the LM start
        double e2[2] = {0.0, 0.0};
        double Hd[9];
        for (int i = 0; i < 9; ++i) Hd[i] = hcur[i];
        const bool dlt_ok = s_flag != 0;
        for (int i = tid; i < n; i += RS_THREADS) if (mask[i]) {
            const float4 p = pts[i];
            e2[0] += dlt_ok ? reproj_err2(Hd, p.x, p.y, p.z, p.w) : 0.0;
            e2[1] += reproj_err2(Hb, p.x, p.y, p.z, p.w);
        }
        block_sum_n<2>(e2, scratch);
        double h[8], err;
        if (dlt_ok && e2[0] < e2[1]) { for (int i = 0; i < 8; ++i) h[i] = Hd[i]; err = e2[0]; }
        else { for (int i = 0; i < 8; ++i) h[i] = Hb[i]; err = e2[1]; }
        // ---- Levenberg-Marquardt on h11..h32 (h33 = 1)
        double lam = 1e-3;
        for (int iter = 0; iter < 10; ++iter) {
            for (int i = 0; i < 44; ++i) acc[i] = 0.0;
            for (int i = tid; i < n; i += RS_THREADS) if (mask[i]) {
                const float4 p = pts[i];
                const double x = p.x, y = p.y;
                const double w = h[6] * x + h[7] * y + 1.0;
                const double wi = fabs(w) > RS_EPS ? 1.0 / w : 0.0;
                const double u = (h[0] * x + h[1] * y + h[2]) * wi, v = (h[3] * x + h[4] * y + h[5]) * wi;
                const double j0[8] = {x * wi, y * wi, wi, 0, 0, 0, -x * wi * u, -y * wi * u};
                const double j1[8] = {0, 0, 0, x * wi, y * wi, wi, -x * wi * v, -y * wi * v};
                acc_rows(acc, j0, u - (double)p.z, j1, v - (double)p.w);
            }
            block_sum_n<44>(acc, scratch);
            bool improved = false;
            for (int attempt = 0; attempt < 6 && !improved; ++attempt) {
                if (tid == 0) {
                    load_system(sys, acc, lam);
                    for (int i = 0; i < 8; ++i) sys[i * 9 + 8] = -sys[i * 9 + 8];
                    s_flag = solve_small(sys, 8);
                }
                __syncthreads();
                const bool solved = s_flag != 0;
                double hn[8];
                for (int i = 0; i < 8; ++i) hn[i] = h[i] + sys[i * 9 + 8];
                __syncthreads();
                if (!solved) { lam *= 10.0; continue; }
                double en[1] = {0.0};
                double Ht[9] = {hn[0], hn[1], hn[2], hn[3], hn[4], hn[5], hn[6], hn[7], 1.0};
                for (int i = tid; i < n; i += RS_THREADS) if (mask[i]) { const float4 p = pts[i]; en[0] += reproj_err2(Ht, p.x, p.y, p.z, p.w); }
                block_sum_n<1>(en, scratch);
                if (en[0] < err) { for (int i = 0; i < 8; ++i) h[i] = hn[i]; err = en[0]; lam /= 10.0; improved = true; }
                else lam *= 10.0;
            }
            if (!improved) break;
        }
        if (tid == 0) { for (int i = 0; i < 8; ++i) Hout[i] = h[i]; Hout[8] = 1.0; }
    } else if (tid < 9) {
        Hout[tid] = Hb[tid];
    }
    if (tid == 0) { info[0] = 1; info[1] = best; info[2] = done; info[3] = 0; }
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int kpb_find_homography(
    kpb_ctx* ctx, const float* m0_dev, int cols0, const float* m1_dev, int cols1, int batch, int max_k, const int32_t* k_dev,
    const float* scale_dev, const uint32_t* seed_dev, uint32_t seed, const kpb_ransac_params* prm, double* out_h_dev, uint8_t* out_mask_dev,
    int32_t* out_info_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_find_homography: null context");
    if (batch <= 0 || max_k < 0 || cols0 < 2 || cols1 < 2 || !scale_dev || !prm || !out_h_dev || !out_info_dev || (max_k && (!m0_dev || !m1_dev || !out_mask_dev)))
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_find_homography: bad argument");
    if ((size_t)max_k * sizeof(float4) > 128 * 1024)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_find_homography: at most %d matches per pair", (int)(128 * 1024 / sizeof(float4)));
    if (!(prm->threshold > 0.0) || prm->max_iters < 1) return kpb_fail(ctx, KPB_E_INVALID, "kpb_find_homography: bad parameters");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    RansacArgs a{m0_dev, cols0, m1_dev, cols1, max_k, k_dev, scale_dev, seed_dev, seed, prm->threshold, prm->confidence, prm->max_iters, prm->refine,
                 out_h_dev, out_mask_dev, out_info_dev};
    const size_t lds = (size_t)max_k * sizeof(float4);
    static bool attr_set = false;
    if (!attr_set) {
        KPB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ransac_homography), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    KPB_LAUNCH(ctx, "ransac_homography", ransac_homography, dim3(batch), dim3(RS_THREADS), lds, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}
