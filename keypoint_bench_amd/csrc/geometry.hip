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
//   * a round = 256 hypotheses, one per thread: the subsets come from OpenCV's own generator (cv::RNG + getSubset + checkSubset,
//     produced sequentially by one thread: see CvRng), the exact model in closed form (projective basis: two 3x3 adjugates,
//     registers only), then every thread scores ALL matches against its own model (LDS broadcast reads);
//   * OpenCV's keep / niters rule (more inliers than any earlier model; RANSACUpdateNumIters after every kept model) is applied
//     to the round's results in iteration order, so the hypothesis stream, the kept model and the iteration at which the loop
//     ends are those of OpenCV's sequential loop;
//   * refit on the inliers: normalised inhomogeneous DLT (8x8 normal equations, 44 block-reduced sums) and up to ten
//     Levenberg-Marquardt steps on the forward reprojection error.  All estimator arithmetic is fp64.
namespace {

constexpr int RS_THREADS = 256;
constexpr double RS_EPS = 2.220446049250313e-16;

// ---- OpenCV's sampler (r03).  cv::RNG is a multiply-with-carry generator on a 64-bit state; RANSACPointSetRegistrator::run seeds
// it with (uint64)-1 at EVERY call (seed 0 here), draws each index of a subset until it differs from the ones already picked,
// and redraws the whole subset -- up to 10 000 times, without spending an iteration -- until checkSubset accepts it
// (modules/calib3d/src/ptsetreg.cpp, OpenCV 4.9: requirements.txt:2).  The stream is sequential, so ONE thread produces the
// subsets of a round (256 hypotheses) and every thread then evaluates one of them; the keep / niters rule is applied to the
// round's results in iteration order (scan_round), which gives what OpenCV's sequential loop gives.
struct CvRng {
    unsigned long long s;
    __device__ __forceinline__ unsigned next() { s = (unsigned long long)(unsigned)s * 4164903690ull + (s >> 32); return (unsigned)s; }
    __device__ __forceinline__ int uniform(int n) { return (int)(next() % (unsigned)n); }      // rng.uniform(0, n), n > 0
};

__device__ __forceinline__ unsigned long long rng_state(uint32_t seed)
{
    return seed == 0u ? ~0ull : (((unsigned long long)(seed ^ 0x9E3779B9u) << 32) | (unsigned long long)seed);
}

constexpr int GETSUBSET_ATTEMPTS = 10000;
constexpr int FM_LMEDS_BELOW = 15;        // cv::findFundamentalMat: FM_RANSAC with fewer points runs the least-median registrator

// thread 0 only.  s_idx[j][0..M) = the subset of hypothesis j of this round; s_ok[j] = 0 from the first getSubset failure on.
template <int M, class Check>
__device__ void gen_round(CvRng& rng, int n, unsigned short* s_idx, unsigned char* s_ok, bool& alive, Check check, int attempts = GETSUBSET_ATTEMPTS)
{
    for (int j = 0; j < 256; ++j) {
        int idx[M];
#pragma unroll
        for (int i = 0; i < M; ++i) idx[i] = 0;
        bool found = false;
        if (alive) {
            for (int att = 0; att < attempts && !found; ++att) {
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    int c;
                    bool dup;
                    do {
                        c = rng.uniform(n);
                        dup = false;
#pragma unroll
                        for (int k = 0; k < M; ++k) dup |= (k < i) && idx[k] == c;
                    } while (dup);
                    idx[i] = c;
                }
                found = check(idx);
            }
        }
        alive = alive && found;
        s_ok[j] = alive ? 1 : 0;
#pragma unroll
        for (int i = 0; i < M; ++i) s_idx[j * M + i] = (unsigned short)(alive ? idx[i] : 0);
    }
}

struct RsState { int niters, max_good, iters, stop, failed; };

__device__ int update_iters(double conf, double outlier_ratio, int m, int max_iters);

// thread 0 only: RANSACPointSetRegistrator::run's keep / niters rule over one round, in iteration order.  s_cnt[j][s] = inliers
// of model s of hypothesis j (0 for a void model).  Returns the (hypothesis, slot) kept LAST in this round, or -1.
template <int M, int NM>
__device__ int scan_round(const int* s_cnt, const unsigned char* s_ok, int base, int n, double conf, RsState& st)
{
    int best = -1;
    for (int j = 0; j < 256; ++j) {
        const int it = base + j;
        if (it >= st.niters) { st.stop = 1; break; }
        if (!s_ok[j]) { st.failed = it == 0; st.stop = 1; break; }        // `if (iter == 0) return false; break;`
        for (int sl = 0; sl < NM; ++sl) {
            const int c = s_cnt[j * NM + sl];
            if (c > max(st.max_good, M - 1)) {
                best = j * NM + sl;
                st.max_good = c;
                st.niters = update_iters(conf, (double)(n - c) / n, M, st.niters);
            }
        }
        st.iters = it + 1;
    }
    return best;
}

// OpenCV's haveCollinearPoints(m, count): the LAST of `count` points against every pair of earlier ones (FLT_EPSILON test)
template <int M>
__device__ __forceinline__ bool last_collinear(const double* x /* M x (u, v) */)
{
    bool bad = false;
#pragma unroll
    for (int j = 0; j < M - 1; ++j) {
        const double dx1 = x[2 * j] - x[2 * (M - 1)], dy1 = x[2 * j + 1] - x[2 * (M - 1) + 1];
#pragma unroll
        for (int k = 0; k < M - 1; ++k) {
            const double dx2 = x[2 * k] - x[2 * (M - 1)], dy2 = x[2 * k + 1] - x[2 * (M - 1) + 1];
            if (k < j) bad = bad || fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2));
        }
    }
    return bad;
}

__device__ __forceinline__ double det3x3(const double* m)      // cv::determinant(Matx33d)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// HomographyEstimatorCallback::checkSubset on a 4-point sample (s, d: x0 y0 x1 y1 ...)
__device__ bool check_subset_h(const double* s, const double* d)
{
    if (last_collinear<4>(s) || last_collinear<4>(d)) return false;
    const int tt[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 2, 3}, {0, 1, 3}};
    int negative = 0;
    for (int i = 0; i < 4; ++i) {
        const int* t = tt[i];
        const double A[9] = {s[2 * t[0]], s[2 * t[0] + 1], 1.0, s[2 * t[1]], s[2 * t[1] + 1], 1.0, s[2 * t[2]], s[2 * t[2] + 1], 1.0};
        const double B[9] = {d[2 * t[0]], d[2 * t[0] + 1], 1.0, d[2 * t[1]], d[2 * t[1] + 1], 1.0, d[2 * t[2]], d[2 * t[2] + 1], 1.0};
        negative += det3x3(A) * det3x3(B) < 0;
    }
    return negative == 0 || negative == 4;
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
    __shared__ unsigned short s_idx[RS_THREADS * 4];
    __shared__ unsigned char s_ok[RS_THREADS];
    __shared__ int s_cnt[RS_THREADS];
    __shared__ int s_ctl[4];            // niters, stop, kept (hypothesis of this round, -1: none), failed
    CvRng rng{rng_state(seed)};
    RsState st{max(a.max_iters, 1), 0, 0, 0, 0};
    bool alive = true;
    int base = 0;
    bool have = false;
    for (;;) {
        if (tid == 0)
            gen_round<4>(rng, n, s_idx, s_ok, alive, [&](const int* idx) {
                double s[8], d[8];
                for (int j = 0; j < 4; ++j) { const float4 p = pts[idx[j]]; s[2 * j] = p.x; s[2 * j + 1] = p.y; d[2 * j] = p.z; d[2 * j + 1] = p.w; }
                return check_subset_h(s, d);
            });
        __syncthreads();
        double s[8], d[8], H[9];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 p = pts[s_idx[tid * 4 + j]]; s[2 * j] = p.x; s[2 * j + 1] = p.y; d[2 * j] = p.z; d[2 * j + 1] = p.w; }
        const bool ok = homography_4pt(s, d, H) && s_ok[tid];
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            const float4 p = pts[i];
            cnt += reproj_err2(H, p.x, p.y, p.z, p.w) <= t2;
        }
        s_cnt[tid] = ok ? cnt : 0;
        __syncthreads();
        if (tid == 0) {
            const int kept = scan_round<4, 1>(s_cnt, s_ok, base, n, a.confidence, st);
            s_ctl[0] = st.niters; s_ctl[1] = st.stop; s_ctl[2] = kept; s_ctl[3] = st.failed;
        }
        __syncthreads();
        if (s_ctl[2] == tid) for (int i = 0; i < 9; ++i) bestH[i] = H[i];
        have = have || s_ctl[2] >= 0;
        base += RS_THREADS;
        const bool more = !s_ctl[1] && base < s_ctl[0];
        __syncthreads();
        if (!more) break;
    }
    if (!have) {
        if (tid < 9) Hout[tid] = 0.0;
        if (tid == 0) { info[0] = 0; info[1] = 0; info[2] = st.iters; info[3] = 0; }
        return;
    }
    const int best_inliers = st.max_good;
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
    if (tid == 0) { info[0] = 1; info[1] = best_inliers; info[2] = st.iters; info[3] = 0; }
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
        return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_find_homography: at most %d matches per pair", (int)(128 * 1024 / sizeof(float4)));
    if (!(prm->threshold > 0.0) || prm->max_iters < 1) return kpb_fail(ctx, KPB_E_INVALID, "kpb_find_homography: bad parameters");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    RansacArgs a{m0_dev, cols0, m1_dev, cols1, max_k, k_dev, scale_dev, seed_dev, seed, prm->threshold, prm->confidence, prm->max_iters, prm->refine,
                 out_h_dev, out_mask_dev, out_info_dev};
    const size_t lds = (size_t)max_k * sizeof(float4);
    if (!(ctx->lds_attr_done & KPB_ATTR_HOMOGRAPHY)) {
        KPB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ransac_homography), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ctx->lds_attr_done |= KPB_ATTR_HOMOGRAPHY;
    }
    KPB_LAUNCH(ctx, "ransac_homography", ransac_homography, dim3(batch), dim3(RS_THREADS), lds, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}

// ================================================================================================ RANSAC essential matrix
// cv2.findEssentialMat(k0, k1, eye(3), threshold, prob, RANSAC) + cv2.recoverPose as tasks/AUC.py:50-64 calls them (maxIters
// 1000, five-point samples, Sampson error, no refinement of the winning model), restated from the published algorithm --
// PARITY UNPINNED, numpy restatement in oracle/geometry_ref.py (same sampler, same elimination order, same root finder).
// One 256-thread workgroup per pair, one hypothesis per thread and round; a hypothesis is Nister's five-point solver:
// null space of the 5 x 9 constraints by Gauss-Jordan, the ten cubic constraints expanded by polynomial arithmetic
// (tables below), Gauss-Jordan on the ten leading monomials, det B(z) (degree 10), its roots by Aberth-Ehrlich iteration,
// up to ten candidate matrices, each scored against ALL matches (LDS broadcast reads).  The solver's small matrices are
// thread-private (scratch): this stage is a few milliseconds per 256 pairs next to a 30 ms extraction step.
namespace {

// products of the monomial bases: linear (x, y, z, 1) x linear -> quadratic DEG2; quadratic x linear -> MONO (Nister's order:
// x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1)
__constant__ int8_t E_MUL11[4][4] = {{0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9}};
__constant__ int8_t E_MUL21[10][4] = {{0, 2, 4, 5},   {2, 3, 8, 9},   {4, 8, 10, 11}, {3, 1, 6, 7},   {8, 6, 13, 14},
                                      {10, 13, 16, 17}, {5, 9, 11, 12}, {9, 7, 14, 15}, {11, 14, 17, 18}, {12, 15, 18, 19}};

__device__ void pmul11_acc(double* out, const double* a, const double* b, double s)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[E_MUL11[i][j]] += s * a[i] * b[j];
}

__device__ void pmul21_acc(double* out, const double* a, const double* b, double s)
{
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 4; ++j) out[E_MUL21[i][j]] += s * a[i] * b[j];
}

// Reduces the first ncol columns of the r x c matrix to the identity (row pivoting, first maximum).  false: singular.
__device__ bool gauss_jordan(double* M, int r, int c, int ncol)
{
    for (int col = 0; col < ncol; ++col) {
        int piv = col;
        double best = fabs(M[col * c + col]);
        for (int rr = col + 1; rr < r; ++rr) if (fabs(M[rr * c + col]) > best) { best = fabs(M[rr * c + col]); piv = rr; }
        if (piv != col) for (int k = 0; k < c; ++k) { const double t = M[col * c + k]; M[col * c + k] = M[piv * c + k]; M[piv * c + k] = t; }
        const double d = M[col * c + col];
        if (!(fabs(d) > 1e-300)) return false;
        for (int k = 0; k < c; ++k) M[col * c + k] = M[col * c + k] / d;
        for (int rr = 0; rr < r; ++rr) {
            if (rr == col) continue;
            const double f = M[rr * c + col];
            for (int k = 0; k < c; ++k) M[rr * c + k] = M[rr * c + k] - f * M[col * c + k];
        }
    }
    return true;
}

__device__ void polymul_acc(double* out, const double* a, int na, const double* b, int nb, double s)
{
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[i + j] += s * a[i] * b[j];
}

__device__ __forceinline__ double polyval_r(const double* c, int n, double z)
{
    double v = c[n - 1];
    for (int i = n - 2; i >= 0; --i) v = v * z + c[i];
    return v;
}

struct Cx { double re, im; };
__device__ __forceinline__ Cx cmul(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Cx cdiv(Cx a, Cx b)
{
    const double d = b.re * b.re + b.im * b.im;
    return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}

// all roots of the degree-10 polynomial cn (ascending, cn[10] = 1) by Aberth-Ehrlich from a fixed start circle
__device__ void aberth10(const double* cn, Cx* z)
{
    double rad = pow(fabs(cn[0]), 0.1);
    if (!(rad > 0.0) || !isfinite(rad)) rad = 1.0;
    rad = fmin(fmax(rad, 1e-3), 1e3);
    for (int k = 0; k < 10; ++k) { const double a = 2.0 * 3.141592653589793 * k / 10.0 + 0.4; z[k] = {rad * cos(a), rad * sin(a)}; }
    double dc[10];
    for (int i = 0; i < 10; ++i) dc[i] = cn[i + 1] * (double)(i + 1);
    for (int it = 0; it < 40; ++it) {
        Cx zn[10];
        for (int k = 0; k < 10; ++k) {
            Cx p = {cn[10], 0.0}, dp = {dc[9], 0.0};
            for (int i = 9; i >= 0; --i) { p = cmul(p, z[k]); p.re += cn[i]; }
            for (int i = 8; i >= 0; --i) { dp = cmul(dp, z[k]); dp.re += dc[i]; }
            if (dp.re == 0.0 && dp.im == 0.0) dp.re = 1e-300;
            const Cx ratio = cdiv(p, dp);
            Cx s = {0.0, 0.0};
            for (int j = 0; j < 10; ++j) {
                if (j == k) continue;
                Cx d = {z[k].re - z[j].re, z[k].im - z[j].im};
                if (d.re == 0.0 && d.im == 0.0) d.re = 1e-300;
                const Cx inv = cdiv({1.0, 0.0}, d);
                s.re += inv.re; s.im += inv.im;
            }
            const Cx rs = cmul(ratio, s);
            const Cx upd = cdiv(ratio, {1.0 - rs.re, -rs.im});
            zn[k] = {z[k].re - upd.re, z[k].im - upd.im};
        }
        for (int k = 0; k < 10; ++k) z[k] = zn[k];      // Jacobi-style update: every root moves on the previous iterate
    }
}

// Five-point solver.  x1 / x2: the sample's five normalised points (u, v interleaved).  E: up to ten candidates (row-major,
// Frobenius norm 1), valid[s] says which slots hold one.  Returns the number of valid slots.
__device__ int essential_5pt(const double* x1, const double* x2, double (*E)[9], bool* valid)
{
    for (int s = 0; s < 10; ++s) valid[s] = false;
    double Q[5 * 9];
    for (int i = 0; i < 5; ++i) {
        const double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        double* q = Q + 9 * i;
        q[0] = u2 * u1; q[1] = u2 * v1; q[2] = u2; q[3] = v2 * u1; q[4] = v2 * v1; q[5] = v2; q[6] = u1; q[7] = v1; q[8] = 1.0;
    }
    bool ok = gauss_jordan(Q, 5, 9, 5);
    // basis row j = (-C[:, j], e_j); entry e of E as a linear polynomial over (x, y, z, 1): Ep[e][k] = basis[k][e]
    double Ep[9][4];
    for (int e = 0; e < 9; ++e)
        for (int k = 0; k < 4; ++k) Ep[e][k] = e < 5 ? -Q[e * 9 + 5 + k] : (e - 5 == k ? 1.0 : 0.0);
    double EEt[9][10];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double* o = EEt[3 * i + j];
            for (int m = 0; m < 10; ++m) o[m] = 0.0;
            for (int k = 0; k < 3; ++k) pmul11_acc(o, Ep[3 * i + k], Ep[3 * j + k], 1.0);
        }
    double tr[10];
    for (int m = 0; m < 10; ++m) tr[m] = EEt[0][m] + EEt[4][m] + EEt[8][m];
    double M[10 * 20];
    for (int m = 0; m < 200; ++m) M[m] = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double* row = M + 20 * (3 * i + j);
            for (int k = 0; k < 3; ++k) pmul21_acc(row, EEt[3 * i + k], Ep[3 * k + j], 2.0);
            pmul21_acc(row, tr, Ep[3 * i + j], -1.0);
        }
    {
        double m01[10], m02[10], m03[10];
        for (int m = 0; m < 10; ++m) { m01[m] = 0.0; m02[m] = 0.0; m03[m] = 0.0; }
        pmul11_acc(m01, Ep[4], Ep[8], 1.0); pmul11_acc(m01, Ep[5], Ep[7], -1.0);
        pmul11_acc(m02, Ep[5], Ep[6], 1.0); pmul11_acc(m02, Ep[3], Ep[8], -1.0);
        pmul11_acc(m03, Ep[3], Ep[7], 1.0); pmul11_acc(m03, Ep[4], Ep[6], -1.0);
        double* row = M + 20 * 9;
        pmul21_acc(row, m01, Ep[0], 1.0); pmul21_acc(row, m02, Ep[1], 1.0); pmul21_acc(row, m03, Ep[2], 1.0);
    }
    ok = gauss_jordan(M, 10, 20, 10) && ok;
    // rows <a> - z <b> over the trailing monomials [xz2 xz x yz2 yz y z3 z2 z 1]: (x, y, 1) parts as polynomials in z, ascending
    double Bx[3][4], By[3][4], B1[3][5];
    for (int r = 0; r < 3; ++r) {
        const double* a = M + 20 * (4 + 2 * r) + 10;
        const double* b = M + 20 * (5 + 2 * r) + 10;
        Bx[r][0] = a[2]; Bx[r][1] = a[1] - b[2]; Bx[r][2] = a[0] - b[1]; Bx[r][3] = -b[0];
        By[r][0] = a[5]; By[r][1] = a[4] - b[5]; By[r][2] = a[3] - b[4]; By[r][3] = -b[3];
        B1[r][0] = a[9]; B1[r][1] = a[8] - b[9]; B1[r][2] = a[7] - b[8]; B1[r][3] = a[6] - b[7]; B1[r][4] = -b[6];
    }
    double det[11];
    for (int i = 0; i < 11; ++i) det[i] = 0.0;
    {
        double t7[7];
        // (Bx0 By1 - By0 Bx1) B1_2
        for (int i = 0; i < 7; ++i) t7[i] = 0.0;
        polymul_acc(t7, Bx[0], 4, By[1], 4, 1.0); polymul_acc(t7, By[0], 4, Bx[1], 4, -1.0);
        polymul_acc(det, t7, 7, B1[2], 5, 1.0);
        // (By0 B1_1 - B1_0 By1) Bx2   (8 coefficients, times 4)
        double t8[8];
        for (int i = 0; i < 8; ++i) t8[i] = 0.0;
        polymul_acc(t8, By[0], 4, B1[1], 5, 1.0); polymul_acc(t8, B1[0], 5, By[1], 4, -1.0);
        polymul_acc(det, t8, 8, Bx[2], 4, 1.0);
        // (B1_0 Bx1 - Bx0 B1_1) By2
        for (int i = 0; i < 8; ++i) t8[i] = 0.0;
        polymul_acc(t8, B1[0], 5, Bx[1], 4, 1.0); polymul_acc(t8, Bx[0], 4, B1[1], 5, -1.0);
        polymul_acc(det, t8, 8, By[2], 4, 1.0);
    }
    bool fin = true;
    for (int i = 0; i < 11; ++i) fin = fin && isfinite(det[i]);
    ok = ok && fin && fabs(det[10]) > 1e-300;
    if (!ok) return 0;
    double cn[11], dc[10];
    for (int i = 0; i < 11; ++i) cn[i] = det[i] / det[10];
    for (int i = 0; i < 10; ++i) dc[i] = cn[i + 1] * (double)(i + 1);
    Cx roots[10];
    aberth10(cn, roots);
    int nvalid = 0;
    for (int s = 0; s < 10; ++s) {
        double zr = roots[s].re;
        for (int it = 0; it < 3; ++it) {                // Newton polish on the real axis
            const double p = polyval_r(cn, 11, zr);
            double dp = polyval_r(dc, 10, zr);
            if (dp == 0.0) dp = 1e-300;
            zr = zr - p / dp;
        }
        bool real = fabs(roots[s].im) < 1e-6 * (1.0 + fabs(roots[s].re)) && isfinite(zr);
        {
            double pabs = fabs(cn[10]);
            const double az = fabs(zr);
            for (int i = 9; i >= 0; --i) pabs = pabs * az + fabs(cn[i]);
            real = real && fabs(polyval_r(cn, 11, zr)) < 1e-6 * (1.0 + pabs);
        }
        const double b00 = polyval_r(Bx[0], 4, zr), b01 = polyval_r(By[0], 4, zr), b02 = polyval_r(B1[0], 5, zr);
        const double b10 = polyval_r(Bx[1], 4, zr), b11 = polyval_r(By[1], 4, zr), b12 = polyval_r(B1[1], 5, zr);
        const double cx = b01 * b12 - b02 * b11, cy = b02 * b10 - b00 * b12;
        double cw = b00 * b11 - b01 * b10;
        real = real && fabs(cw) > 1e-300;
        if (!(fabs(cw) > 1e-300)) cw = 1.0;
        const double cf[4] = {cx / cw, cy / cw, zr, 1.0};
        double nrm = 0.0;
        for (int e = 0; e < 9; ++e) {
            double v = 0.0;
            for (int k = 0; k < 4; ++k) v += cf[k] * Ep[e][k];
            E[s][e] = v;
            nrm += v * v;
        }
        nrm = sqrt(nrm);
        real = real && isfinite(nrm) && nrm > 0.0;
        if (real) { for (int e = 0; e < 9; ++e) E[s][e] = E[s][e] / nrm; ++nvalid; }
        valid[s] = real;
    }
    return nvalid;
}

// OpenCV's EMEstimatorCallback::computeError (Sampson distance) of one correspondence
__device__ __forceinline__ double sampson_err(const double* E, double u1, double v1, double u2, double v2)
{
    const double a0 = E[0] * u1 + E[1] * v1 + E[2], a1 = E[3] * u1 + E[4] * v1 + E[5], a2 = E[6] * u1 + E[7] * v1 + E[8];       // E x1
    const double b0 = E[0] * u2 + E[3] * v2 + E[6], b1 = E[1] * u2 + E[4] * v2 + E[7];                                           // E^T x2
    const double x2tEx1 = a0 * u2 + a1 * v2 + a2;
    double den = a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1;
    if (den == 0.0) den = 1e-300;
    return x2tEx1 * x2tEx1 / den;
}

struct EssArgs {
    const float* m0; int cols0; const float* m1; int cols1;
    int max_k; const int32_t* k_dev;
    const float* scale;       // [B][4]: normalised (x, y) -> pixels, fp32 as AUC.py:125-126
    const double* cam;        // [B][8]: cx0 cy0 fx0 fy0 cx1 cy1 fx1 fy1 (AUC.py:47-48: (k - c) / f)
    int cam_f32;              // the intrinsics are float32 (datasets/megadepth.py:341-342): numpy then normalises in float32
    const double* thr;        // [B]: norm_thresh = thresh / f_mean
    const uint32_t* seed_dev; uint32_t seed;
    double prob; int max_iters;
    double* E; uint8_t* mask; int32_t* info;      // [B][9], [B][max_k], [B][4] = (found, inliers, hypotheses, 0)
    double* pts;                                   // [B][max_k][4] normalised coordinates (u1 v1 u2 v2), kept for kpb_recover_pose
};

__global__ __launch_bounds__(RS_THREADS) void ransac_essential(EssArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* P = reinterpret_cast<double*>(smem);                     // [n][4]
    __shared__ double bestE[9];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.k_dev ? min(a.k_dev[b], a.max_k) : a.max_k;
    const uint32_t seed = a.seed_dev ? a.seed_dev[b] : a.seed;
    const float* sc = a.scale + 4 * b;
    const double* cam = a.cam + 8 * b;
    double* gp = a.pts + (size_t)b * a.max_k * 4;
    for (int i = tid; i < n; i += RS_THREADS) {
        const float* p0 = a.m0 + ((size_t)b * a.max_k + i) * a.cols0;
        const float* p1 = a.m1 + ((size_t)b * a.max_k + i) * a.cols1;
        const float x0 = p0[0] * sc[0], y0 = p0[1] * sc[1], x1 = p1[0] * sc[2], y1 = p1[1] * sc[3];      // fp32 pixel products (AUC.py:125-126)
        double u1, v1, u2, v2;
        if (a.cam_f32) {
            u1 = (double)__fdiv_rn(x0 - (float)cam[0], (float)cam[2]); v1 = (double)__fdiv_rn(y0 - (float)cam[1], (float)cam[3]);
            u2 = (double)__fdiv_rn(x1 - (float)cam[4], (float)cam[6]); v2 = (double)__fdiv_rn(y1 - (float)cam[5], (float)cam[7]);
        } else {
            u1 = ((double)x0 - cam[0]) / cam[2]; v1 = ((double)y0 - cam[1]) / cam[3];
            u2 = ((double)x1 - cam[4]) / cam[6]; v2 = ((double)y1 - cam[5]) / cam[7];
        }
        P[4 * i] = u1; P[4 * i + 1] = v1; P[4 * i + 2] = u2; P[4 * i + 3] = v2;
        gp[4 * i] = u1; gp[4 * i + 1] = v1; gp[4 * i + 2] = u2; gp[4 * i + 3] = v2;
    }
    uint8_t* mask = a.mask + (size_t)b * a.max_k;
    for (int i = tid; i < a.max_k; i += RS_THREADS) mask[i] = 0;
    __syncthreads();
    int32_t* info = a.info + 4 * b;
    double* Eout = a.E + 9 * b;
    if (n < 5) {
        if (tid < 9) Eout[tid] = 0.0;
        if (tid == 0) { info[0] = 0; info[1] = 0; info[2] = 0; info[3] = 0; }
        return;
    }
    const double t2 = a.thr[b] * a.thr[b];
    __shared__ unsigned short s_idx[RS_THREADS * 5];
    __shared__ unsigned char s_ok[RS_THREADS];
    __shared__ int s_cnt[RS_THREADS * 10];
    __shared__ int s_ctl[4];            // niters, stop, kept (hypothesis * 10 + root slot of this round, -1: none), failed
    CvRng rng{rng_state(seed)};
    RsState st{max(a.max_iters, 1), 0, 0, 0, 0};
    bool alive = true, have = false;
    int base = 0;
    for (;;) {
        if (tid == 0) {
            if (n == 5) {       // `count == modelPoints`: the one sample, no loop (EMEstimatorCallback has no checkSubset)
                for (int j = 0; j < RS_THREADS; ++j) { s_ok[j] = j == 0; for (int i = 0; i < 5; ++i) s_idx[j * 5 + i] = (unsigned short)i; }
                st.niters = 1;
            } else
                gen_round<5>(rng, n, s_idx, s_ok, alive, [](const int*) { return true; });
        }
        __syncthreads();
        double x1[10], x2[10];
        for (int j = 0; j < 5; ++j) { const int q = s_idx[tid * 5 + j]; x1[2 * j] = P[4 * q]; x1[2 * j + 1] = P[4 * q + 1]; x2[2 * j] = P[4 * q + 2]; x2[2 * j + 1] = P[4 * q + 3]; }
        double E[10][9];
        bool valid[10];
        int nv = 0;
        if (s_ok[tid]) nv = essential_5pt(x1, x2, E, valid);
        for (int s = 0; s < 10; ++s) {
            int cnt = 0;
            if (nv && valid[s])
                for (int i = 0; i < n; ++i) cnt += sampson_err(E[s], P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]) <= t2;
            s_cnt[tid * 10 + s] = cnt;
        }
        __syncthreads();
        if (tid == 0) {
            const int kept = scan_round<5, 10>(s_cnt, s_ok, base, n, a.prob, st);
            s_ctl[0] = st.niters; s_ctl[1] = st.stop; s_ctl[2] = kept; s_ctl[3] = st.failed;
        }
        __syncthreads();
        if (s_ctl[2] >= 0 && s_ctl[2] / 10 == tid) for (int e = 0; e < 9; ++e) bestE[e] = E[s_ctl[2] % 10][e];
        have = have || s_ctl[2] >= 0;
        base += RS_THREADS;
        const bool more = !s_ctl[1] && base < s_ctl[0];
        __syncthreads();
        if (!more) break;
    }
    if (!have) {
        if (tid < 9) Eout[tid] = 0.0;
        if (tid == 0) { info[0] = 0; info[1] = 0; info[2] = st.iters; info[3] = 0; }
        return;
    }
    double Eb[9];
    for (int e = 0; e < 9; ++e) Eb[e] = bestE[e];
    for (int i = tid; i < n; i += RS_THREADS) mask[i] = sampson_err(Eb, P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]) <= t2;
    if (tid < 9) Eout[tid] = Eb[tid];
    if (tid == 0) { info[0] = 1; info[1] = st.max_good; info[2] = st.iters; info[3] = 0; }
}

// ---- recoverPose: SVD of E (Jacobi on E^T E), the four (R, t), depth signs of the masked points, the best combination
__device__ void jacobi3(double* A /* symmetric 3x3, destroyed */, double* V)
{
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[3 * p + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[3 * k + p], akq = A[3 * k + q];
                    A[3 * k + p] = c * akp - s * akq; A[3 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[3 * p + k], aqk = A[3 * q + k];
                    A[3 * p + k] = c * apk - s * aqk; A[3 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[3 * k + p], vkq = V[3 * k + q];
                    V[3 * k + p] = c * vkp - s * vkq; V[3 * k + q] = s * vkp + c * vkq;
                }
            }
    }
}

__device__ __forceinline__ double det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

struct PoseArgs {
    const double* E; const double* pts; const uint8_t* mask_in; int max_k; const int32_t* k_dev; const int32_t* info;
    double dist; double* Rt; uint8_t* mask_out; int32_t* good;       // Rt [B][12] = R row-major, t; good [B]
};

__global__ __launch_bounds__(RS_THREADS) void recover_pose(PoseArgs a)
{
    __shared__ double Rs[2][9], ts[3];
    __shared__ double scratch[(RS_THREADS / 64 + 1) * 5];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.k_dev ? min(a.k_dev[b], a.max_k) : a.max_k;
    const uint8_t* min_ = a.mask_in + (size_t)b * a.max_k;
    uint8_t* mout = a.mask_out + (size_t)b * a.max_k;
    for (int i = tid; i < a.max_k; i += RS_THREADS) mout[i] = 0;
    if (a.info[4 * b] == 0) {
        if (tid < 12) a.Rt[12 * b + tid] = 0.0;
        if (tid == 0) a.good[b] = 0;
        return;
    }
    if (tid == 0) {
        const double* E = a.E + 9 * b;
        double AtA[9], V[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += E[3 * k + i] * E[3 * k + j]; AtA[3 * i + j] = s; }
        jacobi3(AtA, V);
        // order the eigenvalues descending: singular values s0 >= s1 >= s2 (~0)
        int o[3] = {0, 1, 2};
        for (int i = 0; i < 2; ++i) for (int j = i + 1; j < 3; ++j) if (AtA[4 * o[j]] > AtA[4 * o[i]]) { const int t = o[i]; o[i] = o[j]; o[j] = t; }
        double Vs[9], U[9];
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Vs[3 * r + c] = V[3 * r + o[c]];
        for (int c = 0; c < 2; ++c) {
            double u[3], nn = 0.0;
            for (int r = 0; r < 3; ++r) { u[r] = E[3 * r] * Vs[c] + E[3 * r + 1] * Vs[3 + c] + E[3 * r + 2] * Vs[6 + c]; nn += u[r] * u[r]; }
            nn = sqrt(nn);
            for (int r = 0; r < 3; ++r) U[3 * r + c] = u[r] / (nn > 0 ? nn : 1.0);
        }
        U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];      // u2 = u0 x u1
        double Vt[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Vt[3 * r + c] = Vs[3 * c + r];
        if (det3(U) < 0) for (int i = 0; i < 9; ++i) U[i] = -U[i];
        if (det3(Vt) < 0) for (int i = 0; i < 9; ++i) Vt[i] = -Vt[i];
        const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
        for (int which = 0; which < 2; ++which) {
            double T[9];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += U[3 * i + k] * (which ? W[3 * j + k] : W[3 * k + j]); T[3 * i + j] = s; }
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += T[3 * i + k] * Vt[3 * k + j]; Rs[which][3 * i + j] = s; }
        }
        ts[0] = U[2]; ts[1] = U[5]; ts[2] = U[8];
    }
    __syncthreads();
    // depth signs for the four combinations (R1, t), (R2, t), (R1, -t), (R2, -t): z1 R x1 + t = z2 x2 in least squares
    const double* P = a.pts + (size_t)b * a.max_k * 4;
    double cnt[4] = {0, 0, 0, 0};
    // combination c of point i: both depths positive and below `dist`
    auto in_front = [&](int i, int c) {
        const double u1 = P[4 * i], v1 = P[4 * i + 1], u2 = P[4 * i + 2], v2 = P[4 * i + 3];
        const double* R = Rs[c & 1];
        const double sg = c < 2 ? 1.0 : -1.0;
        const double ax = R[0] * u1 + R[1] * v1 + R[2], ay = R[3] * u1 + R[4] * v1 + R[5], az = R[6] * u1 + R[7] * v1 + R[8];
        const double tx = sg * ts[0], ty = sg * ts[1], tz = sg * ts[2];
        const double aa = ax * ax + ay * ay + az * az, ab = -(ax * u2 + ay * v2 + az), bb = u2 * u2 + v2 * v2 + 1.0;
        const double ra = -(ax * tx + ay * ty + az * tz), rb = u2 * tx + v2 * ty + tz;
        double det = aa * bb - ab * ab;
        if (!(fabs(det) > 1e-300)) det = 1e-300;
        const double z1 = (ra * bb - ab * rb) / det, z2 = (aa * rb - ab * ra) / det;
        return z1 > 0 && z1 < a.dist && z2 > 0 && z2 < a.dist;
    };
    for (int i = tid; i < n; i += RS_THREADS) {
        if (!min_[i]) continue;
        for (int c = 0; c < 4; ++c) cnt[c] += in_front(i, c) ? 1.0 : 0.0;
    }
    block_sum_n<4>(cnt, scratch);
    int bestc = 0;
    for (int c = 1; c < 4; ++c) if (cnt[c] > cnt[bestc]) bestc = c;       // ties keep the earlier combination
    // the winning combination's mask: the same arithmetic again (any number of points per thread, nothing kept in between)
    for (int i = tid; i < n; i += RS_THREADS) if (min_[i] && in_front(i, bestc)) mout[i] = 1;
    if (tid < 9) a.Rt[12 * b + tid] = Rs[bestc & 1][tid];
    if (tid < 3) a.Rt[12 * b + 9 + tid] = (bestc < 2 ? 1.0 : -1.0) * ts[tid];
    if (tid == 0) a.good[b] = (int)cnt[bestc];
}

// ================================================================================================ RANSAC fundamental matrix
// cv2.findFundamentalMat(pts0, pts1, cv2.FM_RANSAC) as utils/mvg.py:16 calls it (threshold 3, confidence 0.99, maxIters
// 1000) -- PARITY UNPINNED like the two estimators above; numpy restatement: oracle/geometry_ref.py find_fundamental_ransac.
// One hypothesis per thread and round: 7 distinct matches (void when the last one is collinear with two earlier ones in
// either image), the two-dimensional null space of the 7 x 9 epipolar system by Gauss-Jordan, the cubic det(x B1 + B2),
// its real roots in closed form, up to three models, each scored against all matches with OpenCV's error (the larger
// squared point-to-epipolar-line distance, rounded to float32 as OpenCV stores it).  The winning minimal model is returned
// as it is (OpenCV does not refit it).
__device__ __forceinline__ double det3r(const double* r0, const double* r1, const double* r2)
{
    return r0[0] * (r1[1] * r2[2] - r1[2] * r2[1]) - r0[1] * (r1[0] * r2[2] - r1[2] * r2[0]) + r0[2] * (r1[0] * r2[1] - r1[1] * r2[0]);
}

// real roots of c3 x^3 + c2 x^2 + c1 x + c0 (trigonometric form for three, Cardano for one); a vanishing c3 voids the sample
__device__ int solve_cubic(double c3, double c2, double c1, double c0, double* r)
{
    if (!(c3 != 0.0) || !isfinite(c3) || !isfinite(c2) || !isfinite(c1) || !isfinite(c0)) return 0;
    const double inv = 1.0 / c3;
    const double a1 = c2 * inv, a2 = c1 * inv, a3 = c0 * inv;
    const double Q = (a1 * a1 - 3.0 * a2) * (1.0 / 9.0);
    const double R = (2.0 * a1 * a1 * a1 - 9.0 * a1 * a2 + 27.0 * a3) * (1.0 / 54.0);
    const double Q3 = Q * Q * Q, d = Q3 - R * R, t2 = a1 * (1.0 / 3.0);
    if (d > 0.0) {
        const double theta = acos(fmin(fmax(R / sqrt(Q3), -1.0), 1.0)), t0 = -2.0 * sqrt(Q);
        r[0] = t0 * cos(theta * (1.0 / 3.0)) - t2;
        r[1] = t0 * cos(theta * (1.0 / 3.0) + 1.0 * (2.0 * 3.141592653589793 / 3.0)) - t2;
        r[2] = t0 * cos(theta * (1.0 / 3.0) + 2.0 * (2.0 * 3.141592653589793 / 3.0)) - t2;
        return 3;
    }
    double e = cbrt(sqrt(-d) + fabs(R));
    if (R > 0.0) e = -e;
    r[0] = (e != 0.0 ? e + Q / e : 0.0) - t2;
    return 1;
}

// x1 / x2: the sample's seven pixel points (u, v interleaved).  F: up to three models, F[8] = 1.  Returns their number.
__device__ int fundamental_7pt(const double* x1, const double* x2, double (*F)[9], bool* valid)
{
    for (int s = 0; s < 3; ++s) valid[s] = false;
    double A[7 * 9];
    for (int i = 0; i < 7; ++i) {
        const double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        double* q = A + 9 * i;
        q[0] = u2 * u1; q[1] = u2 * v1; q[2] = u2; q[3] = v2 * u1; q[4] = v2 * v1; q[5] = v2; q[6] = u1; q[7] = v1; q[8] = 1.0;
    }
    if (!gauss_jordan(A, 7, 9, 7)) return 0;
    double B1[9], B2[9];
    for (int e = 0; e < 7; ++e) { B1[e] = -A[e * 9 + 7]; B2[e] = -A[e * 9 + 8]; }
    B1[7] = 1.0; B1[8] = 0.0; B2[7] = 0.0; B2[8] = 1.0;
    const double c3 = det3r(B1, B1 + 3, B1 + 6), c0 = det3r(B2, B2 + 3, B2 + 6);
    const double c2 = det3r(B2, B1 + 3, B1 + 6) + det3r(B1, B2 + 3, B1 + 6) + det3r(B1, B1 + 3, B2 + 6);
    const double c1 = det3r(B1, B2 + 3, B2 + 6) + det3r(B2, B1 + 3, B2 + 6) + det3r(B2, B2 + 3, B1 + 6);
    double roots[3] = {0.0, 0.0, 0.0};
    const int nr = solve_cubic(c3, c2, c1, c0, roots);
    int nvalid = 0;
    for (int s = 0; s < nr; ++s) {
        bool fin = isfinite(roots[s]);
        double f[9];
        for (int e = 0; e < 9; ++e) f[e] = roots[s] * B1[e] + B2[e];
        const double sc = f[8];
        const bool big = fabs(sc) > RS_EPS;
        for (int e = 0; e < 9; ++e) { F[s][e] = big ? f[e] / sc : f[e]; fin = fin && isfinite(F[s][e]); }
        valid[s] = fin;
        nvalid += fin;
    }
    return nvalid;
}

// OpenCV's FMEstimatorCallback::computeError of one correspondence, rounded to float32 as OpenCV stores it
__device__ __forceinline__ float fm_err(const double* F, double u1, double v1, double u2, double v2)
{
    double a = F[0] * u1 + F[1] * v1 + F[2], b = F[3] * u1 + F[4] * v1 + F[5], c = F[6] * u1 + F[7] * v1 + F[8];
    const double s2 = 1.0 / (a * a + b * b), d2 = u2 * a + v2 * b + c;
    a = F[0] * u2 + F[3] * v2 + F[6]; b = F[1] * u2 + F[4] * v2 + F[7]; c = F[2] * u2 + F[5] * v2 + F[8];
    const double s1 = 1.0 / (a * a + b * b), d1 = u1 * a + v1 * b + c;
    return (float)fmax(d1 * d1 * s1, d2 * d2 * s2);
}

struct FundArgs {
    const float* m0; int cols0; const float* m1; int cols1;
    int max_k; const int32_t* k_dev; const float* scale;      // [B][4]: normalised -> pixels, fp32 (FundamentalMatrix.py:76-77)
    const uint32_t* seed_dev; uint32_t seed;
    double threshold, confidence; int max_iters;
    double* F; uint8_t* mask; int32_t* info;                  // [B][9], [B][max_k], [B][4] = (found, inliers, hypotheses, 0)
};

__global__ __launch_bounds__(RS_THREADS) void ransac_fundamental(FundArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4* pts = reinterpret_cast<float4*>(smem);                       // [n] (u1, v1, u2, v2) pixels
    __shared__ double bestF[9];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.k_dev ? min(a.k_dev[b], a.max_k) : a.max_k;
    const uint32_t seed = a.seed_dev ? a.seed_dev[b] : a.seed;
    const float* sc = a.scale + 4 * b;
    for (int i = tid; i < n; i += RS_THREADS) {
        const float* p0 = a.m0 + ((size_t)b * a.max_k + i) * a.cols0;
        const float* p1 = a.m1 + ((size_t)b * a.max_k + i) * a.cols1;
        pts[i] = make_float4(p0[0] * sc[0], p0[1] * sc[1], p1[0] * sc[2], p1[1] * sc[3]);
    }
    uint8_t* mask = a.mask + (size_t)b * a.max_k;
    for (int i = tid; i < a.max_k; i += RS_THREADS) mask[i] = 0;
    __syncthreads();
    int32_t* info = a.info + 4 * b;
    double* Fout = a.F + 9 * b;
    if (n < 8) {        // utils/mvg.py:13-15 never calls cv2 below eight matches
        if (tid < 9) Fout[tid] = 0.0;
        if (tid == 0) { info[0] = 0; info[1] = 0; info[2] = 0; info[3] = 0; }
        return;
    }
    const float t2 = (float)(a.threshold * a.threshold);
    __shared__ unsigned short s_idx[RS_THREADS * 7];
    __shared__ unsigned char s_ok[RS_THREADS];
    __shared__ int s_cnt[RS_THREADS * 3];
    __shared__ int s_ctl[4];            // niters, stop, kept (hypothesis * 3 + root slot of this round, -1: none), failed
    CvRng rng{rng_state(seed)};
    RsState st{max(a.max_iters, 1), 0, 0, 0, 0};
    bool alive = true, have = false;
    int base = 0;
    auto check = [&](const int* idx) {       // FMEstimatorCallback::checkSubset
        double x1[14], x2[14];
        for (int j = 0; j < 7; ++j) { const float4 q = pts[idx[j]]; x1[2 * j] = q.x; x1[2 * j + 1] = q.y; x2[2 * j] = q.z; x2[2 * j + 1] = q.w; }
        return !last_collinear<7>(x1) && !last_collinear<7>(x2);
    };
    if (n < FM_LMEDS_BELOW) {
        // cv::findFundamentalMat (OpenCV 4.9 fundam.cpp): `(method & ~3) == FM_RANSAC && npoints >= 15` runs the RANSAC registrator,
        // FEWER THAN 15 correspondences go to the least-median one whatever the method.  LMeDSPointSetRegistrator::run:
        // niters = max(RANSACUpdateNumIters(confidence, 0.45, 7, maxIters), 3), the same RNG / getSubset (1000 attempts) / checkSubset
        // stream, every model scored by the median of its float32 errors (element n / 2 in sorted order), the strictly smallest
        // median kept; inliers = error <= sigma^2 with sigma = max(2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median), 0.001); the model
        // stands when at least 7 points are inliers.  info[3] = 1 marks the branch.  PARITY UNPINNED (see the top of this file).
        const int niters = max(update_iters(a.confidence, 0.45, 7, max(a.max_iters, 1)), 3);
        __shared__ double s_minmed;
        __shared__ int s_good;
        if (tid == 0) { s_minmed = INFINITY; s_good = 0; }
        int iters = 0;
        for (;;) {
            if (tid == 0) gen_round<7>(rng, n, s_idx, s_ok, alive, check, 1000);
            __syncthreads();
            double x1[14], x2[14];
            for (int j = 0; j < 7; ++j) { const float4 q = pts[s_idx[tid * 7 + j]]; x1[2 * j] = q.x; x1[2 * j + 1] = q.y; x2[2 * j] = q.z; x2[2 * j + 1] = q.w; }
            double F[3][9];
            bool valid[3] = {false, false, false};
            int nv = 0;
            if (s_ok[tid]) nv = fundamental_7pt(x1, x2, F, valid);
            const int k = n / 2;
            for (int s = 0; s < 3; ++s) {
                float med = INFINITY;
                bool got = false;
                if (nv && valid[s])
                    for (int i = 0; i < n; ++i) {               // the k-th smallest without an array: n <= 14
                        const float4 qi = pts[i];
                        const float ei = fm_err(F[s], qi.x, qi.y, qi.z, qi.w);
                        int less = 0, leq = 0;
                        for (int j = 0; j < n; ++j) { const float4 q = pts[j]; const float ej = fm_err(F[s], q.x, q.y, q.z, q.w); less += ej < ei; leq += ej <= ei; }
                        if (less <= k && k < leq) { med = ei; got = true; }
                    }
                s_cnt[tid * 3 + s] = got ? __float_as_int(med) : -1;      // errors are non-negative: their bit patterns are too
            }
            __syncthreads();
            if (tid == 0) {
                int kept = -1, stop = 0;
                double mm = s_minmed;
                for (int j = 0; j < 256; ++j) {
                    const int it = base + j;
                    if (it >= niters || !s_ok[j]) { stop = 1; break; }        // `if (iter == 0) return false; break;`
                    for (int sl = 0; sl < 3; ++sl) {
                        const int c = s_cnt[j * 3 + sl];
                        if (c >= 0 && (double)__int_as_float(c) < mm) { mm = (double)__int_as_float(c); kept = j * 3 + sl; }
                    }
                    iters = it + 1;
                }
                s_minmed = mm; s_ctl[0] = iters; s_ctl[1] = stop; s_ctl[2] = kept;
            }
            __syncthreads();
            if (s_ctl[2] >= 0 && s_ctl[2] / 3 == tid) for (int e = 0; e < 9; ++e) bestF[e] = F[s_ctl[2] % 3][e];
            have = have || s_ctl[2] >= 0;
            base += RS_THREADS;
            const bool more = !s_ctl[1] && base < niters;
            __syncthreads();
            if (!more) break;
        }
        iters = s_ctl[0];
        if (!have) {
            if (tid < 9) Fout[tid] = 0.0;
            if (tid == 0) { info[0] = 0; info[1] = 0; info[2] = iters; info[3] = 1; }
            return;
        }
        double Fb[9];
        for (int e = 0; e < 9; ++e) Fb[e] = bestF[e];
        const double sigma = fmax(2.5 * 1.4826 * (1.0 + 5.0 / (n - 7)) * sqrt(s_minmed), 0.001);
        const float ts = (float)(sigma * sigma);
        for (int i = tid; i < n; i += RS_THREADS) {
            const float4 q = pts[i];
            const int in = fm_err(Fb, q.x, q.y, q.z, q.w) <= ts;
            mask[i] = (uint8_t)in;
            if (in) atomicAdd(&s_good, 1);
        }
        __syncthreads();
        const int good = s_good, found = good >= 7;
        if (tid < 9) Fout[tid] = found ? Fb[tid] : 0.0;
        if (tid == 0) { info[0] = found; info[1] = good; info[2] = iters; info[3] = 1; }
        return;
    }
    for (;;) {
        if (tid == 0) gen_round<7>(rng, n, s_idx, s_ok, alive, check);
        __syncthreads();
        double x1[14], x2[14];
        for (int j = 0; j < 7; ++j) { const float4 q = pts[s_idx[tid * 7 + j]]; x1[2 * j] = q.x; x1[2 * j + 1] = q.y; x2[2 * j] = q.z; x2[2 * j + 1] = q.w; }
        double F[3][9];
        bool valid[3] = {false, false, false};
        int nv = 0;
        if (s_ok[tid]) nv = fundamental_7pt(x1, x2, F, valid);
        for (int s = 0; s < 3; ++s) {
            int cnt = 0;
            if (nv && valid[s])
                for (int i = 0; i < n; ++i) { const float4 q = pts[i]; cnt += fm_err(F[s], q.x, q.y, q.z, q.w) <= t2; }
            s_cnt[tid * 3 + s] = cnt;
        }
        __syncthreads();
        if (tid == 0) {
            const int kept = scan_round<7, 3>(s_cnt, s_ok, base, n, a.confidence, st);
            s_ctl[0] = st.niters; s_ctl[1] = st.stop; s_ctl[2] = kept; s_ctl[3] = st.failed;
        }
        __syncthreads();
        if (s_ctl[2] >= 0 && s_ctl[2] / 3 == tid) for (int e = 0; e < 9; ++e) bestF[e] = F[s_ctl[2] % 3][e];
        have = have || s_ctl[2] >= 0;
        base += RS_THREADS;
        const bool more = !s_ctl[1] && base < s_ctl[0];
        __syncthreads();
        if (!more) break;
    }
    if (!have) {
        if (tid < 9) Fout[tid] = 0.0;
        if (tid == 0) { info[0] = 0; info[1] = 0; info[2] = st.iters; info[3] = 0; }
        return;
    }
    double Fb[9];
    for (int e = 0; e < 9; ++e) Fb[e] = bestF[e];
    for (int i = tid; i < n; i += RS_THREADS) { const float4 q = pts[i]; mask[i] = fm_err(Fb, q.x, q.y, q.z, q.w) <= t2; }
    if (tid < 9) Fout[tid] = Fb[tid];
    if (tid == 0) { info[0] = 1; info[1] = st.max_good; info[2] = st.iters; info[3] = 0; }
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int kpb_find_fundamental(
    kpb_ctx* ctx, const float* m0_dev, int cols0, const float* m1_dev, int cols1, int batch, int max_k, const int32_t* k_dev,
    const float* scale_dev, const uint32_t* seed_dev, uint32_t seed, const kpb_ransac_params* prm, double* out_f_dev, uint8_t* out_mask_dev,
    int32_t* out_info_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_find_fundamental: null context");
    if (batch <= 0 || max_k < 0 || cols0 < 2 || cols1 < 2 || !scale_dev || !prm || !out_f_dev || !out_info_dev || (max_k && (!m0_dev || !m1_dev || !out_mask_dev)))
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_find_fundamental: bad argument");
    if ((size_t)max_k * sizeof(float4) > 128 * 1024)
        return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_find_fundamental: at most %d matches per pair", (int)(128 * 1024 / sizeof(float4)));
    if (!(prm->threshold > 0.0) || prm->max_iters < 1) return kpb_fail(ctx, KPB_E_INVALID, "kpb_find_fundamental: bad parameters");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    FundArgs a{m0_dev, cols0, m1_dev, cols1, max_k, k_dev, scale_dev, seed_dev, seed, prm->threshold, prm->confidence, prm->max_iters,
               out_f_dev, out_mask_dev, out_info_dev};
    if (!(ctx->lds_attr_done & KPB_ATTR_FUNDAMENTAL)) {
        KPB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ransac_fundamental), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ctx->lds_attr_done |= KPB_ATTR_FUNDAMENTAL;
    }
    KPB_LAUNCH(ctx, "ransac_fundamental", ransac_fundamental, dim3(batch), dim3(RS_THREADS), (size_t)max_k * sizeof(float4), ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}

extern "C" __attribute__((visibility("default"))) int kpb_find_essential(
    kpb_ctx* ctx, const float* m0_dev, int cols0, const float* m1_dev, int cols1, int batch, int max_k, const int32_t* k_dev,
    const float* scale_dev, const double* cam_dev, int cam_f32, const double* thr_dev, const uint32_t* seed_dev, uint32_t seed, double prob, int max_iters,
    double* out_e_dev, uint8_t* out_mask_dev, int32_t* out_info_dev, double* out_pts_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_find_essential: null context");
    if (batch <= 0 || max_k < 0 || cols0 < 2 || cols1 < 2 || !scale_dev || !cam_dev || !thr_dev || !out_e_dev || !out_info_dev ||
        (max_k && (!m0_dev || !m1_dev || !out_mask_dev || !out_pts_dev)) || max_iters < 1)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_find_essential: bad argument");
    const size_t lds = (size_t)max_k * 4 * sizeof(double);
    if (lds > 128 * 1024)      // the matches live in LDS for the whole search (config_vo.yaml's top_k 2000 needs 64 KB)
        return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_find_essential: at most 4096 matches per pair (got %d)", max_k);
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    if (lds > 48 * 1024 && !(ctx->lds_attr_done & KPB_ATTR_ESSENTIAL)) {
        KPB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ransac_essential), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        ctx->lds_attr_done |= KPB_ATTR_ESSENTIAL;
    }
    EssArgs a{m0_dev, cols0, m1_dev, cols1, max_k, k_dev, scale_dev, cam_dev, cam_f32, thr_dev, seed_dev, seed, prob, max_iters, out_e_dev, out_mask_dev,
              out_info_dev, out_pts_dev};
    KPB_LAUNCH(ctx, "ransac_essential", ransac_essential, dim3(batch), dim3(RS_THREADS), lds, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}

extern "C" __attribute__((visibility("default"))) int kpb_recover_pose(
    kpb_ctx* ctx, const double* e_dev, const double* pts_dev, const uint8_t* mask_dev, int batch, int max_k, const int32_t* k_dev,
    const int32_t* info_dev, double dist, double* out_rt_dev, uint8_t* out_mask_dev, int32_t* out_good_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_recover_pose: null context");
    if (batch <= 0 || max_k < 0 || !e_dev || !info_dev || !out_rt_dev || !out_good_dev || (max_k && (!pts_dev || !mask_dev || !out_mask_dev)))
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_recover_pose: bad argument");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    PoseArgs a{e_dev, pts_dev, mask_dev, max_k, k_dev, info_dev, dist, out_rt_dev, out_mask_dev, out_good_dev};
    KPB_LAUNCH(ctx, "recover_pose", recover_pose, dim3(batch), dim3(RS_THREADS), 0, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}
