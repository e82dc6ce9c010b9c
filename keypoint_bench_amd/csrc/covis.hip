// covis.hip -- covisibility warp and ground-truth mutual nearest neighbours.
//
//   kpb_warp_homography   utils/projection.py:137-167  warp_homography
//   kpb_val_keypoints     tasks/repeatability.py:69-85 (inside val_key_points; mutual_argmax 9-32,
//                         compute_keypoints_distance 39-51)
//
// Both are fp32 index/compare work on a few thousand points per image pair: one wave per keypoint row (or column),
// lanes strided over the other set, everything recomputed from the 8-byte coordinates instead of materialising the
// M x N distance matrix the reference builds three times over.  The formulas are the ones torch's CPU kernels
// evaluate for the reference's lines (fixture tests/golden/covis.npz): the einsum row as h0*x, fma(h1, y, .), + h2;
// the 2-norm as sqrt(fma(dy, dy, dx*dx)).  The library is built with -ffp-contract=off, so only the fmaf calls fuse.
#include "kpb_common.h"

namespace {

constexpr int WARP_THREADS = 1024;

struct WarpArgs {
    const float* kps; int max_n, stride; const int32_t* n_dev;
    const float* hmat; const int32_t* wh;
    float* k0v; float* k01v; int32_t* ids; int32_t* n_valid;
};

__device__ __forceinline__ bool warp_point(const float* __restrict__ row, const float* __restrict__ h, float sx, float sy,
                                           float& x, float& y, float& u, float& v)
{
    x = row[0] * sx;                                                   // projection.py:144
    y = row[1] * sy;
    const float r0 = fmaf(h[1], y, h[0] * x) + h[2];                   // 146 (einsum over the homogeneous 1)
    const float r1 = fmaf(h[4], y, h[3] * x) + h[5];
    const float r2 = fmaf(h[7], y, h[6] * x) + h[8];
    u = r0 / r2;                                                       // 147
    v = r1 / r2;
    return u >= 0.f && u <= sx && v >= 0.f && v <= sy;                 // 153
}

// one workgroup per image: two passes over the keypoints (count, then ordered emit) so that the rejected ids can be
// appended behind the kept ones (156-157)
__global__ __launch_bounds__(WARP_THREADS) void warp_homography(WarpArgs a)
{
    const int b = blockIdx.x;
    const int n = a.n_dev ? min(a.n_dev[b], a.max_n) : a.max_n;
    const float* kps = a.kps + (size_t)b * a.max_n * a.stride;
    const float* h = a.hmat + 9 * b;
    const float sx = (float)(a.wh[2 * b] - 1), sy = (float)(a.wh[2 * b + 1] - 1);
    float* k0v = a.k0v + (size_t)b * a.max_n * 2;
    float* k01v = a.k01v + (size_t)b * a.max_n * 2;
    int32_t* ids = a.ids + (size_t)b * a.max_n;
    __shared__ int wave_cnt[WARP_THREADS / 64];
    __shared__ int base_valid, base_out, total_valid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    if (threadIdx.x == 0) total_valid = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += WARP_THREADS) {
        float x, y, u, v;
        mine += warp_point(kps + (size_t)i * a.stride, h, sx, sy, x, y, u, v);
    }
    for (int o = 32; o; o >>= 1) mine += __shfl_down(mine, o);
    if (lane == 0 && mine) atomicAdd(&total_valid, mine);
    if (threadIdx.x == 0) { base_valid = 0; base_out = 0; }
    __syncthreads();
    const int nv = total_valid;

    for (int i0 = 0; i0 < n; i0 += WARP_THREADS) {
        const int i = i0 + threadIdx.x;
        float x = 0, y = 0, u = 0, v = 0;
        const bool in = i < n;
        const bool ok = in && warp_point(kps + (size_t)i * a.stride, h, sx, sy, x, y, u, v);
        const unsigned long long m = __ballot(ok);
        const int before = __popcll(m & ((1ull << lane) - 1));
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wave_cnt[w];
        const int bv = base_valid, bo = base_out;
        if (ok) {
            const int p = bv + wbase + before;
            k0v[2 * p] = x / sx;  k0v[2 * p + 1] = y / sy;                 // 163
            k01v[2 * p] = u / sx; k01v[2 * p + 1] = v / sy;                // 164
            ids[p] = i;
        } else if (in) {
            const int rank_out = (wave * 64 + lane) - (wbase + before);    // rejected points ahead of me in this chunk
            ids[nv + bo + rank_out] = i;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int c = 0;
            for (int w = 0; w < WARP_THREADS / 64; ++w) c += wave_cnt[w];
            base_valid = bv + c;
            base_out = bo + min(WARP_THREADS, n - i0) - c;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) a.n_valid[b] = nv;
}

// ------------------------------------------------------------------------------------------------ warp_se3
// utils/projection.py:195-268 with interpolate_depth (271-373), unproject (30-53) and project (56-75): lift a keypoint of
// image 0 through its interpolated depth, move it with pose01, project it into image 1 and accept it if the depth map
// there agrees to 5 cm.  The three einsums are evaluated as torch's BLAS does for the point counts that matter (a
// fused-multiply-add chain; below 400 multiply-adds torch runs an unfused loop: <= 1 ulp apart).
struct Se3Args {
    const float* kps; int n, stride;
    const float* depth0; int H0, W0; const float* depth1; int H1, W1;
    const float* cam;       // kinv0[9], k1[9], pose[16], bbox0[2], bbox1[2]
    float* k0v; float* k01v; int32_t* ids_valid; int32_t* ids_out; int32_t* counts;
};

// 0: corner outside the bordered image, 1: a corner without depth, 2: interpolated (z valid)
__device__ __forceinline__ int se3_interp(const float* __restrict__ depth, int h, int w, float x, float y, float& z)
{
    constexpr int border = 10;
    const float i = y, j = x;
    const long it = (long)floorf(i), jt = (long)floorf(j), ib = (long)ceilf(i), jr = (long)ceilf(j);
    if (!(it >= border && jt >= border && jr < w - border && ib < h - border)) return 0;
    const float dtl = depth[it * w + jt], dtr = depth[it * w + jr], dbl = depth[ib * w + jt], dbr = depth[ib * w + jr];
    if (!(dtl > 0.f && dtr > 0.f && dbl > 0.f && dbr > 0.f)) return 1;
    const float di = i - (float)it, dj = j - (float)jt;
    const float wtl = (1.f - di) * (1.f - dj), wtr = (1.f - di) * dj, wbl = di * (1.f - dj), wbr = di * dj;
    z = ((wtl * dtl + wtr * dtr) + wbl * dbl) + wbr * dbr;
    return 2;
}

__device__ __forceinline__ float dot3f(const float* a, float b0, float b1, float b2) { return fmaf(a[2], b2, fmaf(a[1], b1, a[0] * b0)); }

// class of point i: 0 dropped (no depth in image 0, or corners but no depth in image 1), 1 outside image 1, 2 occluded,
// 3 covisible; x, y = position in image 0, u, v = position in image 1 (pixels)
__device__ __forceinline__ int se3_point(const Se3Args& a, int i, float& x, float& y, float& u01, float& v01)
{
    const float* kinv0 = a.cam; const float* k1 = a.cam + 9; const float* pose = a.cam + 18;
    const float* bbox0 = a.cam + 34; const float* bbox1 = a.cam + 36;
    x = a.kps[(size_t)i * a.stride] * (float)a.W0; y = a.kps[(size_t)i * a.stride + 1] * (float)a.H0;      // 204
    float z0;
    if (se3_interp(a.depth0, a.H0, a.W0, x, y, z0) != 2) return 0;                                          // 211
    const float bu = (x + bbox0[1]) + 0.5f, bv = (y + bbox0[0]) + 0.5f;                                     // 214
    const float d0 = bu * z0, d1 = bv * z0;                                                                 // 42
    float p[3], q[3], zuv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) p[r] = dot3f(kinv0 + 3 * r, d0, d1, z0);                                    // 46
#pragma unroll
    for (int r = 0; r < 3; ++r) { const float* t = pose + 4 * r; q[r] = fmaf(t[2], p[2], fmaf(t[1], p[1], t[0] * p[0])) + t[3]; }   // 221
#pragma unroll
    for (int r = 0; r < 3; ++r) zuv[r] = dot3f(k1 + 3 * r, q[0], q[1], q[2]);                               // 68
    const float u = zuv[0] / zuv[2], v = zuv[1] / zuv[2], z01 = zuv[2];                                     // 73-75
    u01 = (u - bbox1[1]) - 0.5f; v01 = (v - bbox1[0]) - 0.5f;                                               // 227
    float z1;
    const int c = se3_interp(a.depth1, a.H1, a.W1, u01, v01, z1);                                           // 234
    if (c == 0) return 1;                                                                                   // 236-239
    if (c == 1) return 0;
    return fabsf(z01 - z1) < 0.05f ? 3 : 2;                                                                 // 246-249
}

// one workgroup: count the three lists, then emit them in ascending point order (ids_out = outside, then occluded: 262)
__global__ __launch_bounds__(WARP_THREADS) void warp_se3(Se3Args a)
{
    __shared__ int tot[4], base[4], wave_cnt[3][WARP_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 4) { tot[threadIdx.x] = 0; base[threadIdx.x] = 0; }
    __syncthreads();
    int mine[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < a.n; i += WARP_THREADS) {
        float x, y, u, v;
        mine[se3_point(a, i, x, y, u, v)]++;
    }
#pragma unroll
    for (int c = 1; c < 4; ++c) {
        int m = mine[c];
        for (int o = 32; o; o >>= 1) m += __shfl_down(m, o);
        if (lane == 0 && m) atomicAdd(&tot[c], m);
    }
    __syncthreads();
    const int n_outside = tot[1];
    for (int i0 = 0; i0 < a.n; i0 += WARP_THREADS) {
        const int i = i0 + threadIdx.x;
        float x = 0, y = 0, u = 0, v = 0;
        const int cls = i < a.n ? se3_point(a, i, x, y, u, v) : 0;
        int before[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 1; c < 4; ++c) {
            const unsigned long long m = __ballot(cls == c);
            before[c] = __popcll(m & ((1ull << lane) - 1));
            if (lane == 0) wave_cnt[c - 1][wave] = __popcll(m);
        }
        __syncthreads();
        if (cls) {
            int p = base[cls] + before[cls];
            for (int w = 0; w < wave; ++w) p += wave_cnt[cls - 1][w];
            if (cls == 3) {
                a.k0v[2 * p] = x / (float)a.W0; a.k0v[2 * p + 1] = y / (float)a.H0;                         // 266
                a.k01v[2 * p] = u / (float)a.W1; a.k01v[2 * p + 1] = v / (float)a.H1;                       // 267
                a.ids_valid[p] = i;
            } else {
                a.ids_out[(cls == 2 ? n_outside : 0) + p] = i;
            }
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            int c = 0;
            for (int w = 0; w < WARP_THREADS / 64; ++w) c += wave_cnt[threadIdx.x][w];
            base[threadIdx.x + 1] += c;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { a.counts[0] = tot[3]; a.counts[1] = tot[1] + tot[2]; }
}

// ------------------------------------------------------------------------------------------------ val_key_points
struct ValArgs {
    const float* k0; const float* k01; const float* k1; const float* k10;
    int max_m, max_n; const int32_t* m_dev; const int32_t* n_dev;
    const float* scale;                 // [batch][2] = (scale01, scale10)
    float th;
    float* rmin; float* cmin;           // [batch][max_m], [batch][max_n]
    unsigned* dmax;                     // [batch]  bits of the largest dm (>= 0, so the bit pattern orders like the value)
    int32_t* cnt;                       // [batch][max_m] mutual cells per row, then their exclusive scan
    int32_t* pairs; float* dist; int cap;
    float* errors; int32_t* counts;     // [batch][max_m], [batch][2]
    float* dm;                          // [batch][max_m][max_n] or null (r05): the cells, written by the row pass and READ by the three later passes instead of
                                        // being evaluated four times (two correctly rounded square roots each: the passes were arithmetic-bound, 2.8 ms per 256
                                        // pairs of 1000 x 1000; 1 GB per such batch -- what 288 GB of HBM are for); null above 4 GB: every pass evaluates its cells
};

__device__ __forceinline__ float kp_dist(float ax, float ay, float bx, float by)
{
    const float dx = ax - bx, dy = ay - by;                             // repeatability.py:49
    return sqrtf(fmaf(dy, dy, dx * dx));                                // 50 (torch.norm, p=2)
}

// dm[i][j] (69-73); nd = min(M, N) bounds the masked diagonal
__device__ __forceinline__ float dm_cell(const float2 a0, const float2 a01, const float2 b1, const float2 b10, int i, int j, int nd)
{
    const float d = (kp_dist(a0.x, a0.y, b10.x, b10.y) + kp_dist(b1.x, b1.y, a01.x, a01.y)) / 2.f;
    return (i == j && i < nd) ? 99999.f : d;
}

__device__ __forceinline__ float wave_min(float v) { return kpb_wave_fmin(v); }
__device__ __forceinline__ float wave_max(float v) { return kpb_wave_fmax(v); }

// wave w < M: row w (min over columns, max over columns); wave w >= M: column w - M
__global__ __launch_bounds__(256) void covis_stats(ValArgs a)
{
    const int b = blockIdx.y;
    const int M = a.m_dev ? min(a.m_dev[b], a.max_m) : a.max_m, N = a.n_dev ? min(a.n_dev[b], a.max_n) : a.max_n;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= M + N || M == 0 || N == 0) return;
    const float2* k0 = (const float2*)a.k0 + (size_t)b * a.max_m; const float2* k01 = (const float2*)a.k01 + (size_t)b * a.max_m;
    const float2* k1 = (const float2*)a.k1 + (size_t)b * a.max_n; const float2* k10 = (const float2*)a.k10 + (size_t)b * a.max_n;
    const int nd = min(M, N);
    if (w < M) {
        const int i = w;
        const float2 a0 = k0[i], a01 = k01[i];
        float lo = INFINITY, hi = 0.f;
        float* row = a.dm ? a.dm + ((size_t)b * a.max_m + i) * a.max_n : nullptr;
#pragma unroll 4            // four column blocks' coordinates in flight (rolled, every block of 64 cells waited for its own two loads)
        for (int j = lane; j < N; j += 64) {
            const float d = dm_cell(a0, a01, k1[j], k10[j], i, j, nd);
            if (row) row[j] = d;
            lo = fminf(lo, d); hi = fmaxf(hi, d);
        }
        lo = wave_min(lo); hi = wave_max(hi);
        if (lane == 0) {
            a.rmin[(size_t)b * a.max_m + i] = lo;
            a.errors[(size_t)b * a.max_m + i] = lo * a.scale[2 * b + 1];          // 79/81 then 85: min commutes with the positive scale
            atomicMax(a.dmax + b, __float_as_uint(hi));
        }
    } else if (!a.dm) {
        const int j = w - M;
        const float2 b1 = k1[j], b10 = k10[j];
        float lo = INFINITY;
        for (int i = lane; i < M; i += 64) lo = fminf(lo, dm_cell(k0[i], k01[i], b1, b10, i, j, nd));
        lo = wave_min(lo);
        if (lane == 0) a.cmin[(size_t)b * a.max_n + j] = lo;
    }
}

// column minima from the stored cells: a wave takes 64 adjacent columns and walks down the rows (256 contiguous bytes per load)
__global__ __launch_bounds__(256) void covis_colmin(ValArgs a)
{
    const int b = blockIdx.y;
    const int M = a.m_dev ? min(a.m_dev[b], a.max_m) : a.max_m, N = a.n_dev ? min(a.n_dev[b], a.max_n) : a.max_n;
    const int j = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63);
    if (j >= N || M == 0) return;
    const float* col = a.dm + (size_t)b * a.max_m * a.max_n + j;
    float lo = INFINITY;
#pragma unroll 8
    for (int i = 0; i < M; ++i) lo = fminf(lo, col[(size_t)i * a.max_n]);
    a.cmin[(size_t)b * a.max_n + j] = lo;
}

// value = (-dm) - min(-dm) (18, 36) is a monotone rounding of dm, so a row's maximum of `value` is the image of the
// row's minimum of dm; a cell is mutual when its image equals both (20-32).  EMIT = false: count per row;
// EMIT = true: write the cells of row i behind cnt[i] (already scanned), ascending j.
template <bool EMIT>
__global__ __launch_bounds__(256) void covis_mutual(ValArgs a)
{
    const int b = blockIdx.y;
    const int M = a.m_dev ? min(a.m_dev[b], a.max_m) : a.max_m, N = a.n_dev ? min(a.n_dev[b], a.max_n) : a.max_n;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= M || N == 0) return;
    const float2* k1 = (const float2*)a.k1 + (size_t)b * a.max_n; const float2* k10 = (const float2*)a.k10 + (size_t)b * a.max_n;
    const float2 a0 = ((const float2*)a.k0)[(size_t)b * a.max_m + i], a01 = ((const float2*)a.k01)[(size_t)b * a.max_m + i];
    const float* cmin = a.cmin + (size_t)b * a.max_n;
    const int nd = min(M, N);
    const float c = -__uint_as_float(a.dmax[b]);                    // value.min()
    const float vr = (-a.rmin[(size_t)b * a.max_m + i]) - c;
    const float s01 = a.scale[2 * b];
    int run = EMIT ? a.cnt[(size_t)b * a.max_m + i] : 0;
    int gt = 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        bool hit = false;
        float d = 0.f;
        if (j < N) {
            d = a.dm ? a.dm[((size_t)b * a.max_m + i) * a.max_n + j] : dm_cell(a0, a01, k1[j], k10[j], i, j, nd);
            const float v = (-d) - c;
            hit = v == vr && v == (-cmin[j]) - c;
        }
        const unsigned long long m = __ballot(hit);
        if (EMIT) {
            const int p = run + __popcll(m & ((1ull << lane) - 1));
            if (hit) {
                const float ds = d * s01;                               // 77 / 80
                if (p < a.cap) {
                    a.pairs[((size_t)b * a.cap + p) * 2] = i; a.pairs[((size_t)b * a.cap + p) * 2 + 1] = j;
                    a.dist[(size_t)b * a.cap + p] = ds;
                }
                gt += ds <= a.th;                                       // 82
            }
        }
        run += __popcll(m);
    }
    if (EMIT) {
        for (int o = 32; o; o >>= 1) gt += __shfl_down(gt, o);
        if (lane == 0 && gt) atomicAdd(a.counts + 2 * b + 1, gt);
    } else if (lane == 0) {
        a.cnt[(size_t)b * a.max_m + i] = run;
    }
}

// exclusive scan of the per-row counts (one workgroup per pair); counts[0] = number of mutual cells
__global__ __launch_bounds__(1024) void covis_scan(ValArgs a)
{
    const int b = blockIdx.x;
    const int M = a.m_dev ? min(a.m_dev[b], a.max_m) : a.max_m, N = a.n_dev ? min(a.n_dev[b], a.max_n) : a.max_n;
    int32_t* cnt = a.cnt + (size_t)b * a.max_m;
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows = (N == 0) ? 0 : M;
    for (int i0 = 0; i0 < rows; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const int v = i < rows ? cnt[i] : 0;
        int s = v;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(s, o); if (lane >= o) s += t; }
        if (lane == 63) wsum[wave] = s;
        __syncthreads();
        int wb = 0;
        for (int w = 0; w < wave; ++w) wb += wsum[w];
        const int base = carry;
        if (i < rows) cnt[i] = base + wb + s - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + wb + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.counts[2 * b] = carry;
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int kpb_warp_homography(
    kpb_ctx* ctx, const float* kps_dev, int batch, int max_n, int stride, const int32_t* n_dev, const float* hmat_dev,
    const int32_t* wh_dev, float* out_kps0_dev, float* out_kps01_dev, int32_t* out_ids_dev, int32_t* out_n_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_warp_homography: null context");
    if (batch <= 0 || max_n < 0 || stride < 2 || !hmat_dev || !wh_dev || !out_n_dev)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_warp_homography: bad argument");
    if (max_n && (!kps_dev || !out_kps0_dev || !out_kps01_dev || !out_ids_dev))
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_warp_homography: null buffer");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    WarpArgs a{kps_dev, max_n, stride, n_dev, hmat_dev, wh_dev, out_kps0_dev, out_kps01_dev, out_ids_dev, out_n_dev};
    KPB_LAUNCH(ctx, "warp_homography", warp_homography, dim3(batch), dim3(WARP_THREADS), 0, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}

extern "C" __attribute__((visibility("default"))) int kpb_val_keypoints(
    kpb_ctx* ctx, const float* k0_dev, const float* k01_dev, const float* k1_dev, const float* k10_dev, int batch,
    int max_m, int max_n, const int32_t* m_dev, const int32_t* n_dev, const float* scale_dev, float th,
    int32_t* out_pairs_dev, float* out_dist_dev, int cap, float* out_errors_dev, int32_t* out_counts_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_val_keypoints: null context");
    if (batch <= 0 || max_m < 0 || max_n < 0 || cap < 0 || !scale_dev || !out_counts_dev)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_val_keypoints: bad argument");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    KPB_HIP(ctx, hipMemsetAsync(out_counts_dev, 0, (size_t)batch * 2 * sizeof(int32_t), ctx->stream));
    if (max_m == 0 || max_n == 0) return KPB_OK;
    if (!k0_dev || !k01_dev || !k1_dev || !k10_dev || !out_errors_dev || (cap && (!out_pairs_dev || !out_dist_dev)))
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_val_keypoints: null buffer");
    const size_t cells = (size_t)batch * max_m * max_n;
    // the cells are kept when they fit the context's limit (KPB_OPT_COVIS_STORE_BYTES, default 4 GiB = 256 pairs of 2000 x 2000) AND the
    // workspace can hold them: a device too full for the cells gets the evaluate-in-place form (every pass computes its cells: the
    // same bits, 2.8 instead of 2.4 ms per 256 pairs), not KPB_E_NOMEM
    bool store = cells * 4 <= ctx->covis_store_bytes;
    const size_t small = (size_t)batch * (2 * (size_t)max_m + max_n + 1);
    if (store && kpb_reserve(ctx, ctx->ws_misc, (small + cells + 64) * 4) != KPB_OK) store = false;
    if (!store)
        if (int rc = kpb_reserve(ctx, ctx->ws_misc, small * 4)) return rc;
    float* rmin = (float*)ctx->ws_misc.p;
    float* cmin = rmin + (size_t)batch * max_m;
    int32_t* cnt = (int32_t*)(cmin + (size_t)batch * max_n);
    unsigned* dmax = (unsigned*)(cnt + (size_t)batch * max_m);
    KPB_HIP(ctx, hipMemsetAsync(dmax, 0, (size_t)batch * 4, ctx->stream));
    float* dm = store ? reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(dmax + batch) + 255) & ~(uintptr_t)255) : nullptr;
    ValArgs a{k0_dev, k01_dev, k1_dev, k10_dev, max_m, max_n, m_dev, n_dev, scale_dev, th, rmin, cmin, dmax, cnt,
              out_pairs_dev, out_dist_dev, cap, out_errors_dev, out_counts_dev, dm};
    KPB_LAUNCH(ctx, "covis_stats", covis_stats, dim3(cdiv(max_m + (store ? 0 : max_n), 4), batch), dim3(256), 0, ctx->stream, a);
    if (store) KPB_LAUNCH(ctx, "covis_colmin", covis_colmin, dim3(cdiv(max_n, 256), batch), dim3(256), 0, ctx->stream, a);
    KPB_LAUNCH(ctx, "covis_count", covis_mutual<false>, dim3(cdiv(max_m, 4), batch), dim3(256), 0, ctx->stream, a);
    KPB_LAUNCH(ctx, "covis_scan", covis_scan, dim3(batch), dim3(1024), 0, ctx->stream, a);
    KPB_LAUNCH(ctx, "covis_emit", covis_mutual<true>, dim3(cdiv(max_m, 4), batch), dim3(256), 0, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}

extern "C" __attribute__((visibility("default"))) int kpb_warp_se3(
    kpb_ctx* ctx, const float* kps_dev, int n, int stride, const float* depth0_dev, int H0, int W0, const float* depth1_dev, int H1, int W1,
    const float* cam_dev, float* out_kps0_dev, float* out_kps01_dev, int32_t* out_ids_dev, int32_t* out_ids_out_dev, int32_t* out_counts_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_warp_se3: null context");
    if (n < 0 || stride < 2 || !depth0_dev || !depth1_dev || !cam_dev || !out_counts_dev || H0 <= 0 || W0 <= 0 || H1 <= 0 || W1 <= 0)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_warp_se3: bad argument");
    if (n && (!kps_dev || !out_kps0_dev || !out_kps01_dev || !out_ids_dev || !out_ids_out_dev))
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_warp_se3: null buffer");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    Se3Args a{kps_dev, n, stride, depth0_dev, H0, W0, depth1_dev, H1, W1, cam_dev, out_kps0_dev, out_kps01_dev, out_ids_dev, out_ids_out_dev, out_counts_dev};
    KPB_LAUNCH(ctx, "warp_se3", warp_se3, dim3(1), dim3(WARP_THREADS), 0, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}
