// match.hip -- M1..M3 of the hot path: bilinear descriptor sampling, float64 brute-force mutual
// nearest neighbour, row gather.  Replaces utils/matcher.py:221-233 of the reference, including the
// call into skimage.feature.match_descriptors (-> scipy cdist, float64) at lines 227-230.
//
// Distances are accumulated exactly as scipy's C loop does -- (double)a - (double)b, d*d, running sum
// in ascending k, no FMA contraction -- so argmin decisions (first index on ties) are reproduced bit
// for bit rather than to a tolerance.  This is VALU fp64 work (3 ops per element, 192 MFLOP per
// 1000x1000x64 pair), LDS-tiled 64x64 with 4x4 register blocking; it is not reshaped into an MFMA
// GEMM because the |a|^2+|b|^2-2ab expansion changes the rounding and flips near-tie argmins.
#include "kpb_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ M1
struct SampleArgs {
    const float* desc; const float* pts; const int* n; float* out;
    int C, Hd, Wd, pts_cols, max_n;
    long long sb, sc, sh, sw;
};

__global__ __launch_bounds__(256) void sample_bilinear(SampleArgs a)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = a.n ? min(a.n[b], a.max_n) : a.max_n;
    if (i >= n) return;
    const float* p = a.pts + ((size_t)b * a.max_n + i) * a.pts_cols;
    // matcher.py:221-222 then ATen's align_corners=True un-normalisation (g + 1) * ((size - 1) / 2)
    const float gx = (p[0] - 0.5f) * 2.0f, gy = (p[1] - 0.5f) * 2.0f;
    const float x = (gx + 1.0f) * ((float)(a.Wd - 1) / 2.0f), y = (gy + 1.0f) * ((float)(a.Hd - 1) / 2.0f);
    const float xw = floorf(x), yn = floorf(y);
    const float w = x - xw, e = 1.0f - w, nn = y - yn, s = 1.0f - nn;
    const float c_nw = s * e, c_ne = s * w, c_sw = nn * e, c_se = nn * w;
    const long long x0 = (long long)xw, y0 = (long long)yn, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < a.Wd, vx1 = x1 >= 0 && x1 < a.Wd;
    const bool vy0 = y0 >= 0 && y0 < a.Hd, vy1 = y1 >= 0 && y1 < a.Hd;
    const float* base = a.desc + (size_t)b * a.sb;
    float* o = a.out + ((size_t)b * a.max_n + i) * a.C;
    for (int ch = lane; ch < a.C; ch += 64) {
        const float* q = base + (size_t)ch * a.sc;
        const float nw = (vx0 && vy0) ? q[y0 * a.sh + x0 * a.sw] : 0.0f;   // zero padding
        const float ne = (vx1 && vy0) ? q[y0 * a.sh + x1 * a.sw] : 0.0f;
        const float sw = (vx0 && vy1) ? q[y1 * a.sh + x0 * a.sw] : 0.0f;
        const float se = (vx1 && vy1) ? q[y1 * a.sh + x1 * a.sw] : 0.0f;
        o[ch] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(nw, c_nw), __fmul_rn(ne, c_ne)), __fmul_rn(sw, c_sw)),
                          __fmul_rn(se, c_se));
    }
}

// ------------------------------------------------------------------------------------------------ M2
// (s1, i1) precedes (s2, i2) in the order scipy/numpy induce: sqrt(s) ascending, index ascending.
// sqrt can merge two sums that differ by an ulp, so near-ties are decided on the rooted values.
// A NaN distance (a NaN descriptor) is the minimum, as numpy.argmin has it: the first NaN wins.
__device__ __forceinline__ bool precedes(double s1, int i1, double s2, int i2)
{
    if (s1 == s2) return i1 < i2;
    const bool n1 = s1 != s1, n2 = s2 != s2;
    if (n1 | n2) return n1 && (!n2 || i1 < i2);
    const double hi = fmax(s1, s2), lo = fmin(s1, s2);
    if (hi - lo <= hi * 4.5e-16) {
        const double d1 = sqrt(s1), d2 = sqrt(s2);
        if (d1 == d2) return i1 < i2;
        return d1 < d2;
    }
    return s1 < s2;
}

constexpr int TM = 4, TN = 8;                                 // distances per thread: TM rows x TN columns
constexpr int MTI = 16 * TM, MTJ = 16 * TN, KC = 16, MATCH_THREADS = 256;

struct MatchArgs {
    const float* d0; const float* d1; const int* n; const int* m;
    double* rpart_s; int* rpart_j;   // [B][tiles_j][max_n]
    double* cpart_s; int* cpart_i;   // [B][tiles_i][max_m]
    int C, max_n, max_m, tiles_i, tiles_j;
};

// One MTI x MTJ tile of the distance matrix per workgroup, TM x TN of it per thread: every float64 operand read from LDS
// feeds TN (TM) subtract / multiply / add triples, which keeps the LDS pipe far below the vector ALUs' rate.
//
// Column ownership (r02): thread column c owns the tile columns 2c, 2c+1, 32+2c, 32+2c+1, 64+.., 96+.. (COLJ below), not
// eight consecutive ones: the four 16-byte reads of a k step then find the 16 lanes of a read group on 16 consecutive
// slots (r01: lanes 64 bytes apart -> every group on 4 bank sets, 67 % conflict cycles).  Row pitches are padded by two
// doubles so that the k-major staging writes of a float4 (four k rows) fall on different banks pairwise.
#define COLJ(c, q) (2 * (c) + 32 * ((q) >> 1) + ((q) & 1))
__global__ __launch_bounds__(MATCH_THREADS) void match_tile(MatchArgs a)
{
    constexpr int PA = MTI + 2, PB = MTJ + 2;                             // padded pitches (doubles)
    constexpr int STAGE = KC * (PA + PB) * 2, RED = 16 * MTJ * 3;         // words: operand tiles / column-minimum exchange
    __shared__ __attribute__((aligned(16))) unsigned smem[STAGE > RED ? STAGE : RED];
    double (*A)[PA] = reinterpret_cast<double (*)[PA]>(smem);
    double (*Bt)[PB] = reinterpret_cast<double (*)[PB]>(smem + KC * PA * 2);
    const int b = blockIdx.z, tj = blockIdx.x, ti = blockIdx.y, tid = threadIdx.x;
    const int n = a.n ? min(a.n[b], a.max_n) : a.max_n, m = a.m ? min(a.m[b], a.max_m) : a.max_m;
    const int i0 = ti * MTI, j0 = tj * MTJ;
    if (i0 >= n || j0 >= m) return;   // partials of empty tiles are never read (finalize clips to n, m)
    const float* d0 = a.d0 + (size_t)b * a.max_n * a.C;
    const float* d1 = a.d1 + (size_t)b * a.max_m * a.C;
    const int r = tid >> 4, c = tid & 15;        // thread owns rows i0+TM*r.., cols j0+TN*c..
    double acc[TM][TN];
#pragma unroll
    for (int p = 0; p < TM; ++p)
#pragma unroll
        for (int q = 0; q < TN; ++q) acc[p][q] = 0.0;

    const bool vec = (a.C % 4) == 0;
    for (int k0 = 0; k0 < a.C; k0 += KC) {
        // stage KC channels of the tile's rows of each side as float64, channel-major (float4 loads when C allows)
        if (vec) {
            for (int x = tid; x < (MTI + MTJ) * (KC / 4); x += MATCH_THREADS) {
                const int row = x / (KC / 4), kq = x - row * (KC / 4), kk = k0 + 4 * kq;
                const bool second = row >= MTI;
                const int lr = second ? row - MTI : row;
                const float* src = second ? d1 + (size_t)min(j0 + lr, m - 1) * a.C : d0 + (size_t)min(i0 + lr, n - 1) * a.C;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kk < a.C) v = *reinterpret_cast<const float4*>(src + kk);
                double* dst = second ? &Bt[4 * kq][lr] : &A[4 * kq][lr];
                const int pitch = second ? PB : PA;
                dst[0] = (double)v.x; dst[pitch] = (double)v.y; dst[2 * pitch] = (double)v.z; dst[3 * pitch] = (double)v.w;
            }
        } else {
            for (int x = tid; x < (MTI + MTJ) * KC; x += MATCH_THREADS) {
                const int row = x % (MTI + MTJ), k = x / (MTI + MTJ), kk = k0 + k;
                if (row < MTI) A[k][row] = (kk < a.C) ? (double)d0[(size_t)min(i0 + row, n - 1) * a.C + kk] : 0.0;
                else Bt[k][row - MTI] = (kk < a.C) ? (double)d1[(size_t)min(j0 + row - MTI, m - 1) * a.C + kk] : 0.0;
            }
        }
        __syncthreads();
        const int kend = min(KC, a.C - k0);
        for (int k = 0; k < kend; ++k) {
            double av[TM], bv[TN];
#pragma unroll
            for (int p = 0; p < TM; ++p) av[p] = A[k][TM * r + p];
#pragma unroll
            for (int q = 0; q < TN; ++q) bv[q] = Bt[k][COLJ(c, q)];
#pragma unroll
            for (int p = 0; p < TM; ++p)
#pragma unroll
                for (int q = 0; q < TN; ++q) {
                    const double d = __dsub_rn(av[p], bv[q]);
                    acc[p][q] = __dadd_rn(acc[p][q], __dmul_rn(d, d));   // no FMA: scipy's s += d*d
                }
        }
        __syncthreads();
    }

    // row minima over this tile's columns: TN local columns, then the 16 lanes that share r
    const int lane = tid & 63;
#pragma unroll
    for (int p = 0; p < TM; ++p) {
        double bs = __longlong_as_double(0x7FF0000000000000LL);
        int bj = 0x7FFFFFFF;
#pragma unroll
        for (int q = 0; q < TN; ++q) {
            const int j = j0 + COLJ(c, q);
            if (j < m && precedes(acc[p][q], j, bs, bj)) { bs = acc[p][q]; bj = j; }
        }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const double os = __shfl_xor(bs, d, 64);
            const int oj = __shfl_xor(bj, d, 64);
            if (precedes(os, oj, bs, bj)) { bs = os; bj = oj; }
        }
        const int i = i0 + TM * r + p;
        if ((lane & 15) == 0 && i < n) {
            const size_t o = ((size_t)b * a.tiles_j + tj) * a.max_n + i;
            a.rpart_s[o] = bs; a.rpart_j[o] = bj;
        }
    }
    // column minima over this tile's rows: TM local rows, then across the 16 row groups through LDS (the operand
    // tiles are dead: the last k loop ended with a barrier)
    double (*cs)[MTJ] = reinterpret_cast<double (*)[MTJ]>(smem);
    int (*ci)[MTJ] = reinterpret_cast<int (*)[MTJ]>(smem + 16 * MTJ * 2);
#pragma unroll
    for (int q = 0; q < TN; ++q) {
        double bs = __longlong_as_double(0x7FF0000000000000LL);
        int bi = 0x7FFFFFFF;
#pragma unroll
        for (int p = 0; p < TM; ++p) {
            const int i = i0 + TM * r + p;
            if (i < n && precedes(acc[p][q], i, bs, bi)) { bs = acc[p][q]; bi = i; }
        }
        cs[r][COLJ(c, q)] = bs; ci[r][COLJ(c, q)] = bi;
    }
    __syncthreads();
    if (tid < MTJ) {
        double bs = cs[0][tid]; int bi = ci[0][tid];
        for (int g = 1; g < 16; ++g)
            if (precedes(cs[g][tid], ci[g][tid], bs, bi)) { bs = cs[g][tid]; bi = ci[g][tid]; }
        const int j = j0 + tid;
        if (j < m) {
            const size_t o = ((size_t)b * a.tiles_i + ti) * a.max_m + j;
            a.cpart_s[o] = bs; a.cpart_i[o] = bi;
        }
    }
}

constexpr int FIN_THREADS = 1024;

struct FinArgs {
    const double* rpart_s; const int* rpart_j; const double* cpart_s; const int* cpart_i;
    const int* n; const int* m;
    int* out_pairs; double* out_dist; int* out_k;
    int max_n, max_m, tiles_i, tiles_j, cross_check;
    double max_distance;
};

__global__ __launch_bounds__(FIN_THREADS) void match_finalize(FinArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* colarg = reinterpret_cast<int*>(smem);   // [max_m]
    __shared__ int wsum[FIN_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = a.n ? min(a.n[b], a.max_n) : a.max_n, m = a.m ? min(a.m[b], a.max_m) : a.max_m;
    const int tiles_i = (n + MTI - 1) / MTI, tiles_j = (m + MTJ - 1) / MTJ;
    for (int j = tid; j < m; j += FIN_THREADS) {   // argmin(distances, axis=0)
        double bs = __longlong_as_double(0x7FF0000000000000LL);
        int bi = 0x7FFFFFFF;
        for (int t = 0; t < tiles_i; ++t) {
            const size_t o = ((size_t)b * a.tiles_i + t) * a.max_m + j;
            const double s = a.cpart_s[o]; const int i = a.cpart_i[o];
            if (precedes(s, i, bs, bi)) { bs = s; bi = i; }
        }
        colarg[j] = bi;
    }
    __syncthreads();
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += FIN_THREADS) {   // argmin(distances, axis=1), cross-check, max_distance
        const int i = c0 + tid;
        bool keep = false;
        double dist = 0.0;
        int bj = 0;
        if (i < n && m > 0) {
            double bs = __longlong_as_double(0x7FF0000000000000LL);
            bj = 0x7FFFFFFF;
            for (int t = 0; t < tiles_j; ++t) {
                const size_t o = ((size_t)b * a.tiles_j + t) * a.max_n + i;
                const double s = a.rpart_s[o]; const int j = a.rpart_j[o];
                if (precedes(s, j, bs, bj)) { bs = s; bj = j; }
            }
            dist = sqrt(bs);
            // skimage filters on distance only `if max_distance < np.inf`: inf / nan distances survive max_distance = inf
            keep = (!a.cross_check || colarg[bj] == i) && (!(a.max_distance < __longlong_as_double(0x7FF0000000000000LL)) || dist < a.max_distance);
        }
        // ordered compaction (rows stay sorted by i, as numpy boolean masking leaves them)
        const unsigned long long bal = __ballot(keep);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wid] = __popcll(bal);
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < FIN_THREADS / 64; ++w) { const int s = wsum[w]; if (w < wid) wbase += s; tot += s; }
        __syncthreads();
        if (keep) {
            const int pos = base + wbase + within;
            a.out_pairs[((size_t)b * a.max_n + pos) * 2 + 0] = i;
            a.out_pairs[((size_t)b * a.max_n + pos) * 2 + 1] = bj;
            if (a.out_dist) a.out_dist[(size_t)b * a.max_n + pos] = dist;
        }
        base += tot;
    }
    if (tid == 0) a.out_k[b] = base;
}

// ------------------------------------------------------------------------------------------------ M3
__global__ void gather_rows(const float* src, int src_rows, int cols, const int* idx, int idx_rows, int idx_stride,
                            int idx_col, const int* k, float* out)
{
    const int b = blockIdx.y;
    const int kk = k ? min(k[b], idx_rows) : idx_rows;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = e / cols, c = e - i * cols;
    if (i >= kk) return;
    const int r = idx[((size_t)b * idx_rows + i) * idx_stride + idx_col];
    out[((size_t)b * idx_rows + i) * cols + c] = src[((size_t)b * src_rows + r) * cols + c];
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int kpb_sample(kpb_ctx* ctx, const float* desc_dev, int batch, int C, int Hd, int Wd, int64_t sb,
                          int64_t sc, int64_t sh, int64_t sw, const float* pts_dev, int pts_cols, int max_n,
                          const int32_t* n_dev, float* out_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_sample: null context");
    if (max_n == 0) return KPB_OK;
    if (!desc_dev || !pts_dev || !out_dev || batch <= 0 || C <= 0 || Hd <= 0 || Wd <= 0 || pts_cols < 2 || max_n < 0)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_sample: bad argument");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    SampleArgs a{desc_dev, pts_dev, n_dev, out_dev, C, Hd, Wd, pts_cols, max_n, sb, sc, sh, sw};
    KPB_LAUNCH(ctx, "sample_bilinear", sample_bilinear, dim3(cdiv(max_n, 4), batch), dim3(256), 0, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}

extern "C" __attribute__((visibility("default"))) int kpb_match(kpb_ctx* ctx, const float* d0_dev, const float* d1_dev, int batch, int C, int max_n,
                         int max_m, const int32_t* n_dev, const int32_t* m_dev, const kpb_match_params* prm,
                         int32_t* out_pairs_dev, double* out_dist_dev, int32_t* out_k_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_match: null context");
    if (!prm || !out_k_dev || batch <= 0 || C <= 0 || max_n < 0 || max_m < 0)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_match: bad argument");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    if (max_n == 0 || max_m == 0) {   // nothing can match; mirror the empty result
        KPB_HIP(ctx, hipMemsetAsync(out_k_dev, 0, (size_t)batch * sizeof(int), ctx->stream));
        return KPB_OK;
    }
    if (!d0_dev || !d1_dev || !out_pairs_dev) return kpb_fail(ctx, KPB_E_INVALID, "kpb_match: null buffer");
    if (max_m > 16384) return kpb_fail(ctx, KPB_E_INVALID, "kpb_match: max_m %d > 16384", max_m);
    const int tiles_i = cdiv(max_n, MTI), tiles_j = cdiv(max_m, MTJ);
    const size_t nr = (size_t)batch * tiles_j * max_n, nc = (size_t)batch * tiles_i * max_m;
    const size_t bytes = (nr + nc) * (sizeof(double) + sizeof(int)) + 64;
    if (int rc = kpb_reserve(ctx, ctx->ws_match, bytes)) return rc;
    double* rs = static_cast<double*>(ctx->ws_match.p);
    double* cs = rs + nr;
    int* rj = reinterpret_cast<int*>(cs + nc);
    int* ci = rj + nr;
    MatchArgs a{d0_dev, d1_dev, n_dev, m_dev, rs, rj, cs, ci, C, max_n, max_m, tiles_i, tiles_j};
    KPB_LAUNCH(ctx, "match_tile", match_tile, dim3(tiles_j, tiles_i, batch), dim3(MATCH_THREADS), 0, ctx->stream, a);
    FinArgs f{rs, rj, cs, ci, n_dev, m_dev, out_pairs_dev, out_dist_dev, out_k_dev,
              max_n, max_m, tiles_i, tiles_j, prm->cross_check, prm->max_distance};
    KPB_LAUNCH(ctx, "match_finalize", match_finalize, dim3(batch), dim3(FIN_THREADS), (size_t)max_m * sizeof(int), ctx->stream, f);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}

extern "C" __attribute__((visibility("default"))) int kpb_gather_rows(kpb_ctx* ctx, const float* src_dev, int batch, int src_rows, int cols,
                               const int32_t* idx_dev, int idx_rows, int idx_stride, int idx_col,
                               const int32_t* k_dev, float* out_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_gather_rows: null context");
    if (idx_rows == 0) return KPB_OK;
    if (!src_dev || !idx_dev || !out_dev || batch <= 0 || cols <= 0 || idx_rows < 0 || idx_stride <= 0)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_gather_rows: bad argument");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    KPB_LAUNCH(ctx, "gather_rows", gather_rows, dim3(cdiv(idx_rows * cols, 256), batch), dim3(256), 0, ctx->stream, src_dev,
                       src_rows, cols, idx_dev, idx_rows, idx_stride, idx_col, k_dev, out_dev);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}
