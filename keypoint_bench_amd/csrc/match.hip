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

constexpr int SAMPLE_KPW = 4;       // keypoints per wave: their sixteen taps per channel are requested together (one keypoint per wave left 1 KB in flight per wave)
__global__ __launch_bounds__(256) void sample_bilinear(SampleArgs a)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * SAMPLE_KPW;
    const int n = a.n ? min(a.n[b], a.max_n) : a.max_n;
    if (i0 >= n) return;
    float cf[SAMPLE_KPW][4];
    long long off[SAMPLE_KPW][4];
    bool ok[SAMPLE_KPW][4];
#pragma unroll
    for (int k = 0; k < SAMPLE_KPW; ++k) {
        const int i = min(i0 + k, n - 1);
        const float* p = a.pts + ((size_t)b * a.max_n + i) * a.pts_cols;
        // matcher.py:221-222 then ATen's align_corners=True un-normalisation (g + 1) * ((size - 1) / 2)
        const float gx = (p[0] - 0.5f) * 2.0f, gy = (p[1] - 0.5f) * 2.0f;
        const float x = (gx + 1.0f) * ((float)(a.Wd - 1) / 2.0f), y = (gy + 1.0f) * ((float)(a.Hd - 1) / 2.0f);
        const float xw = floorf(x), yn = floorf(y);
        const float w = x - xw, e = 1.0f - w, nn = y - yn, s = 1.0f - nn;
        cf[k][0] = s * e; cf[k][1] = s * w; cf[k][2] = nn * e; cf[k][3] = nn * w;
        const long long x0 = (long long)xw, y0 = (long long)yn, x1 = x0 + 1, y1 = y0 + 1;
        const bool vx0 = x0 >= 0 && x0 < a.Wd, vx1 = x1 >= 0 && x1 < a.Wd;
        const bool vy0 = y0 >= 0 && y0 < a.Hd, vy1 = y1 >= 0 && y1 < a.Hd;
        ok[k][0] = vx0 && vy0; ok[k][1] = vx1 && vy0; ok[k][2] = vx0 && vy1; ok[k][3] = vx1 && vy1;
        off[k][0] = ok[k][0] ? y0 * a.sh + x0 * a.sw : 0; off[k][1] = ok[k][1] ? y0 * a.sh + x1 * a.sw : 0;
        off[k][2] = ok[k][2] ? y1 * a.sh + x0 * a.sw : 0; off[k][3] = ok[k][3] ? y1 * a.sh + x1 * a.sw : 0;
    }
    const float* base = a.desc + (size_t)b * a.sb;
    for (int ch = lane; ch < a.C; ch += 64) {
        const float* q = base + (size_t)ch * a.sc;
        float t[SAMPLE_KPW][4];
#pragma unroll
        for (int k = 0; k < SAMPLE_KPW; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) t[k][c] = q[off[k][c]];
#pragma unroll
        for (int k = 0; k < SAMPLE_KPW; ++k) {
            const float nw = ok[k][0] ? t[k][0] : 0.0f, ne = ok[k][1] ? t[k][1] : 0.0f;      // zero padding
            const float sw = ok[k][2] ? t[k][2] : 0.0f, se = ok[k][3] ? t[k][3] : 0.0f;
            if (i0 + k < n)
                a.out[((size_t)b * a.max_n + i0 + k) * a.C + ch] =
                    __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(nw, cf[k][0]), __fmul_rn(ne, cf[k][1])), __fmul_rn(sw, cf[k][2])), __fmul_rn(se, cf[k][3]));
        }
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
    const int* only;                 // optional [B]: run only for pairs whose flag is set (the prefilter's exact fallback)
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
    if (a.only && !a.only[b]) return;
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
    int* host_k;        // pinned host mirror of out_k (kpb_match_counts), written by the kernel
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
            if (i != 0x7FFFFFFF && precedes(s, i, bs, bi)) { bs = s; bi = i; }        // INT_MAX: an empty slot of the prefilter path
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
                if (j != 0x7FFFFFFF && precedes(s, j, bs, bj)) { bs = s; bj = j; }
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
    if (tid == 0) { a.out_k[b] = base; a.host_k[b] = base; }
}

// ------------------------------------------------------------------------------------------------ M2, prefilter
// The exact float64 distances above cost 3 N M C non-fusable fp64 operations per pair (2.2 ms per 256 pairs, 65 % of what the
// fp64 vector ALUs deliver without FMA).  Only the arg-minima and the distances of the matched pairs have to be exact, so the
// bulk is replaced by a FILTER on the matrix cores and the exact arithmetic runs on what survives it:
//   1. match_prep: squared norms in fp32 and every descriptor split into two half-precision terms (x = hi + lo);
//   2. match_approx<.., 0>: D~_ij = |a_i|^2 + |b_j|^2 - 2 a_i.b_j with the dot products on v_mfma_f32_16x16x32_f16 (three MFMAs
//      per product, fp32 accumulation); row and column minima of D~;
//   3. match_approx<.., 1>: the same pass again; (i, j) is listed as a row candidate when D~_ij <= rowmin_i + mr_i and as a
//      column candidate when D~_ij <= colmin_j + mc_j.  Error bound of the filter, worst case: each operand is carried to
//      2^-20 relative (two toward-zero half-precision terms) or 2^-25 absolute (f16 subnormal grid), the dropped lo.lo terms
//      are <= 2^-20 of their product, the 3 C products are accumulated in fp32 (<= 3 C 2^-24 |a||b|), the norms are fp32 sums of
//      C squares:  |D~ - D| <= (2.9e-6 + 2.4e-7 C)(|a|^2 + |b|^2) + 2e-6 (|a| + |b|)  =: delta.  The exact arg-minimum j* of row i
//      satisfies D~_ij* <= D_ij* + delta_ij* <= D_ij' + delta_ij* <= rowmin_i + delta_ij' + delta_ij* (j' the approximate
//      arg-minimum), so mr_i = 2.5 x delta evaluated with the LARGEST column norm of the pair covers it with a quarter to
//      spare, for every exact tie as well; mc_j likewise with the largest row norm;
//   4. match_exact: scipy's float64 sum for the listed pairs only (same order, no FMA), filed under their row and column
//      in the slots match_finalize reads (the per-tile partial arrays of match_tile, one slot per column / row tile).
// A pair whose descriptors are not finite or reach 2^15 in magnitude (the split would saturate), or that needs more slots than
// there are tiles (massive exact ties), raises a flag: match_tile then runs for that pair alone (it returns at once for the others) and overwrites the slots.  Results
// are bit-identical to match_tile's by construction; tests/test_gpu_match.py checks them on the goldens and the tie / NaN cases.
typedef _Float16 mh8 __attribute__((ext_vector_type(8)));
typedef float mf4 __attribute__((ext_vector_type(4)));
constexpr unsigned INF_BITS = 0x7F800000u;
__device__ __forceinline__ float match_margin(float na, float nb, int C)
{
    return (7.5e-6f + 6.0e-7f * (float)C) * (na + nb) + 5.0e-6f * (sqrtf(na) + sqrtf(nb));
}

struct PreArgs {
    const float* d0; const float* d1; const int* n; const int* m;
    int C, max_n, max_m;
    uint4* h0; uint4* h1;            // [B][max][C/8] hi pieces then lo pieces (second half of each array)
    float* nrm0; float* nrm1;        // [B][max]
    unsigned* rowmin; unsigned* colmin;      // [B][max_n], [B][max_m] float bits, start at +inf
    int* rcnt; int* ccnt;            // slot counters
    int* flag;                       // [B] 1: exact fallback for the pair
    int* ncand;                      // [B]
    unsigned* nmax;                  // [B][2] largest squared norm of either side (float bits; zeroed by the host)
};

__global__ __launch_bounds__(256) void match_prep(PreArgs a)
{
    // eight lanes per descriptor row: lane q takes the 8-float pieces q, q + 8, ... (coalesced 32-byte loads, 16-byte stores)
    __shared__ unsigned s_max[2];
    const int b = blockIdx.y, tid = threadIdx.x, q = tid & 7;
    const int n = a.n ? min(a.n[b], a.max_n) : a.max_n, m = a.m ? min(a.m[b], a.max_m) : a.max_m;
    const int P8 = a.C / 8;
    if (tid < 2) s_max[tid] = 0u;
    __syncthreads();
    const int r = min(blockIdx.x * 32 + (tid >> 3), a.max_n + a.max_m - 1);      // (a duplicate of the last row does no harm)
    const bool second = r >= a.max_n;
    const int row = second ? r - a.max_n : r;
    const int cnt = second ? m : n, cap = second ? a.max_m : a.max_n;
    const float* src = (second ? a.d1 : a.d0) + ((size_t)b * cap + row) * a.C;
    uint4* dst = (second ? a.h1 : a.h0) + ((size_t)b * cap + row) * P8;
    const size_t lo_off = (size_t)gridDim.y * cap * P8;
    float nn = 0.0f, big = 0.0f;
    if (row < cnt) {
        for (int p8 = q; p8 < P8; p8 += 8) {
            const float4 u = *reinterpret_cast<const float4*>(src + 8 * p8), v = *reinterpret_cast<const float4*>(src + 8 * p8 + 4);
            const float f[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
            big = fmaxf(fmaxf(fmaxf(big, fmaxf(fabsf(u.x), fabsf(u.y))), fmaxf(fabsf(u.z), fabsf(u.w))),
                        fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            unsigned hw[4], lw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 hh = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(f[2 * k], f[2 * k + 1]));
                const h2 ll = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(f[2 * k] - (float)hh[0], f[2 * k + 1] - (float)hh[1]));
                hw[k] = __builtin_bit_cast(unsigned, hh); lw[k] = __builtin_bit_cast(unsigned, ll);
                nn = fmaf(f[2 * k], f[2 * k], nn); nn = fmaf(f[2 * k + 1], f[2 * k + 1], nn);
            }
            dst[p8] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            dst[lo_off + p8] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
    }
    nn += kpb_shfl_xor<1>(nn); nn += kpb_shfl_xor<2>(nn); nn += kpb_shfl_xor<4>(nn);       // the same tree for every row: deterministic
    // The filter's error bound (match_margin) holds while every component is carried to 2^-20 relative or 2^-25 absolute, i.e.
    // while its hi half does not saturate: a component of magnitude >= 2^15 (un-normalised descriptors can be anything) hands
    // the whole pair to the exact kernel, like a non-finite one.
    if (row < cnt && big >= 32768.0f) a.flag[b] = 1;
    if (row < cnt && q == 0) {
        if (!(nn < 3.0e38f)) a.flag[b] = 1;             // NaN / inf / overflow: the exact kernel takes this pair
        else atomicMax(&s_max[second ? 1 : 0], __float_as_uint(nn));
    }
    if (q == 0) {
        (second ? a.nrm1 : a.nrm0)[(size_t)b * cap + row] = nn;
        (second ? a.colmin : a.rowmin)[(size_t)b * cap + row] = INF_BITS;
        (second ? a.ccnt : a.rcnt)[(size_t)b * cap + row] = 0;
    }
    __syncthreads();
    if (tid < 2 && s_max[tid]) atomicMax(&a.nmax[2 * b + tid], s_max[tid]);      // one global atomic per workgroup and side
}

struct ApxArgs {
    const uint4* h0; const uint4* h1; const float* nrm0; const float* nrm1; const int* n; const int* m;
    int max_n, max_m, P8; size_t lo0, lo1;      // lo*: offset of the lo pieces inside h0 / h1
    unsigned* rowmin; unsigned* colmin;
    int2* cand; int* ncand; int cap; int* flag;
    const unsigned* nmax; int C;
};

// A wave owns RT 16-row tiles (all columns); KB = C / 32 k-blocks.  PASS 0: minima.  PASS 1: candidate list.
constexpr int CBUF = 192;       // candidates a wave collects in LDS before it reserves room in the pair's list with ONE global atomic

template <int KB, int RT, int PASS>
__global__ __launch_bounds__(256) void match_approx(ApxArgs a)
{
    __shared__ int2 cbuf[PASS == 1 ? 4 : 1][PASS == 1 ? CBUF : 1];
    int wcnt = 0;                   // wave-uniform: every append goes through a ballot
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    auto flush = [&]() {
        if (PASS == 1 && wcnt > 0) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&a.ncand[b], wcnt);
            base = __shfl(base, 0, 64);
            for (int i = lane; i < wcnt; i += 64) {
                if (base + i < a.cap) a.cand[(size_t)b * a.cap + base + i] = cbuf[PASS == 1 ? wv : 0][i];
                else a.flag[b] = 1;
            }
            wcnt = 0;
        }
    };
    const int n = a.n ? min(a.n[b], a.max_n) : a.max_n, m = a.m ? min(a.m[b], a.max_m) : a.max_m;
    if (a.flag[b]) return;                                   // the exact kernel takes this pair
    const int i0 = (blockIdx.x * 4 + wv) * (16 * RT);
    if (i0 >= n) return;
    const int c16 = lane & 15, g = lane >> 4;
    const uint4* A = a.h0 + (size_t)b * a.max_n * a.P8;
    const uint4* Bm = a.h1 + (size_t)b * a.max_m * a.P8;
    mh8 ah[RT][KB], al[RT][KB];
    float na[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = min(i0 + 16 * rt + c16, n - 1);      // A operand: lane (row c16, k-group g)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            ah[rt][kb] = __builtin_bit_cast(mh8, A[(size_t)row * a.P8 + 4 * kb + g]);
            al[rt][kb] = __builtin_bit_cast(mh8, A[a.lo0 + (size_t)row * a.P8 + 4 * kb + g]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                         // D rows of this lane: 4 g + r
            const int i = i0 + 16 * rt + 4 * g + r;
            na[rt][r] = i < n ? a.nrm0[(size_t)b * a.max_n + i] : __uint_as_float(INF_BITS);
        }
    }
    const float namax = __uint_as_float(a.nmax[2 * b]), nbmax = __uint_as_float(a.nmax[2 * b + 1]);
    float rmin[RT][4];        // PASS 1: rowmin_i + mr_i
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) rmin[rt][r] = PASS == 0 ? __uint_as_float(INF_BITS)
                                                             : (i0 + 16 * rt + 4 * g + r < n ? __uint_as_float(a.rowmin[(size_t)b * a.max_n + i0 + 16 * rt + 4 * g + r]) + match_margin(na[rt][r], nbmax, a.C) : -1.0f);
    for (int j0 = 0; j0 < m; j0 += 16) {
        const int j = j0 + c16;
        const int jr = min(j, m - 1);
        mh8 bh[KB], bl[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            bh[kb] = __builtin_bit_cast(mh8, Bm[(size_t)jr * a.P8 + 4 * kb + g]);
            bl[kb] = __builtin_bit_cast(mh8, Bm[a.lo1 + (size_t)jr * a.P8 + 4 * kb + g]);
        }
        const float nb = j < m ? a.nrm1[(size_t)b * a.max_m + j] : __uint_as_float(INF_BITS);
        const float cmin_j = PASS == 1 && j < m ? __uint_as_float(a.colmin[(size_t)b * a.max_m + j]) + match_margin(namax, nb, a.C) : -1.0f;     // colmin_j + mc_j
        float cm = __uint_as_float(INF_BITS);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            mf4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt][kb], bh[kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][kb], bl[kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt][kb], bh[kb], acc, 0, 0, 0);
            }
            float dd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)                       // entry (row 4 g + r of tile rt, column c16); +0 (never -0: the minima are kept as unsigned bit patterns); inf past the end
                dd[r] = fmaxf(__fadd_rn(__fadd_rn(na[rt][r], nb), -2.0f * acc[r]), 0.0f) + 0.0f;
            if (PASS == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { rmin[rt][r] = fminf(rmin[rt][r], dd[r]); cm = fminf(cm, dd[r]); }
            } else {
                unsigned rowc = 0, colc = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool fin = dd[r] < __uint_as_float(INF_BITS);
                    rowc |= (fin && dd[r] <= rmin[rt][r]) ? 1u << r : 0u;
                    colc |= (fin && dd[r] <= cmin_j) ? 1u << r : 0u;
                }
                const unsigned any = rowc | colc;
                if (__ballot(any != 0)) {                     // wave-uniform: most 16 x 16 tiles hold no candidate at all
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool hit = (any >> r) & 1u;
                        const unsigned long long bal = __ballot(hit);
                        if (bal) {
                            if (hit) cbuf[PASS == 1 ? wv : 0][wcnt + __popcll(bal & ((1ull << lane) - 1ull))] =
                                make_int2((i0 + 16 * rt + 4 * g + r) | (((rowc >> r) & 1u) ? 1 << 30 : 0) | (((colc >> r) & 1u) ? 1 << 29 : 0), j);
                            wcnt += __popcll(bal);
                            if (wcnt > CBUF - 64) flush();      // a wave's LDS operations execute in order: the copy sees the writes
                        }
                    }
                }
            }
        }
        if (PASS == 0) {            // column minimum over this wave's rows: over the four lane groups, then one atomic per column
            cm = fminf(cm, kpb_shfl_xor<16>(cm));
            cm = kpb_min32(cm);
            if (g == 0 && j < m) atomicMin(&a.colmin[(size_t)b * a.max_m + j], __float_as_uint(cm));
        }
    }
    flush();
    if (PASS == 0) {                // row minima: over the 16 columns of a lane group; each row belongs to exactly one wave
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = rmin[rt][r];
                v = fminf(v, kpb_shfl_xor<1>(v)); v = fminf(v, kpb_shfl_xor<2>(v));
                v = fminf(v, kpb_shfl_xor<4>(v)); v = fminf(v, kpb_shfl_xor<8>(v));
                const int i = i0 + 16 * rt + 4 * g + r;
                if (c16 == 0 && i < n) a.rowmin[(size_t)b * a.max_n + i] = __float_as_uint(v);
            }
    }
}

struct ExactArgs {
    const float* d0; const float* d1; const int2* cand; const int* ncand; int cap, C, max_n, max_m;
    int* rcnt; int* ccnt; double* rs; int* rj; double* cs; int* ci; int tiles_i, tiles_j; const int* m; const int* n; int* flag;
};

// scipy's float64 sum (ascending k, no FMA: the same three instructions as match_tile) for the listed pairs only
__global__ __launch_bounds__(256) void match_exact(ExactArgs a)
{
    const int b = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (a.flag[b] || e >= min(a.ncand[b], a.cap)) return;
    const int2 c = a.cand[(size_t)b * a.cap + e];
    const int i = c.x & 0x1FFFFFFF, j = c.y;
    const float* pa = a.d0 + ((size_t)b * a.max_n + i) * a.C;
    const float* pb = a.d1 + ((size_t)b * a.max_m + j) * a.C;
    double s = 0.0;
    for (int k = 0; k < a.C; k += 4) {
        const float4 u = *reinterpret_cast<const float4*>(pa + k), v = *reinterpret_cast<const float4*>(pb + k);
        double d = __dsub_rn((double)u.x, (double)v.x); s = __dadd_rn(s, __dmul_rn(d, d));
        d = __dsub_rn((double)u.y, (double)v.y); s = __dadd_rn(s, __dmul_rn(d, d));
        d = __dsub_rn((double)u.z, (double)v.z); s = __dadd_rn(s, __dmul_rn(d, d));
        d = __dsub_rn((double)u.w, (double)v.w); s = __dadd_rn(s, __dmul_rn(d, d));
    }
    const int n = a.n ? min(a.n[b], a.max_n) : a.max_n, m = a.m ? min(a.m[b], a.max_m) : a.max_m;
    const int slots_r = (m + MTJ - 1) / MTJ, slots_c = (n + MTI - 1) / MTI;      // what match_finalize reads for this pair
    if (c.x & (1 << 30)) {
        const int t = atomicAdd(&a.rcnt[(size_t)b * a.max_n + i], 1);
        if (t < slots_r) { const size_t o = ((size_t)b * a.tiles_j + t) * a.max_n + i; a.rs[o] = s; a.rj[o] = j; }
        else a.flag[b] = 1;
    }
    if (c.x & (1 << 29)) {
        const int t = atomicAdd(&a.ccnt[(size_t)b * a.max_m + j], 1);
        if (t < slots_c) { const size_t o = ((size_t)b * a.tiles_i + t) * a.max_m + j; a.cs[o] = s; a.ci[o] = i; }
        else a.flag[b] = 1;
    }
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
    KPB_LAUNCH(ctx, "sample_bilinear", sample_bilinear, dim3(cdiv(max_n, 4 * SAMPLE_KPW), batch), dim3(256), 0, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}

extern "C" __attribute__((visibility("default"))) int kpb_match(kpb_ctx* ctx, const float* d0_dev, const float* d1_dev, int batch, int C, int max_n,
                         int max_m, const int32_t* n_dev, const int32_t* m_dev, const kpb_match_params* prm,
                         int32_t* out_pairs_dev, double* out_dist_dev, int32_t* out_k_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_match: null context");
    ctx->match_counts_valid = 0;        // set again only when everything of this call has been enqueued
    if (!prm || !out_k_dev || batch <= 0 || C <= 0 || max_n < 0 || max_m < 0)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_match: bad argument");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->host_match_cap < batch) {
        KPB_HIP(ctx, hipStreamSynchronize(ctx->stream));        // an earlier call's match_finalize may still be writing the old mirror
        if (ctx->host_match) KPB_HIP(ctx, hipHostFree(ctx->host_match));
        ctx->host_match = nullptr; ctx->host_match_cap = 0;
        KPB_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->host_match), (size_t)batch * sizeof(int), hipHostMallocDefault));
        ctx->host_match_cap = batch;
    }
    ctx->host_match_n = batch;
    if (max_n == 0 || max_m == 0) {   // nothing can match; mirror the empty result
        KPB_HIP(ctx, hipMemsetAsync(out_k_dev, 0, (size_t)batch * sizeof(int), ctx->stream));
        KPB_HIP(ctx, hipStreamSynchronize(ctx->stream));        // (a kernel of an earlier call may still be writing the mirror)
        for (int b = 0; b < batch; ++b) ctx->host_match[b] = 0;
        ctx->match_counts_valid = 1;
        return KPB_OK;
    }
    if (!d0_dev || !d1_dev || !out_pairs_dev) return kpb_fail(ctx, KPB_E_INVALID, "kpb_match: null buffer");
    if (max_m > 16384) return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_match: max_m %d > 16384", max_m);
    const int tiles_i = cdiv(max_n, MTI), tiles_j = cdiv(max_m, MTJ);
    const size_t nr = (size_t)batch * tiles_j * max_n, nc = (size_t)batch * tiles_i * max_m;
    const size_t bytes = (nr + nc) * (sizeof(double) + sizeof(int)) + 64;
    if (int rc = kpb_reserve(ctx, ctx->ws_match, bytes)) return rc;
    double* rs = static_cast<double*>(ctx->ws_match.p);
    double* cs = rs + nr;
    int* rj = reinterpret_cast<int*>(cs + nc);
    int* ci = rj + nr;
    // prefilter on the matrix cores + exact refinement (see match_prep): C a multiple of 32 up to 256, at least two slots per row and column
    static const int prefilter = kpb_env_int("KPB_MATCH_PREFILTER", 1);
    // ... and a batch that fills the chip (KPB_MATCH_PREFILTER: 0 never, 1 = default: from 8 pairs, 2 always -- the tests' way to the prefilter with one pair): for a handful of pairs the prefilter's five dependent launches are LATENCY (a single
    // 1000 x 1000 pair: 0.30 ms with it, 0.175 ms on match_tile alone -- profiles/r04_single_pair_latency.txt); both give the same bits
    const bool pre = prefilter && (batch >= 8 || prefilter == 2) && (C % 32 == 0) && C <= 256 && tiles_i >= 2 && tiles_j >= 2 && (reinterpret_cast<uintptr_t>(d0_dev) % 16 == 0) &&
                     (reinterpret_cast<uintptr_t>(d1_dev) % 16 == 0);
    const int* only = nullptr;
    if (pre) {
        const int P8 = C / 8, cap = 8 * (max_n + max_m);
        const size_t nh0 = (size_t)batch * max_n * P8, nh1 = (size_t)batch * max_m * P8;
        const size_t words = 4 * 2 * (nh0 + nh1) + 3 * ((size_t)batch * (max_n + max_m)) + 4 * (size_t)batch + 2 * (size_t)batch * cap + 64;
        if (int rc = kpb_reserve(ctx, ctx->ws_misc, words * 4)) return rc;
        uint4* h0 = static_cast<uint4*>(ctx->ws_misc.p);
        uint4* h1 = h0 + 2 * nh0;
        float* nrm0 = reinterpret_cast<float*>(h1 + 2 * nh1);
        float* nrm1 = nrm0 + (size_t)batch * max_n;
        unsigned* rowmin = reinterpret_cast<unsigned*>(nrm1 + (size_t)batch * max_m);
        unsigned* colmin = rowmin + (size_t)batch * max_n;
        int* rcnt = reinterpret_cast<int*>(colmin + (size_t)batch * max_m);
        int* ccnt = rcnt + (size_t)batch * max_n;
        int* flag = ccnt + (size_t)batch * max_m;
        int* ncand = flag + batch;
        unsigned* nmax = reinterpret_cast<unsigned*>(ncand + batch);
        int2* cand = reinterpret_cast<int2*>(nmax + 2 * batch + ((reinterpret_cast<uintptr_t>(nmax + 2 * batch) & 4) ? 1 : 0));
        KPB_HIP(ctx, hipMemsetAsync(flag, 0, (size_t)batch * 4 * sizeof(int), ctx->stream));      // flag, ncand, nmax
        PreArgs pa{d0_dev, d1_dev, n_dev, m_dev, C, max_n, max_m, h0, h1, nrm0, nrm1, rowmin, colmin, rcnt, ccnt, flag, ncand, nmax};
        // empty slots are marked by the index INT_MAX (match_finalize skips them): one coalesced 32-bit fill of both index arrays
        KPB_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(rj), 0x7FFFFFFF, nr + nc, ctx->stream));
        KPB_LAUNCH(ctx, "match_prep", match_prep, dim3(cdiv(max_n + max_m, 32), batch), dim3(256), 0, ctx->stream, pa);
        ApxArgs xa{h0, h1, nrm0, nrm1, n_dev, m_dev, max_n, max_m, P8, nh0, nh1, rowmin, colmin, cand, ncand, cap, flag, nmax, C};
        const int KB = C / 32;
#define KPB_APX(KB_, RT_)                                                                                                                   \
        {                                                                                                                                       \
            const dim3 grid(cdiv(max_n, 64 * RT_), batch);                                                                                      \
            KPB_LAUNCH(ctx, "match_approx_min", (match_approx<KB_, RT_, 0>), grid, dim3(256), 0, ctx->stream, xa);                              \
            KPB_LAUNCH(ctx, "match_approx_cand", (match_approx<KB_, RT_, 1>), grid, dim3(256), 0, ctx->stream, xa);                             \
        }
        if (KB == 1) KPB_APX(1, 4) else if (KB == 2) KPB_APX(2, 4) else if (KB == 3) KPB_APX(3, 2) else if (KB == 4) KPB_APX(4, 2)
        else if (KB == 5) KPB_APX(5, 1) else if (KB == 6) KPB_APX(6, 1) else if (KB == 7) KPB_APX(7, 1) else KPB_APX(8, 1)
#undef KPB_APX
        ExactArgs ea{d0_dev, d1_dev, cand, ncand, cap, C, max_n, max_m, rcnt, ccnt, rs, rj, cs, ci, tiles_i, tiles_j, m_dev, n_dev, flag};
        KPB_LAUNCH(ctx, "match_exact", match_exact, dim3(cdiv(cap, 256), batch), dim3(256), 0, ctx->stream, ea);
        only = flag;
    }
    MatchArgs a{only, d0_dev, d1_dev, n_dev, m_dev, rs, rj, cs, ci, C, max_n, max_m, tiles_i, tiles_j};
    KPB_LAUNCH(ctx, "match_tile", match_tile, dim3(tiles_j, tiles_i, batch), dim3(MATCH_THREADS), 0, ctx->stream, a);
    FinArgs f{rs, rj, cs, ci, n_dev, m_dev, out_pairs_dev, out_dist_dev, out_k_dev, ctx->host_match,
              max_n, max_m, tiles_i, tiles_j, prm->cross_check, prm->max_distance};
    KPB_LAUNCH(ctx, "match_finalize", match_finalize, dim3(batch), dim3(FIN_THREADS), (size_t)max_m * sizeof(int), ctx->stream, f);
    KPB_HIP(ctx, hipGetLastError());
    ctx->match_counts_valid = 1;
    return KPB_OK;
}

extern "C" __attribute__((visibility("default"))) int kpb_match_counts(kpb_ctx* ctx, int32_t* out_k_host, int batch)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_match_counts: null context");
    if (!out_k_host || batch <= 0) return kpb_fail(ctx, KPB_E_INVALID, "kpb_match_counts: bad argument");
    if (!ctx->host_match || !ctx->match_counts_valid)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_match_counts: no completed match (the last kpb_match failed, was rejected, or none has run)");
    if (batch != ctx->host_match_n)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_match_counts: the last kpb_match had %d pairs, not %d", ctx->host_match_n, batch);
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    KPB_HIP(ctx, kpb_wait_stream(ctx, batch < 8));
    for (int b = 0; b < batch; ++b) out_k_host[b] = ctx->host_match[b];
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
