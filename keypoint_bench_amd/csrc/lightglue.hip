// lightglue.hip -- N5: the LightGlue matcher (models/lightglue.py: LightGlue.match 447-477, _forward 506-652 and the
// blocks it calls) on gfx950, fp32 throughout, following the reference's CPU path: exact-fp32 attention, early
// stopping (check_if_stop 670-681) and point pruning (get_pruning_mask 659-668) decided ON THE DEVICE, so a batch of
// pairs runs through all nine layers without a host round trip; pairs that stopped are skipped by every later kernel.
//
//   Linear layers   = conv_mfma<1,...> over tokens laid out as a 16-wide "image" (weights pre-packed once)
//   attention       = lg_flash: per wave 32 queries, K tile as the MFMA A operand and Q as B (S^T = K.Q^T), so each
//                     lane owns one query column and its softmax statistics; the probability tile then feeds the
//                     P.V MFMA straight from its accumulator registers (k order permuted, V rows loaded to match)
//   assignment      = NT GEMM of the projected descriptors, row/column log-sum-exp, mutual arg-max, threshold
#include "conv_mfma.h"

namespace {

constexpr int D = 256, NH = 4, HD = 64, NF = 32, NL = 9;

// ------------------------------------------------------------------------------------------------ preparation
struct PrepArgs {
    const float* pts0; const float* pts1;   // [B][K][3] normalised (x, y, score)
    const int* n0; const int* n1;           // [B] or null
    const float* Wr;                        // posenc.Wr.weight [32][2]
    float* kpx;                             // [S][MP][2] pixel keypoints
    float* cosb; float* sinb;               // [S][MP][32]
    int* ind; int* cnt; int* cnt_orig;      // [S][MP], [S], [S]
    int* active_seq; int* active_pair; int* stop;
    int* done_pair;                         // [B] 1 once the pair has finished (lg_decide): the assignment stage runs once, after the last layer, for these
    int K, MP; float w1, h1;                // w - 1, h - 1
};

// lightglue.py:449-450 (pixel keypoints), 45-57 (normalize_keypoints, size=None), 93-98 (Fourier encoding)
__global__ __launch_bounds__(256) void lg_prepare(PrepArgs a)
{
    __shared__ float red[4][4];
    const int s = blockIdx.x, b = s >> 1, side = s & 1, tid = threadIdx.x;
    const float* pts = (side ? a.pts1 : a.pts0) + (size_t)b * a.K * 3;
    const int* np = side ? a.n1 : a.n0;
    const int n = np ? min(np[b], a.K) : a.K;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int t = tid; t < n; t += 256) {
        const float x = pts[3 * t] * a.w1, y = pts[3 * t + 1] * a.h1;
        a.kpx[((size_t)s * a.MP + t) * 2] = x; a.kpx[((size_t)s * a.MP + t) * 2 + 1] = y;
        mnx = fminf(mnx, x); mny = fminf(mny, y); mxx = fmaxf(mxx, x); mxy = fmaxf(mxy, y);
    }
    mnx = kpb_wave_fmin(mnx); mny = kpb_wave_fmin(mny); mxx = kpb_wave_fmax(mxx); mxy = kpb_wave_fmax(mxy);
    if ((tid & 63) == 0) { red[tid >> 6][0] = mnx; red[tid >> 6][1] = mny; red[tid >> 6][2] = mxx; red[tid >> 6][3] = mxy; }
    __syncthreads();
    mnx = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0])); mny = fminf(fminf(red[0][1], red[1][1]), fminf(red[2][1], red[3][1]));
    mxx = fmaxf(fmaxf(red[0][2], red[1][2]), fmaxf(red[2][2], red[3][2])); mxy = fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3]));
    const float sx = (1.0f + mxx) - mnx, sy = (1.0f + mxy) - mny;       // size = 1 + max - min
    const float shx = sx / 2.0f, shy = sy / 2.0f, scale = fmaxf(sx, sy) / 2.0f;
    for (int i = tid; i < n * NF; i += 256) {
        const int t = i / NF, f = i - t * NF;
        const float kx = __fdiv_rn(a.kpx[((size_t)s * a.MP + t) * 2] - shx, scale), ky = __fdiv_rn(a.kpx[((size_t)s * a.MP + t) * 2 + 1] - shy, scale);
        const float pr = __fadd_rn(__fmul_rn(kx, a.Wr[2 * f]), __fmul_rn(ky, a.Wr[2 * f + 1]));
        a.cosb[((size_t)s * a.MP + t) * NF + f] = cosf(pr);
        a.sinb[((size_t)s * a.MP + t) * NF + f] = sinf(pr);
    }
    for (int t = tid; t < a.MP; t += 256) a.ind[(size_t)s * a.MP + t] = t;
    if (tid == 0) {
        a.cnt[s] = n; a.cnt_orig[s] = n;
        const int n_other = (side ? a.n0 : a.n1) ? min((side ? a.n0 : a.n1)[b], a.K) : a.K;
        const int act = (n > 0 && n_other > 0) ? 1 : 0;       // lightglue.py:553-554: nothing to do without keypoints
        a.active_seq[s] = act;
        if (side == 0) { a.active_pair[b] = act; a.stop[b] = act ? 0 : 1; a.done_pair[b] = 0; }
    }
}

struct SampleArgs {
    const float* d0; const float* d1;       // descriptor maps of side 0 / 1: [B] x (C, Hd, Wd) with element strides
    long long sb, sc, sh, sw;
    const float* kpx; const int* cnt; float* out;   // out [S][MP][C]
    int C, Hd, Wd, MP; float ds;            // ds = desc_scale
};

// lightglue.py:24-41 sample_descriptors: grid_sample(align_corners=True) at the LightGlue pixel convention, then L2 norm
__global__ __launch_bounds__(256) void lg_sample(SampleArgs a)
{
    const int s = blockIdx.y, b = s >> 1, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= a.cnt[s]) return;
    const float* map = ((s & 1) ? a.d1 : a.d0) + (size_t)b * a.sb;
    const float half = a.ds / 2.0f;
    float kx = (a.kpx[((size_t)s * a.MP + t) * 2] - half) + 0.5f, ky = (a.kpx[((size_t)s * a.MP + t) * 2 + 1] - half) + 0.5f;
    kx = __fdiv_rn(kx, (float)((double)a.Wd * a.ds - a.ds / 2.0 - 0.5)); ky = __fdiv_rn(ky, (float)((double)a.Hd * a.ds - a.ds / 2.0 - 0.5));
    const float gx = kx * 2.0f - 1.0f, gy = ky * 2.0f - 1.0f;
    const float x = (gx + 1.0f) * ((float)(a.Wd - 1) / 2.0f), y = (gy + 1.0f) * ((float)(a.Hd - 1) / 2.0f);
    const float xw = floorf(x), yn = floorf(y);
    const float w = x - xw, e = 1.0f - w, nn = y - yn, so = 1.0f - nn;
    const float c_nw = so * e, c_ne = so * w, c_sw = nn * e, c_se = nn * w;
    const long long x0 = (long long)xw, y0 = (long long)yn, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < a.Wd, vx1 = x1 >= 0 && x1 < a.Wd, vy0 = y0 >= 0 && y0 < a.Hd, vy1 = y1 >= 0 && y1 < a.Hd;
    float v[4];
    float ss = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ch = lane + 64 * q;
        v[q] = 0.0f;
        if (ch < a.C) {
            const float* p = map + (size_t)ch * a.sc;
            const float nw = (vx0 && vy0) ? p[y0 * a.sh + x0 * a.sw] : 0.0f, ne = (vx1 && vy0) ? p[y0 * a.sh + x1 * a.sw] : 0.0f;
            const float sw = (vx0 && vy1) ? p[y1 * a.sh + x0 * a.sw] : 0.0f, se = (vx1 && vy1) ? p[y1 * a.sh + x1 * a.sw] : 0.0f;
            v[q] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(nw, c_nw), __fmul_rn(ne, c_ne)), __fmul_rn(sw, c_sw)), __fmul_rn(se, c_se));
            ss = fmaf(v[q], v[q], ss);
        }
    }
    ss = kpb_wave_sum(ss);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);                      // F.normalize
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (lane + 64 * q < a.C) a.out[((size_t)s * a.MP + t) * a.C + lane + 64 * q] = __fdiv_rn(v[q], nrm);
}

// ------------------------------------------------------------------------------------------------ transformer pieces
// lightglue.py:179-183: qkv.unflatten(-1, (heads, -1, 3)) then rotary embedding of q and k (71-78)
__global__ __launch_bounds__(256) void lg_rotary(const float* qkv, const float* cosb, const float* sinb, float* q, float* k, float* v,
                                                 const int* cnt, const int* active, int MP)
{
    const int s = blockIdx.y;
    if (!active[s]) return;
    const int i = blockIdx.x * 256 + threadIdx.x;        // (token, head, pair f)
    const int t = i >> 7, hf = i & 127, head = hf >> 5, f = hf & 31;
    if (t >= cnt[s]) return;
    const float* src = qkv + ((size_t)s * MP + t) * 768 + head * 192 + (2 * f) * 3;
    const float q0 = src[0], k0 = src[1], v0 = src[2], q1 = src[3], k1 = src[4], v1 = src[5];
    const float c = cosb[((size_t)s * MP + t) * NF + f], sn = sinb[((size_t)s * MP + t) * NF + f];
    const size_t o = ((size_t)s * MP + t) * D + head * HD + 2 * f;
    q[o] = __fadd_rn(__fmul_rn(q0, c), __fmul_rn(-q1, sn)); q[o + 1] = __fadd_rn(__fmul_rn(q1, c), __fmul_rn(q0, sn));
    k[o] = __fadd_rn(__fmul_rn(k0, c), __fmul_rn(-k1, sn)); k[o + 1] = __fadd_rn(__fmul_rn(k1, c), __fmul_rn(k0, sn));
    v[o] = v0; v[o + 1] = v1;
}

struct FlashArgs {
    const float* q; const float* k; const float* v; float* out;   // [S][MP][256] token major, head h at columns 64h..64h+63
    const int* cnt; const int* active;
    int MP, cross; float scale;
};

// softmax(q k^T * scale) v for one head and 32 queries per wave (lightglue.py:142-147 / 226-233), online softmax
__global__ __launch_bounds__(256) void lg_flash(FlashArgs a)
{
    const int s = blockIdx.z, head = blockIdx.y;
    if (!a.active[s]) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, p = lane & 31, h = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wv) * 32;
    const int nq = a.cnt[s];
    if (q0 >= nq) return;
    const int kv = a.cross ? (s ^ 1) : s;
    const int nk = a.cnt[kv];
    float Qr[32];
    {
        const float* qp = a.q + ((size_t)s * a.MP + q0 + p) * D + head * HD + 32 * h;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 t4 = *reinterpret_cast<const float4*>(qp + 4 * c);
            Qr[4 * c] = t4.x; Qr[4 * c + 1] = t4.y; Qr[4 * c + 2] = t4.z; Qr[4 * c + 3] = t4.w;
        }
    }
    f32x16 O0 = {0}, O1 = {0};
    float m_run = -INFINITY, l_run = 0.0f;
    const float* kbase = a.k + (size_t)kv * a.MP * D + head * HD;
    const float* vbase = a.v + (size_t)kv * a.MP * D + head * HD;
    for (int k0 = 0; k0 < nk; k0 += 32) {
        float Kr[32];
        const float* kp = kbase + (size_t)(k0 + p) * D + 32 * h;      // rows past nk stay inside the padded buffer
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 t4 = *reinterpret_cast<const float4*>(kp + 4 * c);
            Kr[4 * c] = t4.x; Kr[4 * c + 1] = t4.y; Kr[4 * c + 2] = t4.z; Kr[4 * c + 3] = t4.w;
        }
        f32x16 st = {0};
#pragma unroll
        for (int c = 0; c < 32; ++c) st = __builtin_amdgcn_mfma_f32_32x32x2f32(Kr[c], Qr[c], st, 0, 0, 0);   // S^T: rows = keys, col = query p
        float sc[16], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            sc[r] = key < nk ? st[r] * a.scale : -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
        mx = kpb_max32(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_run == -INFINITY) ? 0.0f : expf(m_run - m_new);
        float ps = 0.0f, pr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { pr[r] = (sc[r] == -INFINITY) ? 0.0f : expf(sc[r] - m_new); ps += pr[r]; }
        ps = kpb_sum32(ps);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        // rescale O: its rows are queries (r&3) + 8*(r>>2) + 4h, whose alpha lives in the lane of that query
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ar = __shfl(alpha, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
            O0[r] *= ar; O1[r] *= ar;
        }
        // P.V: step r of the K loop consumes key (r&3) + 8*(r>>2) + 4h from this lane half -- exactly where pr[r] sits
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float* vp = vbase + (size_t)key * D + p;
            const float v0 = key < nk ? vp[0] : 0.0f, v1 = key < nk ? vp[32] : 0.0f;
            O0 = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[r], v0, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x2f32(pr[r], v1, O1, 0, 0, 0);
        }
    }
    const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
    float* op = a.out + (size_t)s * a.MP * D + head * HD + p;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qi = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float ir = __shfl(inv, qi, 64);
        if (q0 + qi < nq) {
            op[(size_t)(q0 + qi) * D] = O0[r] * ir;
            op[(size_t)(q0 + qi) * D + 32] = O1[r] * ir;
        }
    }
}

// ---- the same attention on the f16 matrix pipe (split-f16 products, see conv_mfma.h): K and V are first re-laid in MFMA
// operand order, split into (hi, lo) halves, by lg_kv_frags -- once per attention call instead of once per query tile --
// so the flash loop loads every operand with one coalesced 16-byte load per lane and converts nothing but its own P.
//   KF [S][NH][MP/32][kb 4][hi/lo][64 lanes] x 8 halves: lane (p, h) = K[32 blk + p][16 kb + 8 h + j]
//   VF [S][NH][MP/32][nh 2][kb 2][hi/lo][64 lanes] x 8 halves: lane (p, h) = V[key(kb, h, j)][32 nh + p],
//      key(kb, h, j) = 32 blk + 16 kb + 8 (j >> 2) + 4 h + (j & 3): exactly the keys whose probabilities registers
//      8 kb .. 8 kb + 7 of the S^T accumulator hold in lane half h, so P needs no shuffle to become the A operand.
// Keys past the sequence's count are written as zeros (P is zero there too; garbage times zero could be NaN).
// The softmax exponentials use v_exp_f32 (__expf, ~2 ulp): 17 per key block and lane; the match scores stay within the 2e-3 the
// parity tests allow against the reference's fp32 CPU path (1.65 -> 1.46 ms per 16 pairs).
// (Two query tiles per wave, to load each K / V fragment once for 64 queries, need 357 registers: one wave per SIMD, 20 % slower.
// Fetching a key block's fragments once per workgroup through a double-buffered LDS strip, a barrier per block: 13 % slower --
// the L1 is not the bound; the chain S -> softmax -> P -> P V inside a wave is.)
__device__ __forceinline__ void cm_split8(const float* f, cm_h8& hi, cm_h8& lo)
{
    uint2 h0, l0, h1, l1;
    cm_split4(make_float4(f[0], f[1], f[2], f[3]), h0, l0);
    cm_split4(make_float4(f[4], f[5], f[6], f[7]), h1, l1);
    hi = __builtin_bit_cast(cm_h8, make_uint4(h0.x, h0.y, h1.x, h1.y));
    lo = __builtin_bit_cast(cm_h8, make_uint4(l0.x, l0.y, l1.x, l1.y));
}

// Operand range (r03, conv_mfma.h cm_scale_of): every 32-key block of K and of V is split at the power-of-two scale that fits the
// block's own largest magnitude; the two exponents go to `kvexp` and lg_flash_h takes the scales out again (K: in the factor
// the scores are multiplied by anyway; V: see there).  Q is scaled per query tile inside lg_flash_h.
struct FragArgs {
    const float* k; const float* v; uint4* kf; uint4* vf;
    int* kvexp;         // [S][NH][MP/32][2]: cm_exp_of(amax) of the K block and of the V block
    const int* cnt; const int* active;
    int MP, cross;
    int f16only;        // attention = "f16" (kpb_lg_set_attention): the operands are the plain half-precision casts x.half() of
                        // lightglue.py:131 -- scale 1 (exponent 141: cm_scale_of = cm_unscale_of = 1), the lo halves unused
};

__global__ __launch_bounds__(256) void lg_kv_frags(FragArgs a)
{
    __shared__ float s_am[2][4];
    const int s = blockIdx.z, head = blockIdx.y, blk = blockIdx.x;
    // the fragments of sequence s are read by the queries of sequence s (self) or s ^ 1 (cross)
    if (!a.active[a.cross ? (s ^ 1) : s]) return;
    const int n = a.cnt[s];
    if (blk * 32 >= n) return;
    const int t = threadIdx.x, lane = t & 63, p = lane & 31, h = lane >> 5, NB = a.MP / 32;
    const size_t fb = (((size_t)s * NH + head) * NB + blk) * 8 * 64;
    float fk[8], fv[8], mk = 0.0f, mv = 0.0f;
    const int kbk = t >> 6, nh = (t >> 6) & 1, kbv = t >> 7;
    {
        const int key = 32 * blk + p;
        const float* src = a.k + ((size_t)s * a.MP + key) * D + head * HD + 16 * kbk + 8 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) { fk[j] = key < n ? src[j] : 0.0f; mk = fmaxf(mk, fabsf(fk[j])); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kv = 32 * blk + 16 * kbv + 8 * (j >> 2) + 4 * h + (j & 3);
            fv[j] = kv < n ? a.v[((size_t)s * a.MP + kv) * D + head * HD + 32 * nh + p] : 0.0f;
            mv = fmaxf(mv, fabsf(fv[j]));
        }
    }
    mk = cm_wave_max(mk); mv = cm_wave_max(mv);
    if (lane == 0) { s_am[0][t >> 6] = mk; s_am[1][t >> 6] = mv; }
    __syncthreads();
    const int ek = a.f16only ? 141 : cm_exp_of(fmaxf(fmaxf(s_am[0][0], s_am[0][1]), fmaxf(s_am[0][2], s_am[0][3])));
    const int ev = a.f16only ? 141 : cm_exp_of(fmaxf(fmaxf(s_am[1][0], s_am[1][1]), fmaxf(s_am[1][2], s_am[1][3])));
    if (t == 0) { int* e = a.kvexp + (((size_t)s * NH + head) * NB + blk) * 2; e[0] = ek; e[1] = ev; }
    const float sk = cm_scale_of(ek), sv = cm_scale_of(ev);
#pragma unroll
    for (int j = 0; j < 8; ++j) { fk[j] *= sk; fv[j] *= sv; }
    cm_h8 hi, lo;
    cm_split8(fk, hi, lo);
    a.kf[fb + (kbk * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    a.kf[fb + (kbk * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
    cm_split8(fv, hi, lo);
    a.vf[fb + ((nh * 2 + kbv) * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    a.vf[fb + ((nh * 2 + kbv) * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
}

struct FlashHArgs {
    const float* q; const uint4* kf; const uint4* vf; const int* kvexp; float* out;
    const int* cnt; const int* active;
    int MP, cross; float scale;
};

// F16 (r05, VERDICT r04 missing 3): the arithmetic the REFERENCE runs on a GPU -- lightglue.py:129-134 hands q.half(), k.half(), v.half() to
// scaled_dot_product_attention and casts the half-precision result back: Q, K, V as plain f16 casts (no scale, no lo halves: ONE MFMA
// per product instead of three), scores and softmax in fp32, the probabilities cast to f16 for P.V with fp32 accumulation, the
// output rounded to f16.  Opt-in (kpb_lg_set_attention / LightGlue(attention="f16")), reported separately; the default stays the
// fp32-equivalent split form, which is what the reference's CPU path -- the parity contract -- computes.
template <bool F16 = false>
__global__ __launch_bounds__(256) void lg_flash_h(FlashHArgs a)
{
    const int s = blockIdx.z, head = blockIdx.y;
    if (!a.active[s]) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, p = lane & 31, h = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wv) * 32;
    const int nq = a.cnt[s];
    if (q0 >= nq) return;
    const int kv = a.cross ? (s ^ 1) : s;
    const int nk = a.cnt[kv], NB = a.MP / 32;
    cm_h8 Qh[4], Ql[4];
    float qscale;           // a.scale / (the tile's Q scale): with the key block's K scale, what turns the accumulator into scores
    {
        // rows past nq are inside the padded buffer but hold whatever an earlier call left there: they enter as zeros, so the
        // tile's operand scale depends on its own queries only
        const float* qp = a.q + ((size_t)s * a.MP + q0 + p) * D + head * HD + 8 * h;
        const bool qvalid = q0 + p < nq;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 t0[4], t1[4];
        float am = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            t0[kb] = qvalid ? *reinterpret_cast<const float4*>(qp + 16 * kb) : z4; t1[kb] = qvalid ? *reinterpret_cast<const float4*>(qp + 16 * kb + 4) : z4;
            am = cm_amax4(cm_amax4(am, t0[kb]), t1[kb]);
        }
        const int eq = F16 ? 141 : cm_exp_of(cm_wave_max(am));
        const float sq = cm_scale_of(eq);
        qscale = a.scale * cm_unscale_of(eq);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const float f[8] = {t0[kb].x * sq, t0[kb].y * sq, t0[kb].z * sq, t0[kb].w * sq, t1[kb].x * sq, t1[kb].y * sq, t1[kb].z * sq, t1[kb].w * sq};
            cm_split8(f, Qh[kb], Ql[kb]);
        }
    }
    f32x16 O0 = {0}, O1 = {0};
    float m_run = -INFINITY, l_run = 0.0f;
    const uint4* kfb = a.kf + ((size_t)kv * NH + head) * NB * 8 * 64 + lane;
    const uint4* vfb = a.vf + ((size_t)kv * NH + head) * NB * 8 * 64 + lane;
    const int* kve = a.kvexp + ((size_t)kv * NH + head) * NB * 2;
    int ev_run = 0;         // exponent whose V scale the output accumulators carry (wave-uniform); 0: nothing accumulated yet
    for (int k0 = 0; k0 < nk; k0 += 32) {
        const uint4* kf = kfb + (size_t)(k0 >> 5) * 8 * 64;
        const uint4* vf = vfb + (size_t)(k0 >> 5) * 8 * 64;
        const int ek = __builtin_amdgcn_readfirstlane(kve[2 * (k0 >> 5)]), ev = __builtin_amdgcn_readfirstlane(kve[2 * (k0 >> 5) + 1]);
        const float kscale = qscale * cm_unscale_of(ek);
        cm_h8 Kh[4], Kl[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            Kh[kb] = __builtin_bit_cast(cm_h8, kf[(kb * 2 + 0) * 64]);
            Kl[kb] = __builtin_bit_cast(cm_h8, kf[(kb * 2 + 1) * 64]);
        }
        // V of this block goes out now too: it lands while the scores and the softmax are worked on
        cm_h8 Vh[2][2], Vl[2][2];
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                Vh[nh][kb] = __builtin_bit_cast(cm_h8, vf[((nh * 2 + kb) * 2 + 0) * 64]);
                Vl[nh][kb] = __builtin_bit_cast(cm_h8, vf[((nh * 2 + kb) * 2 + 1) * 64]);
            }
        f32x16 st = {0};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {       // S^T: rows = keys, col = query p
            if constexpr (!F16) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(Kl[kb], Qh[kb], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(Kh[kb], Ql[kb], st, 0, 0, 0);
            }
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(Kh[kb], Qh[kb], st, 0, 0, 0);
        }
        float sc[16], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            sc[r] = key < nk ? st[r] * kscale : -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
        mx = kpb_max32(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_run == -INFINITY) ? 0.0f : __expf(m_run - m_new);
        float ps = 0.0f, pr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { pr[r] = (sc[r] == -INFINITY) ? 0.0f : __expf(sc[r] - m_new); ps += pr[r]; }
        ps = kpb_sum32(ps);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (__ballot(alpha != 1.0f)) {      // once the running maxima have settled every alpha is exactly 1: skip the 16 lane exchanges
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ar = __shfl(alpha, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
                O0[r] *= ar; O1[r] *= ar;
            }
        }
        // V blocks come at their own scales: the accumulators keep the scale of the LARGEST block so far (a larger one brings
        // them down to it, exactly), a smaller block's probabilities are brought down instead (its products are that much
        // smaller than what is already there).  The probabilities themselves are scaled by 2^13 for the split (P <= 1).
        if (ev > ev_run) {
            if (ev_run) {
                const float dn = cm_pow2(max(ev_run - ev, -126));
#pragma unroll
                for (int r = 0; r < 16; ++r) { O0[r] *= dn; O1[r] *= dn; }
            }
            ev_run = ev;
        }
        const float pfac = F16 ? 1.0f : cm_pow2(max(ev - ev_run, -100) + 13);      // F16: p.half(), unscaled, as the reference's kernel casts it
#pragma unroll
        for (int r = 0; r < 16; ++r) pr[r] *= pfac;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {       // registers 8 kb .. 8 kb + 7 are this lane half's keys of k-block kb
            cm_h8 Ph, Pl;
            cm_split8(pr + 8 * kb, Ph, Pl);
            if constexpr (!F16) {
                O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Pl, Vh[0][kb], O0, 0, 0, 0);
                O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ph, Vl[0][kb], O0, 0, 0, 0);
            }
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ph, Vh[0][kb], O0, 0, 0, 0);
            if constexpr (!F16) {
                O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Pl, Vh[1][kb], O1, 0, 0, 0);
                O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ph, Vl[1][kb], O1, 0, 0, 0);
            }
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ph, Vh[1][kb], O1, 0, 0, 0);
        }
    }
    const float inv = l_run > 0.0f ? (1.0f / l_run) * cm_unscale_of(max(ev_run, 24)) * (F16 ? 1.0f : cm_pow2(-13)) : 0.0f;
    float* op = a.out + (size_t)s * a.MP * D + head * HD + p;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qi = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float ir = __shfl(inv, qi, 64);
        if (q0 + qi < nq) {
            float o0 = O0[r] * ir, o1 = O1[r] * ir;
            if constexpr (F16) { o0 = (float)(_Float16)o0; o1 = (float)(_Float16)o1; }      // the half-precision result, .to(q.dtype)
            op[(size_t)(q0 + qi) * D] = o0;
            op[(size_t)(q0 + qi) * D + 32] = o1;
        }
    }
}

// ffn.1 + ffn.2: LayerNorm(512, eps 1e-5, affine) then exact GELU, in place (lightglue.py:166-170)
__global__ __launch_bounds__(256) void lg_ln_gelu(float* hbuf, const float* g, const float* bta, const int* cnt, const int* active, int MP)
{
    const int s = blockIdx.y, lane = threadIdx.x & 63;
    if (!active[s]) return;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= cnt[s]) return;
    float* x = hbuf + ((size_t)s * MP + t) * 512;
    float v[8], sum = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { v[q] = x[lane + 64 * q]; sum += v[q]; }
    sum = kpb_wave_sum(sum);
    const float mean = sum / 512.0f;
    float var = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float d = v[q] - mean; var = fmaf(d, d, var); }
    var = kpb_wave_sum(var);
    const float rstd = 1.0f / sqrtf(var / 512.0f + 1e-5f);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = lane + 64 * q;
        const float y = (v[q] - mean) * rstd * g[c] + bta[c];
        x[c] = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
    }
}

// x += y on the first 256 columns of the 512-wide [x | message] rows
__global__ void lg_residual(float* cat, const float* y, const int* cnt, const int* active, int MP)
{
    const int s = blockIdx.y;
    if (!active[s]) return;
    const int i = blockIdx.x * 256 + threadIdx.x, t = i >> 6, c4 = (i & 63) * 4;
    if (t >= cnt[s]) return;
    float4* x = reinterpret_cast<float4*>(cat + ((size_t)s * MP + t) * 512 + c4);
    const float4 d = *reinterpret_cast<const float4*>(y + ((size_t)s * MP + t) * D + c4);
    float4 v = *x;
    v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
    *x = v;
}

// token confidence (lightglue.py:101-111) and matchability (308-309): two 256 -> 1 linears per token
__global__ __launch_bounds__(256) void lg_conf(const float* cat, const float* wc, const float* bc, const float* wm, const float* bm,
                                               float* conf, float* msc, float* zlog, const int* cnt, const int* active, int MP, int has_conf)
{
    const int s = blockIdx.y, lane = threadIdx.x & 63;
    if (!active[s]) return;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= cnt[s]) return;
    const float4 x = *reinterpret_cast<const float4*>(cat + ((size_t)s * MP + t) * 512 + 4 * lane);
    const float4 m = *reinterpret_cast<const float4*>(wm + 4 * lane);
    float zc = 0.0f, zm = fmaf(x.x, m.x, fmaf(x.y, m.y, fmaf(x.z, m.z, x.w * m.w)));
    if (has_conf) {
        const float4 c = *reinterpret_cast<const float4*>(wc + 4 * lane);
        zc = fmaf(x.x, c.x, fmaf(x.y, c.y, fmaf(x.z, c.z, x.w * c.w)));
    }
    zc = kpb_wave_sum(zc); zm = kpb_wave_sum(zm);
    if (lane == 0) {
        const size_t o = (size_t)s * MP + t;
        zm += bm[0];
        zlog[o] = zm;
        msc[o] = 1.0f / (1.0f + expf(-zm));
        conf[o] = has_conf ? 1.0f / (1.0f + expf(-(zc + bc[0]))) : 0.0f;
    }
}

struct DecideArgs {
    const float* conf; const float* msc;
    int* cnt; const int* cnt_orig; int* newcnt; int* dst;   // dst [S][MP]: new row of a kept token, -1 if pruned
    int* active_pair; int* active_seq; int* fin_pair; int* fin_seq; int* stop; int* done_pair;
    int MP, layer, prune_min; float thr, depth_conf, width_keep;   // width_keep = 1 - width_confidence
    int do_stop, do_prune;
};

// check_if_stop (lightglue.py:670-681) and get_pruning_mask (659-668), one workgroup per pair
__global__ __launch_bounds__(256) void lg_decide(DecideArgs a)
{
    __shared__ int wsum[4];
    __shared__ int s_flag;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) { a.fin_pair[b] = 0; a.fin_seq[2 * b] = 0; a.fin_seq[2 * b + 1] = 0; }
    if (!a.active_pair[b]) return;
    const int s0 = 2 * b, s1 = 2 * b + 1;
    const bool last = a.layer == NL - 1;
    bool finish = last;
    if (!last && a.do_stop) {
        int unconf = 0;
        for (int side = 0; side < 2; ++side) {
            const int s = s0 + side, n = a.cnt[s];
            for (int t = tid; t < n; t += 256) unconf += a.conf[(size_t)s * a.MP + t] < a.thr;
        }
        unconf = kpb_wave_sum(unconf);
        if (lane == 0) wsum[wid] = unconf;
        __syncthreads();
        if (tid == 0) {
            const float tot = (float)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
            const float ratio = 1.0f - __fdiv_rn(tot, (float)(a.cnt_orig[s0] + a.cnt_orig[s1]));
            s_flag = ratio > a.depth_conf;
        }
        __syncthreads();
        finish = s_flag != 0;
        __syncthreads();
    }
    if (finish) {
        if (tid == 0) {
            a.fin_pair[b] = 1; a.fin_seq[s0] = 1; a.fin_seq[s1] = 1;
            a.stop[b] = a.layer + 1;
            a.done_pair[b] = 1;
            a.active_pair[b] = 0; a.active_seq[s0] = 0; a.active_seq[s1] = 0;
            a.newcnt[s0] = a.cnt[s0]; a.newcnt[s1] = a.cnt[s1];
        }
        return;
    }
    // pruning: ordered compaction of the kept tokens (index_select keeps order, lightglue.py:568-571)
    for (int side = 0; side < 2; ++side) {
        const int s = s0 + side, n = a.cnt[s];
        const bool prune = a.do_prune && n > a.prune_min;
        int base = 0;
        for (int c0 = 0; c0 < n; c0 += 256) {
            const int t = c0 + tid;
            bool keep = false;
            if (t < n) {
                const size_t o = (size_t)s * a.MP + t;
                keep = !prune || (a.msc[o] > a.width_keep) || (a.do_stop && a.conf[o] <= a.thr);
            }
            const unsigned long long bal = __ballot(keep);
            const int within = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wsum[wid] = __popcll(bal);
            __syncthreads();
            int wbase = 0, tot = 0;
            for (int w = 0; w < 4; ++w) { if (w < wid) wbase += wsum[w]; tot += wsum[w]; }
            __syncthreads();
            if (t < n) a.dst[(size_t)s * a.MP + t] = keep ? base + wbase + within : -1;
            base += tot;
        }
        if (tid == 0) a.newcnt[s] = base;
    }
}

// moves kept tokens (x, encoding, original index) to their new rows in the other buffer set
__global__ void lg_gather(const float* cat_in, float* cat_out, const float* cos_in, const float* sin_in, float* cos_out, float* sin_out,
                          const int* ind_in, int* ind_out, const int* dst, const int* cnt, const int* active, int MP)
{
    const int s = blockIdx.y;
    if (!active[s]) return;
    const int i = blockIdx.x * 256 + threadIdx.x, t = i >> 6, c4 = (i & 63) * 4;
    if (t >= cnt[s]) return;
    const int d = dst[(size_t)s * MP + t];
    if (d < 0) return;
    *reinterpret_cast<float4*>(cat_out + ((size_t)s * MP + d) * 512 + c4) = *reinterpret_cast<const float4*>(cat_in + ((size_t)s * MP + t) * 512 + c4);
    if (c4 < NF) {
        *reinterpret_cast<float4*>(cos_out + ((size_t)s * MP + d) * NF + c4) = *reinterpret_cast<const float4*>(cos_in + ((size_t)s * MP + t) * NF + c4);
        *reinterpret_cast<float4*>(sin_out + ((size_t)s * MP + d) * NF + c4) = *reinterpret_cast<const float4*>(sin_in + ((size_t)s * MP + t) * NF + c4);
    }
    if (c4 == 0) ind_out[(size_t)s * MP + d] = ind_in[(size_t)s * MP + t];
}

__global__ void lg_commit_counts(int* cnt, const int* newcnt, const int* active, int S)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s < S && active[s]) cnt[s] = newcnt[s];
}

// ------------------------------------------------------------------------------------------------ assignment
// sim[b][i][j] = (md0[i] / 4) . (md1[j] / 4), md = final_proj(desc) (lightglue.py:299-303); 32x32 tile per wave
__global__ __launch_bounds__(256) void lg_sim(const float* md, float* sim, const int* cnt, const int* fin_pair, int MP)
{
    const int b = blockIdx.z;
    if (!fin_pair[b]) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, p = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.y * 32, j0 = (blockIdx.x * 4 + wv) * 32;
    const int m = cnt[2 * b], n = cnt[2 * b + 1];
    if (i0 >= m || j0 >= n) return;
    const float* A = md + ((size_t)(2 * b) * MP + i0 + p) * D + 32 * h;
    const float* Bm = md + ((size_t)(2 * b + 1) * MP + j0 + p) * D + 32 * h;
    f32x16 acc = {0};
    // r06: the next half-chunk's rows (sixteen channels of A and of B per lane) are requested before this half-chunk's sixteen MFMAs, the order pinned: it was four
    // chunks of 32 channels per lane, each one's sixteen loads waited for where they were issued (231 us per 16 pairs of 1 000 x 1 000 at 42 registers)
    float4 xa[2][4], xb[2][4];
    auto fetch = [&](int buf, int hc) {          // half-chunk hc = 0..7: channels 64 (hc / 2) + 32 h + 16 (hc & 1) .. + 15 of this lane's rows
        const int off = (hc >> 1) * 64 + (hc & 1) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) { xa[buf][q] = *reinterpret_cast<const float4*>(A + off + 4 * q); xb[buf][q] = *reinterpret_cast<const float4*>(Bm + off + 4 * q); }
    };
    auto mm = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x = xa[buf][q], y = xb[buf][q];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x * 0.25f, y.x * 0.25f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y * 0.25f, y.y * 0.25f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z * 0.25f, y.z * 0.25f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w * 0.25f, y.w * 0.25f, acc, 0, 0, 0);
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int hc = 0; hc < 8; ++hc) {
        if (hc + 1 < 8) fetch((hc + 1) & 1, hc + 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(hc & 1);
    }
    float* o = sim + (size_t)b * MP * MP;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (i < m && j0 + p < n) o[(size_t)i * MP + j0 + p] = acc[r];
    }
}

__device__ __forceinline__ float logsigmoid(float z) { return fminf(z, 0.0f) - log1pf(expf(-fabsf(z))); }

// max and log(sum(exp(x - max))) of every ROW of sim (one half of log_softmax): a wave per row, lanes along the row.  Also logsigmoid(z) of the row's
// token of image 0 and (lines < n) of the same-numbered token of image 1, once per token (r06: lg_best evaluated both per ELEMENT: an exp and a log1p
// for each of the 16 M cells, most of its 138 us).
constexpr int LG_CROWS = 32;       // rows per workgroup of the column kernels

__device__ __forceinline__ void lg_lse_rows(const float* sim, float* mxo, float* lgo, const float* zlog, float* lso, const int* cnt, const int* fin_pair, int MP, int bx)
{
    const int b = blockIdx.y;
    if (!fin_pair[b]) return;
    const int lane = threadIdx.x & 63;
    const int m = cnt[2 * b], n = cnt[2 * b + 1];
    const int line = bx * 4 + (threadIdx.x >> 6);
    if (lane == 0 && line < n) lso[(size_t)(2 * b + 1) * MP + line] = logsigmoid(zlog[(size_t)(2 * b + 1) * MP + line]);
    if (line >= m) return;
    const float* base = sim + (size_t)b * MP * MP + (size_t)line * MP;
    float mx = -INFINITY;
    for (int k = lane; k < n; k += 64) mx = fmaxf(mx, base[k]);
    mx = kpb_wave_fmax(mx);
    float sum = 0.0f;
    for (int k = lane; k < n; k += 64) sum += expf(base[k] - mx);
    sum = kpb_wave_sum(sum);
    if (lane == 0) {
        mxo[(size_t)(2 * b) * MP + line] = mx; lgo[(size_t)(2 * b) * MP + line] = logf(sum);
        lso[(size_t)(2 * b) * MP + line] = logsigmoid(zlog[(size_t)(2 * b) * MP + line]);
    }
}

// The same for every COLUMN, in two steps (r06; r03 .. r05 gave a column to a wave, its lanes a row pitch apart: 64 lines per load, 191 us per 16 pairs).
// Step 1: a workgroup takes LG_CROWS rows x 256 columns, a thread one column: coalesced loads, all of a thread's rows in flight; it leaves the
// column's (max, sum exp(x - max)) over those rows.  Step 2 folds a column's partials.
__device__ __forceinline__ void lg_lse_cols(const float* sim, float* pmx, float* psum, const int* cnt, const int* fin_pair, int MP, int NCH, int bx, int chunk)
{
    const int b = blockIdx.y;
    if (!fin_pair[b]) return;
    const int m = cnt[2 * b], n = cnt[2 * b + 1];
    const int j = bx * 256 + threadIdx.x, i0 = chunk * LG_CROWS;
    if (i0 >= m || j >= n) return;
    const float* base = sim + (size_t)b * MP * MP + (size_t)i0 * MP + j;
    float x[LG_CROWS];
#pragma unroll
    for (int r = 0; r < LG_CROWS; ++r) x[r] = i0 + r < m ? base[(size_t)r * MP] : -INFINITY;
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < LG_CROWS; ++r) mx = fmaxf(mx, x[r]);
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < LG_CROWS; ++r) sum += i0 + r < m ? expf(x[r] - mx) : 0.0f;
    pmx[((size_t)b * NCH + chunk) * MP + j] = mx;
    psum[((size_t)b * NCH + chunk) * MP + j] = sum;
}

// one launch for both: workgroups [0, nrow) take four rows each, the rest a (256-column group, row chunk) tile (grid: nrow + ncolgrp * NCH, pairs)
__global__ __launch_bounds__(256) void lg_lse(const float* sim, float* mxo, float* lgo, const float* zlog, float* lso, float* pmx, float* psum, const int* cnt,
                                              const int* fin_pair, int MP, int NCH, int nrow)
{
    const int bx = blockIdx.x;
    if (bx < nrow) lg_lse_rows(sim, mxo, lgo, zlog, lso, cnt, fin_pair, MP, bx);
    else lg_lse_cols(sim, pmx, psum, cnt, fin_pair, MP, NCH, (bx - nrow) / NCH, (bx - nrow) % NCH);
}

__global__ __launch_bounds__(256) void lg_lse_colfin(const float* pmx, const float* psum, float* mxo, float* lgo, const int* cnt, const int* fin_pair, int MP, int NCH)
{
    const int b = blockIdx.y;
    if (!fin_pair[b]) return;
    const int m = cnt[2 * b], n = cnt[2 * b + 1], j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int nch = (m + LG_CROWS - 1) / LG_CROWS;
    float mx = -INFINITY;
    for (int c = 0; c < nch; ++c) mx = fmaxf(mx, pmx[((size_t)b * NCH + c) * MP + j]);
    float sum = 0.0f;
    for (int c = 0; c < nch; ++c) sum += psum[((size_t)b * NCH + c) * MP + j] * expf(pmx[((size_t)b * NCH + c) * MP + j] - mx);
    mxo[(size_t)(2 * b + 1) * MP + j] = mx;
    lgo[(size_t)(2 * b + 1) * MP + j] = logf(sum);
}

// scores = log_softmax(sim, rows) + log_softmax(sim, cols) + logsigmoid(z0) + logsigmoid(z1)^T (lightglue.py:278-290);
// best column of every row, first index on ties (filter_matches 317): a wave per row
__device__ __forceinline__ void lg_best_rows(const float* sim, const float* mxo, const float* lgo, const float* lso, float* bestv, int* besti,
                                             const int* cnt, const int* fin_pair, int MP, int bx)
{
    const int b = blockIdx.y;
    if (!fin_pair[b]) return;
    const int lane = threadIdx.x & 63;
    const int m = cnt[2 * b], n = cnt[2 * b + 1];
    const int i = bx * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    const float* S = sim + (size_t)b * MP * MP + (size_t)i * MP;
    const size_t r0 = (size_t)(2 * b) * MP, r1 = (size_t)(2 * b + 1) * MP;
    const float mi = mxo[r0 + i], li = lgo[r0 + i], si = lso[r0 + i];
    float bv = -INFINITY; int bi = 0x7FFFFFFF;
    for (int j = lane; j < n; j += 64) {
        const float x = S[j];
        const float s0 = (x - mi) - li, s1 = (x - mxo[r1 + j]) - lgo[r1 + j];
        const float v = (s0 + s1) + (si + lso[r1 + j]);
        if (v > bv) { bv = v; bi = j; }
    }
    kpb_butterfly([&](auto o) {
        constexpr int O = decltype(o)::value;
        const float ov = kpb_shfl_xor<O>(bv); const int oi = kpb_shfl_xor<O>(bi);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    });
    if (lane == 0) { bestv[r0 + i] = bv; besti[r0 + i] = bi; }
}

// best ROW of every column, first index on ties: the tiling of lg_lse_col; a column's partial (value, row) per row chunk, folded by lg_emit where it is needed
__device__ __forceinline__ void lg_best_cols(const float* sim, const float* mxo, const float* lgo, const float* lso, float* pbv, int* pbi,
                                             const int* cnt, const int* fin_pair, int MP, int NCH, int bx, int chunk)
{
    const int b = blockIdx.y;
    if (!fin_pair[b]) return;
    const int m = cnt[2 * b], n = cnt[2 * b + 1];
    const int j = bx * 256 + threadIdx.x, i0 = chunk * LG_CROWS;
    if (i0 >= m || j >= n) return;
    const float* base = sim + (size_t)b * MP * MP + (size_t)i0 * MP + j;
    const size_t r0 = (size_t)(2 * b) * MP, r1 = (size_t)(2 * b + 1) * MP;
    float x[LG_CROWS];
#pragma unroll
    for (int r = 0; r < LG_CROWS; ++r) x[r] = i0 + r < m ? base[(size_t)r * MP] : 0.0f;
    const float mj = mxo[r1 + j], lj = lgo[r1 + j], sj = lso[r1 + j];
    float bv = -INFINITY; int bi = 0x7FFFFFFF;
#pragma unroll
    for (int r = 0; r < LG_CROWS; ++r) {
        const int i = min(i0 + r, m - 1);               // (wave-uniform: scalar loads)
        const float s0 = (x[r] - mxo[r0 + i]) - lgo[r0 + i], s1 = (x[r] - mj) - lj;
        const float v = (s0 + s1) + (lso[r0 + i] + sj);
        if (i0 + r < m && v > bv) { bv = v; bi = i0 + r; }
    }
    pbv[((size_t)b * NCH + chunk) * MP + j] = bv;
    pbi[((size_t)b * NCH + chunk) * MP + j] = bi;
}

__global__ __launch_bounds__(256) void lg_best(const float* sim, const float* mxo, const float* lgo, const float* lso, float* bestv, int* besti, float* pbv, int* pbi,
                                               const int* cnt, const int* fin_pair, int MP, int NCH, int nrow)
{
    const int bx = blockIdx.x;
    if (bx < nrow) lg_best_rows(sim, mxo, lgo, lso, bestv, besti, cnt, fin_pair, MP, bx);
    else lg_best_cols(sim, mxo, lgo, lso, pbv, pbi, cnt, fin_pair, MP, NCH, (bx - nrow) / NCH, (bx - nrow) % NCH);
}

// filter_matches (lightglue.py:315-331) + the index mapping of pruned points (616-623): ordered by the row index
__global__ __launch_bounds__(256) void lg_emit(const float* bestv, const int* besti, const float* pbv, const int* pbi, int NCH, const int* ind0, const int* ind1,
                                               const int* stop, const int* cnt, const int* fin_pair, int* out_pairs, float* out_scores, int* out_k, int MP, int K, float th)
{
    __shared__ int wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (!fin_pair[b]) return;
    const int* ind = ((stop[b] - 1) & 1) ? ind1 : ind0;      // the index map of the layer the pair finished at (layer stop - 1 used buffer (stop - 1) & 1)
    const int m = cnt[2 * b];
    const size_t r0 = (size_t)(2 * b) * MP, r1 = (size_t)(2 * b + 1) * MP;
    int base = 0;
    for (int c0 = 0; c0 < m; c0 += 256) {
        const int i = c0 + tid;
        bool valid = false; int j = 0; float ms = 0.0f;
        if (i < m) {
            j = besti[r0 + i];
            ms = expf(bestv[r0 + i]);
            // the best row of column j: the best of its row chunks' partials, the first chunk on ties (each partial is its chunk's first row on ties)
            const int nch = (m + LG_CROWS - 1) / LG_CROWS;
            float cv = -INFINITY; int ci = 0x7FFFFFFF;
            if (j < MP) {       // (no column won: bi = 0x7FFFFFFF, e.g. a row of NaNs)
                int cc = -1;
                for (int q0 = 0; q0 < nch; q0 += 16) {          // sixteen partials in flight, then the comparisons
                    float pv[16];
#pragma unroll
                    for (int c = 0; c < 16; ++c) pv[c] = q0 + c < nch ? pbv[((size_t)b * NCH + q0 + c) * MP + j] : -INFINITY;
#pragma unroll
                    for (int c = 0; c < 16; ++c) if (pv[c] > cv) { cv = pv[c]; cc = q0 + c; }
                }
                if (cc >= 0) ci = pbi[((size_t)b * NCH + cc) * MP + j];
            }
            valid = (ci == i) && (ms > th);
        }
        const unsigned long long bal = __ballot(valid);
        const int within = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wid] = __popcll(bal);
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < 4; ++w) { if (w < wid) wbase += wsum[w]; tot += wsum[w]; }
        __syncthreads();
        if (valid) {
            const int pos = base + wbase + within;
            out_pairs[((size_t)b * K + pos) * 2] = ind[r0 + i];
            out_pairs[((size_t)b * K + pos) * 2 + 1] = ind[r1 + j];
            out_scores[(size_t)b * K + pos] = ms;
        }
        base += tot;
    }
    if (tid == 0) out_k[b] = base;
}

__global__ void lg_init_out(int* out_k, int* out_stop, const int* stop, int B, int final_pass)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    if (!final_pass) out_k[b] = 0;
    else out_stop[b] = stop[b];
}

}  // namespace

// ================================================================================================ host side
struct kpb_lg {
    kpb_ctx* ctx = nullptr;
    int input_dim = 256;
    float desc_scale = 8.0f;
    float* wdev = nullptr;
    std::map<std::string, size_t> off;
    std::map<std::string, float> wscale;   // split-f16 packs: the power-of-two weight scale of each Linear
    kpb_buf ws;
    int attn_f16 = 0;                      // kpb_lg_set_attention: 1 = the reference's GPU arithmetic (lg_flash_h<true>)
    float* wp(const std::string& n) { return wdev + off.at(n); }
};

namespace {

struct LgNetShim : kpb_net {     // WeightStage::upload wants a kpb_net; only ctx / wdev / off are used
    int forward(const float*, int, int, int, float*, float*) override { return KPB_E_INVALID; }
};

// One Linear over the tokens of every sequence.  Split-f16 form: gemm_h, rows past cnt[s] neither read nor written; `epi` selects
// the fused epilogue (conv_mfma.h GE_*) and `x` carries its operands.  Strict fp32 form: conv_mfma<1,...>, plain epilogue only.
struct LgEpi {
    int epi = GE_PLAIN;
    const float* res = nullptr; int rstride = 0;          // GE_RESIDUAL
    const float* cosb = nullptr; const float* sinb = nullptr; float* out1 = nullptr; float* out2 = nullptr;   // GE_ROTARY (out1 also GE_SPLIT2)
};

int lg_linear(kpb_ctx* ctx, kpb_lg* lg, const char* tag, const std::string& name, int cin, int cout, const float* in, int istride,
              float* out, int ostride, int ooff, int S, int MP, const int* active, const int* cnt, const LgEpi& x = LgEpi())
{
    ConvM a;
    a.in = in; a.out = out; a.wp = lg->wp(name + ".w"); a.bias = lg->wp(name + ".b"); a.xf = nullptr; a.res = nullptr; a.active = active;
    a.Hi = MP / 16; a.Wi = 16; a.H = MP / 16; a.W = 16;
    a.CIN = cin; a.COUT = cout; a.NCH = cin / 32; a.relu = 0; a.nblk = (cout + 63) / 64;
    a.istride = istride; a.ostride = ostride; a.ooff = ooff;
    if (conv_mfma_use_h16()) {
        a.unscale = 1.0f / (lg->wscale.at(name + ".w"));
        a.rowcnt = cnt; a.res = x.res; a.rstride = x.rstride; a.aux0 = x.cosb; a.aux1 = x.sinb; a.out1 = x.out1; a.out2 = x.out2;
        // 128-row tiles: four workgroups per CU (8-15 % faster than 256-row tiles, r02)
        const dim3 grid(cdiv(MP, 128), 1, S * a.nblk);
        if (x.epi == GE_RESIDUAL) KPB_LAUNCH(ctx, tag, (gemm_h<2, 1, GE_RESIDUAL>), grid, dim3(256), 0, ctx->stream, a);
        else if (x.epi == GE_ROTARY) KPB_LAUNCH(ctx, tag, (gemm_h<2, 1, GE_ROTARY>), grid, dim3(256), 0, ctx->stream, a);
        else if (x.epi == GE_SPLIT2) KPB_LAUNCH(ctx, tag, (gemm_h<2, 1, GE_SPLIT2>), grid, dim3(256), 0, ctx->stream, a);
        else KPB_LAUNCH(ctx, tag, (gemm_h<2, 1>), grid, dim3(256), 0, ctx->stream, a);
    } else {
        if (x.epi != GE_PLAIN) return kpb_fail(ctx, KPB_E_INVALID, "lg_linear: fused epilogues exist in the split-f16 form only");
        KPB_LAUNCH(ctx, tag, (conv_mfma<1, 1, 32, false, false, false, 2>), dim3(1, MP / 128, S * a.nblk), dim3(256), 0, ctx->stream, a);
    }
    return KPB_OK;
}

// Row order of the fused Wqkv product (GE_ROTARY): packed row (2 nb + par) 32 + f, nb = 3 head + which, is the reference's row
// head 192 + (2 f + par) 3 + which of qkv.unflatten(-1, (heads, -1, 3)) (lightglue.py:179-180).
inline int lg_qkv_row(int o)
{
    const int nb = o / 64, par = (o % 64) / 32, f = o % 32, head = nb / 3, which = nb % 3;
    return head * 192 + (2 * f + par) * 3 + which;
}

}  // namespace

#define KPB_API extern "C" __attribute__((visibility("default")))

KPB_API int kpb_lg_create(kpb_ctx* ctx, const void* blob, size_t len, float desc_scale, kpb_lg** out)
{
    if (!ctx || !blob || !out) return kpb_fail(ctx, KPB_E_INVALID, "kpb_lg_create: null argument");
    *out = nullptr;
    KpbwBlob bl;
    if (!bl.parse(blob, len) || bl.arch != KPB_ARCH_LIGHTGLUE) return kpb_fail(ctx, KPB_E_WEIGHTS, "kpb_lg_create: malformed .kpbw blob");
    kpb_lg* lg = new kpb_lg();
    lg->ctx = ctx; lg->desc_scale = desc_scale;
    WeightStage ws;
    auto fail = [&](const std::string& n) { delete lg; return kpb_fail(ctx, KPB_E_WEIGHTS, "kpb_lg_create: tensor %s missing or mis-shaped", n.c_str()); };
    auto linear = [&](const std::string& key, const std::string& name, uint32_t cout, uint32_t cin) -> bool {
        const float* w = bl.get((key + ".weight").c_str(), {cout, cin});
        const float* b = bl.get((key + ".bias").c_str(), {cout});
        if (!w || !b) return false;
        if (conv_mfma_use_h16()) {
            const float sc = weight_scale_h(w, (size_t)cout * cin);
            ws.put(name + ".w", pack_mfma_h(w, (int)cout, (int)cin, 1, 32, 2, sc));
            ws.wscale[name + ".w"] = sc;
        } else {
            ws.put(name + ".w", pack_mfma(w, (int)cout, (int)cin, 1, 32, 2));
        }
        ws.put(name + ".b", pad_bias(b, (int)cout, 64));
        return true;
    };
    // split-f16 form only: the rows of `key` in the order `perm` (GE_ROTARY), or `key` and `key2` stacked (GE_SPLIT2)
    auto linear_rows = [&](const std::string& key, const std::string& key2, const std::string& name, uint32_t cout, uint32_t cin, bool permute) -> bool {
        const uint32_t c1 = key2.empty() ? cout : cout / 2;
        const float* w = bl.get((key + ".weight").c_str(), {c1, cin});
        const float* b = bl.get((key + ".bias").c_str(), {c1});
        const float* w2 = key2.empty() ? nullptr : bl.get((key2 + ".weight").c_str(), {c1, cin});
        const float* b2 = key2.empty() ? nullptr : bl.get((key2 + ".bias").c_str(), {c1});
        if (!w || !b || (!key2.empty() && (!w2 || !b2))) return false;
        std::vector<float> wr((size_t)cout * cin), br(cout);
        for (uint32_t o = 0; o < cout; ++o) {
            const float* srcw = o < c1 ? w : w2;
            const float* srcb = o < c1 ? b : b2;
            const uint32_t r = o < c1 ? (permute ? (uint32_t)lg_qkv_row((int)o) : o) : o - c1;
            std::memcpy(&wr[(size_t)o * cin], srcw + (size_t)r * cin, cin * sizeof(float));
            br[o] = srcb[r];
        }
        const float sc = weight_scale_h(wr.data(), wr.size());
        ws.put(name + ".w", pack_mfma_h(wr.data(), (int)cout, (int)cin, 1, 32, 2, sc));
        ws.wscale[name + ".w"] = sc;
        ws.put(name + ".b", pad_bias(br.data(), (int)cout, 64));
        return true;
    };
    auto vec = [&](const std::string& key, const std::string& name, std::vector<uint32_t> dims) -> bool {
        const float* v = bl.get(key.c_str(), dims);
        if (!v) return false;
        size_t n = 1; for (uint32_t d : dims) n *= d;
        ws.put_raw(name, v, n);
        return true;
    };
    {
        auto it = bl.t.find("input_proj.weight");
        if (it != bl.t.end()) {
            lg->input_dim = (int)it->second.second[1];
            if (lg->input_dim % 32 || !linear("input_proj", "input_proj", 256, (uint32_t)lg->input_dim)) return fail("input_proj");
        }
    }
    if (!vec("posenc.Wr.weight", "posenc.Wr", {32, 2})) return fail("posenc.Wr.weight");
    for (int i = 0; i < NL; ++i) {
        const std::string sa = "transformers." + std::to_string(i) + ".self_attn", ca = "transformers." + std::to_string(i) + ".cross_attn";
        const std::string L = "L" + std::to_string(i);
        // the self-attention q/k/v product and the cross-attention qk / v products: fused forms for split-f16 (see lg_linear)
        if (conv_mfma_use_h16() ? (!linear_rows(sa + ".Wqkv", "", L + ".Wqkv_r", 768, 256, true) ||
                                   !linear_rows(ca + ".to_qk", ca + ".to_v", L + ".toqkv", 512, 256, false))
                                : (!linear(sa + ".Wqkv", L + ".Wqkv", 768, 256) || !linear(ca + ".to_qk", L + ".toqk", 256, 256) ||
                                   !linear(ca + ".to_v", L + ".tov", 256, 256)))
            return fail(L);
        if (!linear(sa + ".out_proj", L + ".sout", 256, 256) ||
            !linear(sa + ".ffn.0", L + ".sffn0", 512, 512) || !linear(sa + ".ffn.3", L + ".sffn3", 256, 512) ||
            !vec(sa + ".ffn.1.weight", L + ".sln.g", {512}) || !vec(sa + ".ffn.1.bias", L + ".sln.b", {512}) ||
            !linear(ca + ".to_out", L + ".toout", 256, 256) ||
            !linear(ca + ".ffn.0", L + ".cffn0", 512, 512) || !linear(ca + ".ffn.3", L + ".cffn3", 256, 512) ||
            !vec(ca + ".ffn.1.weight", L + ".cln.g", {512}) || !vec(ca + ".ffn.1.bias", L + ".cln.b", {512}))
            return fail(L);
        const std::string la = "log_assignment." + std::to_string(i);
        if (!linear(la + ".final_proj", L + ".fproj", 256, 256) || !vec(la + ".matchability.weight", L + ".mw", {1, 256}) || !vec(la + ".matchability.bias", L + ".mb", {1}))
            return fail(la);
        if (i < NL - 1) {
            const std::string tc = "token_confidence." + std::to_string(i) + ".token.0";
            if (!vec(tc + ".weight", L + ".cw", {1, 256}) || !vec(tc + ".bias", L + ".cb", {1})) return fail(tc);
        }
    }
    LgNetShim shim;
    shim.ctx = ctx;
    if (int rc = ws.upload(&shim)) { delete lg; return rc; }
    lg->wdev = shim.wdev; lg->off = shim.off; lg->wscale = shim.wscale;
    shim.wdev = nullptr;
    *out = lg;
    return KPB_OK;
}

KPB_API void kpb_lg_destroy(kpb_lg* lg)
{
    if (!lg) return;
    (void)hipSetDevice(lg->ctx->device);
    (void)hipStreamSynchronize(lg->ctx->stream);
    if (lg->wdev) (void)hipFree(lg->wdev);
    if (lg->ws.p) (void)hipFree(lg->ws.p);
    delete lg;
}

KPB_API int kpb_lg_input_dim(const kpb_lg* lg) { return lg ? lg->input_dim : 0; }

KPB_API int kpb_lg_set_attention(kpb_lg* lg, int mode)
{
    if (!lg) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_lg_set_attention: null matcher");
    if (mode != 0 && mode != 1) return kpb_fail(lg->ctx, KPB_E_INVALID, "kpb_lg_set_attention: mode %d (0 = fp32-equivalent, 1 = f16 as the reference on a GPU)", mode);
    if (mode == 1 && !conv_mfma_use_h16())
        return kpb_fail(lg->ctx, KPB_E_UNSUPPORTED, "kpb_lg_set_attention: the f16 form needs the matrix-pipe build of the attention (KPB_FP32_MATRIX is set)");
    lg->attn_f16 = mode;
    return KPB_OK;
}

KPB_API int kpb_lg_match(kpb_lg* lg, const float* pts0_dev, const float* pts1_dev, const int32_t* n0_dev, const int32_t* n1_dev,
                         int batch, int max_k, const float* desc0_dev, const float* desc1_dev, int C, int Hd, int Wd,
                         int64_t sb, int64_t sc, int64_t sh, int64_t sw, int img_w, int img_h, const kpb_lg_params* prm,
                         int32_t* out_pairs_dev, float* out_scores_dev, int32_t* out_k_dev, int32_t* out_stop_dev)
{
    if (!lg) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_lg_match: null matcher");
    kpb_ctx* ctx = lg->ctx;
    if (!pts0_dev || !pts1_dev || !desc0_dev || !desc1_dev || !prm || !out_pairs_dev || !out_scores_dev || !out_k_dev || batch <= 0 || max_k <= 0)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_lg_match: bad argument");
    if (C != lg->input_dim) return kpb_fail(ctx, KPB_E_INVALID, "kpb_lg_match: descriptor maps have %d channels, the weights expect %d", C, lg->input_dim);
    if (C > 256) return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_lg_match: input_dim %d > 256", C);
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    const int B = batch, S = 2 * B, MP = ((max_k + 127) / 128) * 128;
    const size_t T = (size_t)S * MP;
    // workspace carve (floats unless noted)
    size_t need = 0;
    auto take = [&](size_t n) { const size_t o = need; need += (n + 63) / 64 * 64; return o; };
    const size_t o_kpx = take(T * 2), o_cos0 = take(T * NF), o_sin0 = take(T * NF), o_cos1 = take(T * NF), o_sin1 = take(T * NF),
                 o_din = take(T * C), o_cat0 = take(T * 512), o_cat1 = take(T * 512), o_qkv = take(T * 768), o_q = take(T * D), o_k = take(T * D),
                 o_v = take(T * D), o_ctx = take(T * D), o_h1 = take(T * 512), o_y = take(T * D), o_conf = take(T), o_msc = take(T), o_z = take(T),
                 o_md = take(T * D), o_sim = take((size_t)B * MP * MP), o_mx = take(T), o_lg = take(T), o_bv = take(T),
                 o_ind0 = take(T), o_ind1 = take(T), o_dst = take(T), o_bi = take(T), o_ints = take((size_t)8 * S + 64), o_ls = take(T),
                 o_pmx = take((size_t)B * (MP / 32) * MP), o_psum = take((size_t)B * (MP / 32) * MP),      // column partials per chunk of LG_CROWS rows
                 o_pbv = take((size_t)B * (MP / 32) * MP), o_pbi = take((size_t)B * (MP / 32) * MP),
                 o_kf = take(T * D), o_vf = take(T * D),      // K / V in MFMA operand order, (hi, lo) halves: 4 bytes per element
                 o_kve = take((size_t)S * NH * (MP / 32) * 2);  // per 32-key block: exponents of the K and V scales
    const bool fresh = need * sizeof(float) > lg->ws.cap;
    if (int rc = kpb_reserve(ctx, lg->ws, need * sizeof(float))) return rc;
    float* base = static_cast<float*>(lg->ws.p);
    hipStream_t st = ctx->stream;
    if (fresh) KPB_HIP(ctx, hipMemsetAsync(base, 0, need * sizeof(float), st));   // padded rows must stay finite
    float *kpx = base + o_kpx, *cosb[2] = {base + o_cos0, base + o_cos1}, *sinb[2] = {base + o_sin0, base + o_sin1}, *din = base + o_din,
          *cat[2] = {base + o_cat0, base + o_cat1}, *qkv = base + o_qkv, *q = base + o_q, *k = base + o_k, *v = base + o_v, *cx = base + o_ctx,
          *h1 = base + o_h1, *y = base + o_y, *conf = base + o_conf, *msc = base + o_msc, *zlog = base + o_z, *md = base + o_md, *sim = base + o_sim,
          *mxo = base + o_mx, *lgo = base + o_lg, *bestv = base + o_bv, *lso = base + o_ls, *pmx = base + o_pmx, *psum = base + o_psum, *pbv = base + o_pbv;
    int* pbi = reinterpret_cast<int*>(base + o_pbi);
    const int NCHK = MP / LG_CROWS;
    static_assert(LG_CROWS == 32, "the partial arrays are carved for 32-row chunks");
    int *ind[2] = {reinterpret_cast<int*>(base + o_ind0), reinterpret_cast<int*>(base + o_ind1)}, *dst = reinterpret_cast<int*>(base + o_dst),
        *besti = reinterpret_cast<int*>(base + o_bi), *ints = reinterpret_cast<int*>(base + o_ints);
    uint4 *kfrag = reinterpret_cast<uint4*>(base + o_kf), *vfrag = reinterpret_cast<uint4*>(base + o_vf);
    int* kvexp = reinterpret_cast<int*>(base + o_kve);
    int *cnt = ints, *cnt_orig = ints + S, *newcnt = ints + 2 * S, *active_seq = ints + 3 * S, *fin_seq = ints + 4 * S, *active_pair = ints + 5 * S,
        *fin_pair = ints + 5 * S + B, *stop = ints + 6 * S, *done_pair = ints + 6 * S + B;

    KPB_LAUNCH(ctx, "lg_init_out", lg_init_out, dim3(cdiv(B, 256)), dim3(256), 0, st, out_k_dev, out_stop_dev, stop, B, 0);
    PrepArgs pa{pts0_dev, pts1_dev, n0_dev, n1_dev, lg->wp("posenc.Wr"), kpx, cosb[0], sinb[0], ind[0], cnt, cnt_orig, active_seq, active_pair, stop, done_pair,
                max_k, MP, (float)(img_w - 1), (float)(img_h - 1)};
    KPB_LAUNCH(ctx, "lg_prepare", lg_prepare, dim3(S), dim3(256), 0, st, pa);
    SampleArgs sa{desc0_dev, desc1_dev, sb, sc, sh, sw, kpx, cnt, din, C, Hd, Wd, MP, lg->desc_scale};
    KPB_LAUNCH(ctx, "lg_sample", lg_sample, dim3(cdiv(max_k, 4), S), dim3(256), 0, st, sa);
    int rc;
    if (lg->off.count("input_proj.w")) {
        if ((rc = lg_linear(ctx, lg, "lg_input_proj", "input_proj", C, 256, din, C, cat[0], 512, 0, S, MP, active_seq, cnt))) return rc;
    } else {
        KPB_HIP(ctx, hipMemcpy2DAsync(cat[0], 512 * sizeof(float), din, 256 * sizeof(float), 256 * sizeof(float), T, hipMemcpyDeviceToDevice, st));
    }
    const dim3 tokgrid4(cdiv(max_k, 4), S), tokgrid64(cdiv(max_k * 64, 256), S);
    const bool h16 = conv_mfma_use_h16();
    auto ffn = [&](const std::string& L, const char* pfx, float* c) -> int {     // x + ffn(cat([x, msg])), lightglue.py:185 / 241-242
        int r;
        if ((r = lg_linear(ctx, lg, "lg_ffn0", L + pfx + "ffn0", 512, 512, c, 512, h1, 512, 0, S, MP, active_seq, cnt))) return r;
        KPB_LAUNCH(ctx, "lg_ln_gelu", lg_ln_gelu, tokgrid4, dim3(256), 0, st, h1, lg->wp(L + (pfx[1] == 's' ? ".sln.g" : ".cln.g")),
                   lg->wp(L + (pfx[1] == 's' ? ".sln.b" : ".cln.b")), cnt, active_seq, MP);
        if (h16) {          // the residual sum rides in the product's epilogue: c[:, :256] = c[:, :256] + ffn3(h1)
            LgEpi x; x.epi = GE_RESIDUAL; x.res = c; x.rstride = 512;
            return lg_linear(ctx, lg, "lg_ffn3", L + pfx + "ffn3", 512, 256, h1, 512, c, 512, 0, S, MP, active_seq, cnt, x);
        }
        if ((r = lg_linear(ctx, lg, "lg_ffn3", L + pfx + "ffn3", 512, 256, h1, 512, y, 256, 0, S, MP, active_seq, cnt))) return r;
        KPB_LAUNCH(ctx, "lg_residual", lg_residual, tokgrid64, dim3(256), 0, st, c, y, cnt, active_seq, MP);
        return KPB_OK;
    };
    for (int i = 0; i < NL; ++i) {
        const std::string L = "L" + std::to_string(i);
        float* c = cat[i & 1];
        float *cs = cosb[i & 1], *sn = sinb[i & 1];
        // self attention (lightglue.py:173-185)
        if (h16) {          // q, k rotated and all three laid out head-major by the product's own epilogue
            LgEpi x; x.epi = GE_ROTARY; x.cosb = cs; x.sinb = sn; x.out1 = k; x.out2 = v;
            if ((rc = lg_linear(ctx, lg, "lg_Wqkv", L + ".Wqkv_r", 256, 768, c, 512, q, 256, 0, S, MP, active_seq, cnt, x))) return rc;
        } else {
            if ((rc = lg_linear(ctx, lg, "lg_Wqkv", L + ".Wqkv", 256, 768, c, 512, qkv, 768, 0, S, MP, active_seq, cnt))) return rc;
            KPB_LAUNCH(ctx, "lg_rotary", lg_rotary, dim3(cdiv(max_k * 128, 256), S), dim3(256), 0, st, qkv, cs, sn, q, k, v, cnt, active_seq, MP);
        }
        if (h16) {
            FragArgs fr{k, v, kfrag, vfrag, kvexp, cnt, active_seq, MP, 0, lg->attn_f16};
            KPB_LAUNCH(ctx, "lg_kv_frags", lg_kv_frags, dim3(cdiv(max_k, 32), NH, S), dim3(256), 0, st, fr);
            FlashHArgs fa{q, kfrag, vfrag, kvexp, cx, cnt, active_seq, MP, 0, 0.125f};
            if (lg->attn_f16) KPB_LAUNCH(ctx, "lg_flash_self", lg_flash_h<true>, dim3(cdiv(max_k, 128), NH, S), dim3(256), 0, st, fa);
            else KPB_LAUNCH(ctx, "lg_flash_self", lg_flash_h<false>, dim3(cdiv(max_k, 128), NH, S), dim3(256), 0, st, fa);
        } else {
            FlashArgs fa{q, k, v, cx, cnt, active_seq, MP, 0, 0.125f};
            KPB_LAUNCH(ctx, "lg_flash_self", lg_flash, dim3(cdiv(max_k, 128), NH, S), dim3(256), 0, st, fa);
        }
        if ((rc = lg_linear(ctx, lg, "lg_out_proj", L + ".sout", 256, 256, cx, 256, c, 512, 256, S, MP, active_seq, cnt))) return rc;
        if ((rc = ffn(L, ".s", c))) return rc;
        // cross attention (lightglue.py:216-243)
        if (h16) {
            LgEpi x; x.epi = GE_SPLIT2; x.out1 = v;
            if ((rc = lg_linear(ctx, lg, "lg_to_qkv", L + ".toqkv", 256, 512, c, 512, q, 256, 0, S, MP, active_seq, cnt, x))) return rc;
        } else {
            if ((rc = lg_linear(ctx, lg, "lg_to_qk", L + ".toqk", 256, 256, c, 512, q, 256, 0, S, MP, active_seq, cnt))) return rc;
            if ((rc = lg_linear(ctx, lg, "lg_to_v", L + ".tov", 256, 256, c, 512, v, 256, 0, S, MP, active_seq, cnt))) return rc;
        }
        if (h16) {
            FragArgs fr{q, v, kfrag, vfrag, kvexp, cnt, active_seq, MP, 1, lg->attn_f16};
            KPB_LAUNCH(ctx, "lg_kv_frags", lg_kv_frags, dim3(cdiv(max_k, 32), NH, S), dim3(256), 0, st, fr);
            FlashHArgs fc{q, kfrag, vfrag, kvexp, cx, cnt, active_seq, MP, 1, 0.125f};
            if (lg->attn_f16) KPB_LAUNCH(ctx, "lg_flash_cross", lg_flash_h<true>, dim3(cdiv(max_k, 128), NH, S), dim3(256), 0, st, fc);
            else KPB_LAUNCH(ctx, "lg_flash_cross", lg_flash_h<false>, dim3(cdiv(max_k, 128), NH, S), dim3(256), 0, st, fc);
        } else {
            FlashArgs fc{q, q, v, cx, cnt, active_seq, MP, 1, 0.125f};
            KPB_LAUNCH(ctx, "lg_flash_cross", lg_flash, dim3(cdiv(max_k, 128), NH, S), dim3(256), 0, st, fc);
        }
        if ((rc = lg_linear(ctx, lg, "lg_to_out", L + ".toout", 256, 256, cx, 256, c, 512, 256, S, MP, active_seq, cnt))) return rc;
        if ((rc = ffn(L, ".c", c))) return rc;
        // confidences, stop / prune decision (lightglue.py:557-579)
        const bool last = i == NL - 1;
        KPB_LAUNCH(ctx, "lg_conf", lg_conf, tokgrid4, dim3(256), 0, st, c, last ? lg->wp(L + ".mw") : lg->wp(L + ".cw"), last ? lg->wp(L + ".mb") : lg->wp(L + ".cb"),
                   lg->wp(L + ".mw"), lg->wp(L + ".mb"), conf, msc, zlog, cnt, active_seq, MP, last ? 0 : 1);
        const double thr = std::min(std::max(0.8 + 0.1 * std::exp(-4.0 * i / NL), 0.0), 1.0);
        DecideArgs da{conf, msc, cnt, cnt_orig, newcnt, dst, active_pair, active_seq, fin_pair, fin_seq, stop, done_pair, MP, i, prm->prune_min_kpts,
                      (float)thr, prm->depth_confidence, (float)(1.0 - (double)prm->width_confidence), prm->depth_confidence > 0 ? 1 : 0,
                      prm->width_confidence > 0 ? 1 : 0};
        KPB_LAUNCH(ctx, "lg_decide", lg_decide, dim3(B), dim3(256), 0, st, da);
        // the projected descriptors of the pairs that finish at this layer (lightglue.py:606-614); their assignment runs after the loop
        if ((rc = lg_linear(ctx, lg, "lg_final_proj", L + ".fproj", 256, 256, c, 512, md, 256, 0, S, MP, fin_seq, cnt))) return rc;
        if (!last) {
            KPB_LAUNCH(ctx, "lg_gather", lg_gather, tokgrid64, dim3(256), 0, st, c, cat[(i + 1) & 1], cs, sn, cosb[(i + 1) & 1], sinb[(i + 1) & 1],
                       ind[i & 1], ind[(i + 1) & 1], dst, cnt, active_seq, MP);
            KPB_LAUNCH(ctx, "lg_commit_counts", lg_commit_counts, dim3(cdiv(S, 256)), dim3(256), 0, st, cnt, newcnt, active_seq, S);
        }
    }
    // the assignment of every finished pair (lightglue.py:606-614), ONCE: a pair's projected descriptors, matchability logits, counts and index map stay as its last
    // layer left them (every later kernel skips inactive sequences).  (r03 .. r05 launched the stage after every layer for the pairs finishing there: 40 empty
    // launches per call when all pairs run to the last layer.)
    KPB_LAUNCH(ctx, "lg_sim", lg_sim, dim3(cdiv(max_k, 128), cdiv(max_k, 32), B), dim3(256), 0, st, md, sim, cnt, done_pair, MP);
    const int nrow = cdiv(max_k, 4), ntile = cdiv(max_k, 256) * NCHK;
    KPB_LAUNCH(ctx, "lg_lse", lg_lse, dim3(nrow + ntile, B), dim3(256), 0, st, sim, mxo, lgo, zlog, lso, pmx, psum, cnt, done_pair, MP, NCHK, nrow);
    KPB_LAUNCH(ctx, "lg_lse", lg_lse_colfin, dim3(cdiv(max_k, 256), B), dim3(256), 0, st, pmx, psum, mxo, lgo, cnt, done_pair, MP, NCHK);
    KPB_LAUNCH(ctx, "lg_best", lg_best, dim3(nrow + ntile, B), dim3(256), 0, st, sim, mxo, lgo, lso, bestv, besti, pbv, pbi, cnt, done_pair, MP, NCHK, nrow);
    KPB_LAUNCH(ctx, "lg_emit", lg_emit, dim3(B), dim3(256), 0, st, bestv, besti, pbv, pbi, NCHK, ind[0], ind[1], stop, cnt, done_pair, out_pairs_dev, out_scores_dev,
               out_k_dev, MP, max_k, prm->filter_threshold);
    if (out_stop_dev) KPB_LAUNCH(ctx, "lg_init_out", lg_init_out, dim3(cdiv(B, 256)), dim3(256), 0, st, out_k_dev, out_stop_dev, stop, B, 1);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}
