// detect.hip -- A1..A5 of the hot path: radius-NMS to its fixed point, border mask, raster compaction,
// top-K.  Replaces utils/extracter.py:6-100 (fast_nms), 164-190, 129-161 and 193-221 of the reference.
//
// NMS formulation (DESIGN.md "NMS fixed point"): for a non-negative map the reference's
// unfold/argmax/fold loop converges to the unique fixed point in which a pixel survives iff it is
// the (value desc, raster index asc) maximum of its (2r+1)^2 window among survivors.  The two
// inference rules -- "p is a maximum once every pixel ahead of it in its window is dead" and "q is
// dead once a maximum lies within r of it" -- are monotone, so they may be applied in any order and
// on stale neighbour state; every schedule ends in the same map.  nms_sweep therefore iterates whole
// rounds inside LDS on a 24x64 tile with a 2r halo (no 207 MB unfold buffers, the 1.2 MB map stays in
// L2), updates the map in place, and is re-launched until no tile changed.
#include <algorithm>

#include "kpb_common.h"

namespace {

// 24 x 64 tiles and a 512-entry maxima list keep nms_sweep_r<6> at 38.9 KB of LDS = four workgroups per CU (32 x 64 with 1024: 46.4 KB,
// three per CU, 8 % slower: the sweep is latency-bound)
constexpr int TH = 24, TW = 64, NMS_THREADS = 256, MAXLIST = 512;

struct NmsArgs {
    const float* src;   // [B][P] input maps (read by sweep 0)
    float* cur;         // [B][P] working maps (written by sweep 0, updated in place afterwards)
    int* tchg_prev;     // [B][ntiles] tile-changed flags of the previous sweep (unused in sweep 0)
    int* tchg_cur;      // [B][ntiles] flags written by this sweep
    int* lastchg;       // [B] 1 + index of the last sweep that changed anything
    int* negflag;       // [B] set when a negative score is seen
    // What sweep 0 of nms_sweep_r hands to nms_tail (r06; r03 .. r05 built two lists per image here -- the appends, two returning global atomics per
    // tile and the list flush were 17 % of the kernel, profiles/r06_nms_knockouts.txt):
    //   ubits  [B][H][wb] bytes (wb = W / 8 rounded up to a multiple of 4): bit (y, x) = pixel left undecided (alive, not confirmed); one byte = the
    //          eight pixels an owner thread holds anyway.  nms_tail builds its watch list from them.  Null: not collected.
    //   chist  [B][NMS_HBINS] counters, top-K pruning (kpb_detect with top_k < H*W; null: none): a histogram of the scores of the maxima sweep 0
    //          CONFIRMED inside the border frame and above the output thresholds, one fire-and-forget atomic each.  If an image has more than top_k of
    //          them, nothing that scores below the (top_k+1)-th can reach the output, and nms_tail drops those undecided pixels unresolved; the lower
    //          edge of the bin that holds the (top_k+1)-th is such a bound (nms_score_bin).
    //   cbits  the same layout: bit (y, x) = pixel CONFIRMED as a maximum (sweep 0 writes its tiles' bytes, nms_tail sets the bits of what it confirms):
    //          select_topk reads these 38 KB per image instead of the 1.2 MB map when the tail has settled the image (r06).  Null: not kept.
    unsigned char* ubits; int wb;
    unsigned char* cbits;
    unsigned* chist;
    int border;
    float cmin;         // a confirmed maximum counts when its score is > cmin (threshold / min_score of the detection)
    int H, W, r, tiles_y, tiles_x, sweep, max_local;
    int xcd_map;        // kpb_xcd_tile (nms_sweep_r)
};

// Bin of a positive score: 64 bins per octave (the six leading mantissa bits) over the 64 octaves [2^-62, 4); what lies outside goes to the end bins.
// nms_bin_floor(b) <= every score of bin b (0 for bin 0: no bound).  Any T at or below the true (top_k+1)-th score prunes correctly; this one is at
// most 1.6 % below it.
constexpr int NMS_HBINS = 4096, NMS_HBASE = (129 << 6) - NMS_HBINS;
__device__ __forceinline__ int nms_score_bin(float v)
{
    const int key = (int)(__float_as_uint(v) >> 17) - NMS_HBASE;
    return min(max(key, 0), NMS_HBINS - 1);
}
__device__ __forceinline__ float nms_bin_floor(int b) { return b > 0 ? __uint_as_float((unsigned)(b + NMS_HBASE) << 17) : 0.0f; }


__global__ __launch_bounds__(NMS_THREADS) void nms_sweep(NmsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int r = a.r, LH = TH + 4 * r, LW = TW + 4 * r;
    float* t = reinterpret_cast<float*>(smem);      // tile values: >0 alive, 0 dead, <0 confirmed maximum
    float* e = t + LH * LW;                         // row maxima over [x-r, x+r]
    int* maxlist = reinterpret_cast<int*>(e + LH * LW);
    int* s_cnt = maxlist + MAXLIST;                 // [0] new maxima, [1] any kill, [2] owned kill, [3] overflow

    const int img = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int ntiles = a.tiles_x * a.tiles_y;
    int* tcur = a.tchg_cur + (size_t)img * ntiles;

    if (a.sweep > 0) {  // skip tiles whose 3x3 neighbourhood was quiet last sweep (uniform branch)
        const int* tprev = a.tchg_prev + (size_t)img * ntiles;
        int need = 0;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = ty + dy, xx = tx + dx;
                if (yy >= 0 && yy < a.tiles_y && xx >= 0 && xx < a.tiles_x) need |= tprev[yy * a.tiles_x + xx];
            }
        if (!need) {
            if (tid == 0) tcur[tile] = 0;
            return;
        }
    }

    const size_t P = (size_t)a.H * a.W;
    const float* in = (a.sweep == 0 ? a.src : a.cur) + (size_t)img * P;
    float* out = a.cur + (size_t)img * P;
    const int gy0 = ty * TH - 2 * r, gx0 = tx * TW - 2 * r;

    int neg = 0;
    for (int i = tid; i < LH * LW; i += NMS_THREADS) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = gy0 + ly, gx = gx0 + lx;
        float v = 0.0f;  // F.unfold's zero padding (extracter.py:54-60)
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = in[(size_t)gy * a.W + gx];
        neg |= (v < 0.0f);
        t[i] = v;
    }
    if (tid < 4) s_cnt[tid] = 0;
    __syncthreads();
    if (neg) a.negflag[img] = 1;

    // the 8 owned pixels this thread writes back
    float orig[TH * TW / NMS_THREADS];
#pragma unroll
    for (int k = 0; k < TH * TW / NMS_THREADS; ++k) {
        const int o = tid + k * NMS_THREADS, oy = o / TW, ox = o - oy * TW;
        orig[k] = t[(oy + 2 * r) * LW + ox + 2 * r];
    }

    const int ks = 2 * r + 1, ks2 = ks * ks, IW = LW - 2 * r, IH = LH - 2 * r;
    int unconverged = 1;
    for (int iter = 0; iter < a.max_local; ++iter) {
        // pass 1: e[y][x] = max t[y][x-r..x+r], all rows, columns [r, LW-r)
        for (int i = tid; i < LH * IW; i += NMS_THREADS) {
            const int y = i / IW, x = i - y * IW + r;
            const float* row = t + y * LW + x;
            float m = row[-r];
            for (int d = -r + 1; d <= r; ++d) m = fmaxf(m, row[d]);
            e[y * LW + x] = m;
        }
        __syncthreads();
        // pass 2: maxima among alive pixels of the inset region (their whole window is in LDS).
        // argmax returns the first index of the maximum (extracter.py:69-70): the centre must be
        // strictly greater than the rows above and the cells to its left, >= the rest.
        for (int i = tid; i < IH * IW; i += NMS_THREADS) {
            const int y = i / IW + r, x = i - (y - r) * IW + r;
            const float v = t[y * LW + x];
            if (v > 0.0f && v == e[y * LW + x]) {
                bool ok = true;
                for (int d = 1; d <= r && ok; ++d) ok = v > t[y * LW + x - d];
                for (int d = 1; d <= r && ok; ++d) ok = (v > e[(y - d) * LW + x]) && (v >= e[(y + d) * LW + x]);
                if (ok) {
                    const int slot = atomicAdd(&s_cnt[0], 1);
                    if (slot < MAXLIST) maxlist[slot] = y * LW + x;
                    else s_cnt[3] = 1;
                }
            }
        }
        __syncthreads();
        const int nmax = min(s_cnt[0], MAXLIST);
        // kill pass: zero the window of every new maximum (extracter.py:81-96), mark the maximum itself
        for (int i = tid; i < nmax * ks2; i += NMS_THREADS) {
            const int m = i / ks2, c = i - m * ks2;
            const int dy = c / ks - r, dx = c - (dy + r) * ks - r;
            const int pos = maxlist[m] + dy * LW + dx;
            if (dy == 0 && dx == 0) {
                t[pos] = -t[pos];
            } else if (t[pos] != 0.0f) {
                t[pos] = 0.0f;
                s_cnt[1] = 1;
                const int py = pos / LW - 2 * r, px = pos - (py + 2 * r) * LW - 2 * r;
                if (py >= 0 && py < TH && px >= 0 && px < TW) s_cnt[2] = 1;
            }
        }
        __syncthreads();
        const int any_kill = s_cnt[1], overflow = s_cnt[3];
        __syncthreads();
        if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; s_cnt[3] = 0; }
        __syncthreads();
        if (!any_kill && !overflow) { unconverged = 0; break; }
    }

    // write back owned pixels that changed (sweep 0 writes everything: it materialises cur)
    const int by = ty * TH, bx = tx * TW;
#pragma unroll
    for (int k = 0; k < TH * TW / NMS_THREADS; ++k) {
        const int o = tid + k * NMS_THREADS, oy = o / TW, ox = o - oy * TW;
        const int gy = by + oy, gx = bx + ox;
        if (gy < a.H && gx < a.W) {
            const float v = fabsf(t[(oy + 2 * r) * LW + ox + 2 * r]);
            if (a.sweep == 0 || v != orig[k]) out[(size_t)gy * a.W + gx] = v;
        }
    }
    if (tid == 0) {
        const int flag = (s_cnt[2] || unconverged) ? 1 : 0;
        tcur[tile] = flag;
        if (flag) atomicMax(&a.lastchg[img], a.sweep + 1);
    }
}

// ------------------------------------------------------------------------------------------------
// Specialised sweep for nms_dist 1..8 (compile-time radius): same rules as nms_sweep, with the two
// window passes done on 8-pixel register strips -- a horizontal strip is five ds_read_b128 for
// r = 6 instead of 104 ds_read_b32, the running maxima are built by tripling on v_max3_f32 (widths 3, 9) --
// so one local round costs ~4 LDS accesses and ~10 VALU ops per pixel.
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }       // one v_max3_f32

// o[i] = max(a[i .. i+WIN-1]); NIN >= NOUT + WIN - 1.  Running maxima are built by TRIPLING on the three-input maximum (widths
// 3, 9) and the window is covered by two or three of them, overlapping where they must: 38 instructions for eight outputs of a
// 13-wide window (r = 6) where doubling on the two-input maximum (r02) took 61 -- the sweep is bound by vector issue.
template <int WIN, int NOUT, int NIN>
__device__ __forceinline__ void window_max(const float (&a)[NIN], float (&o)[NOUT])
{
    static_assert(NIN >= NOUT + WIN - 1, "window_max input too short");
    static_assert(WIN >= 1 && WIN <= 18, "window_max: windows of up to 18");
    if constexpr (WIN == 1) {
#pragma unroll
        for (int i = 0; i < NOUT; ++i) o[i] = a[i];
    } else if constexpr (WIN == 2) {
#pragma unroll
        for (int i = 0; i < NOUT; ++i) o[i] = fmaxf(a[i], a[i + 1]);
    } else {
        constexpr int N3 = NOUT + WIN - 3;          // m3[i] = max a[i .. i+2], i < N3
        float m3[N3];
#pragma unroll
        for (int i = 0; i < N3; ++i) m3[i] = max3f(a[i], a[i + 1], a[i + 2]);
        if constexpr (WIN == 3) {
#pragma unroll
            for (int i = 0; i < NOUT; ++i) o[i] = m3[i];
        } else if constexpr (WIN <= 6) {
#pragma unroll
            for (int i = 0; i < NOUT; ++i) o[i] = fmaxf(m3[i], m3[i + WIN - 3]);
        } else if constexpr (WIN <= 8) {
#pragma unroll
            for (int i = 0; i < NOUT; ++i) o[i] = max3f(m3[i], m3[i + 3], m3[i + WIN - 3]);
        } else {
            constexpr int N9 = NOUT + WIN - 9;      // m9[i] = max a[i .. i+8], i < N9
            float m9[N9];
#pragma unroll
            for (int i = 0; i < N9; ++i) m9[i] = max3f(m3[i], m3[i + 3], m3[i + 6]);
            if constexpr (WIN == 9) {
#pragma unroll
                for (int i = 0; i < NOUT; ++i) o[i] = m9[i];
            } else {
#pragma unroll
                for (int i = 0; i < NOUT; ++i) o[i] = fmaxf(m9[i], m9[i + WIN - 9]);
            }
        }
    }
}

template <int R>
__global__ __launch_bounds__(NMS_THREADS) void nms_sweep_r(NmsArgs a)
{
    constexpr int LH = TH + 4 * R, LW = TW + 4 * R;
    constexpr int IW = LW - 2 * R, IH = LH - 2 * R;            // inset region: whole window inside the tile
    constexpr int NHS = (IW + 7) / 8;                           // 8-pixel strips per row
    // vertical strips: as few per column as one pass of the workgroup can take (threads >= NVS * IW), so that the column pass is
    // ONE trip of the thread loop (r = 6: three strips of 12 rows, 228 of 256 threads; five strips of 8 took two trips, the second
    // half empty)
    constexpr int NVS = (3 * IW <= NMS_THREADS) ? 3 : (4 * IW <= NMS_THREADS ? 4 : (IH + 7) / 8), VS = (IH + NVS - 1) / NVS;
    static_assert(VS <= 16, "vertical strip too tall for the hit mask / register budget");
    // Row pitches (r06): 16-byte aligned rows whose pitch in 16-byte slots is ODD.  The horizontal pass walks DOWN a strip column -- consecutive
    // lanes are consecutive rows -- so the sixteen lanes one ds_read_b128 lane group holds (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27} ...,
    // sixteen different rows mod 16) land on sixteen different 4-bank slots when the row pitch is an odd number of slots, and the eight lanes of a
    // ds_write_b128 group on eight different slots of the 32 banks.  (r05: consecutive lanes were consecutive STRIPS of a row, 8 floats apart
    // against a pitch of 92 / 80 floats: two-way conflicts on most reads and every write -- 40 % of the LDS cycles.)
    constexpr int PITCH0 = ((8 * NHS + 2 * R) + 3) / 4 * 4;
    constexpr int PITCH = (PITCH0 / 4) % 2 ? PITCH0 : PITCH0 + 4;      // floats per LDS row of t
    constexpr int EP = 8 * NHS + 4;                                     // floats per row of e (2 NHS + 1 slots); e[y][x'] belongs to tile column x' + R
    constexpr int ROWS = (VS * NVS + 2 * R > LH) ? VS * NVS + 2 * R : LH;      // rows incl. zero padding read by the last strip
    constexpr int SW8 = (NHS % 2 == 0) ? 2 : 1, NST = NHS / SW8, SWP = 8 * SW8;      // 8-pixel strips per thread of the horizontal pass, threads per row, pixels per thread
    constexpr int HQ2 = (SWP + 2 * R + 3) / 4;                  // float4s one horizontal strip reads
    constexpr int KS = 2 * R + 1;
    // the working map carries the state in the sign: > 0 alive, 0 dead, < 0 confirmed maximum (its value negated);
    // confirmed maxima are never re-derived, a tile only has to clear what they still cover
    __shared__ __attribute__((aligned(16))) float t[ROWS * PITCH];
    __shared__ __attribute__((aligned(16))) float e[ROWS * EP];      // row maxima over [x-R, x+R]
    // row candidates (r05): bit i of hm[strip][row] = pixel 8 strip + i + R of the row is alive, equals its row maximum and is greater
    // than the R cells to its left -- everything the maximum test needs to know about the pixel's own row, decided by the horizontal
    // pass while the row is in its registers.  Bytes, [strip][row + HOFF] so that the rows of one vertical strip are whole words.
    constexpr int HOFF = (4 - R % 4) % 4;
    constexpr bool HALIGNED = VS % 4 == 0;                               // every strip starts on a word: (R + HOFF + VS s) % 4 == 0
    constexpr int HNW = HALIGNED ? VS / 4 : (VS + 6) / 4;                // words one vertical strip reads
    constexpr int HMR = (R + VS * NVS + HOFF + 8 + 3) / 4 * 4;           // bytes per strip column, incl. what the last strip's words overhang
    static_assert(HMR >= LH + HOFF, "row-candidate plane too short");
    __shared__ unsigned hm[NHS * HMR / 4];
    __shared__ int maxlist[MAXLIST];
    __shared__ int s_n[2], s_changed, s_over;

    const kpb_tile3 wg = kpb_xcd_tile(a.xcd_map);      // neighbouring tiles (they re-read each other's 2R halo) on one XCD's L2
    const int img = wg.y, tile = wg.x, tid = threadIdx.x, lane = tid & 63;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int ntiles = a.tiles_x * a.tiles_y;
    int* tcur = a.tchg_cur + (size_t)img * ntiles;
    if (a.sweep > 0) {
        const int* tprev = a.tchg_prev + (size_t)img * ntiles;
        int need = 0;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = ty + dy, xx = tx + dx;
                if (yy >= 0 && yy < a.tiles_y && xx >= 0 && xx < a.tiles_x) need |= tprev[yy * a.tiles_x + xx];
            }
        if (!need) {
            if (tid == 0) tcur[tile] = 0;
            return;
        }
    }
    const size_t P = (size_t)a.H * a.W;
    const float* in = (a.sweep == 0 ? a.src : a.cur) + (size_t)img * P;
    float* out = a.cur + (size_t)img * P;
    const int gy0 = ty * TH - 2 * R, gx0 = tx * TW - 2 * R;
    const bool first = a.sweep == 0;

    int neg = 0;
    // zero padding outside the image (extracter.py:54-60) and in the pad rows/columns
    if ((R % 2 == 0) && (a.W % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0)) {
        // gx0 is a multiple of 4: whole float4s are inside or outside the image.  All loads of a thread go out before the first is used,
        // without a branch: a float4 outside the image is fetched from the image's first bytes and replaced by zeros (r05: the nine
        // exec-masked blocks and 64-bit address chains of the branchy form were a fifth of the sweep's instructions).  The image base is
        // wave-uniform, the offset 32 bits (an image has < 2^30 pixels: nms_open checks).
        constexpr int Q = PITCH / 4, NQ = ROWS * Q, PER = (NQ + NMS_THREADS - 1) / NMS_THREADS;
        float4 buf[PER];
        bool okk[PER];
        const unsigned char* inb = reinterpret_cast<const unsigned char*>(in);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * NMS_THREADS;
            const int ly = i / Q, lx = (i - ly * Q) * 4;
            const int gy = gy0 + ly, gx = gx0 + lx;
            okk[k] = i < NQ && ly < LH && lx < LW && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const unsigned off = okk[k] ? ((unsigned)gy * (unsigned)a.W + (unsigned)gx) * 4u : 0u;
            buf[k] = *reinterpret_cast<const float4*>(inb + off);
        }
        float mn = 0.0f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * NMS_THREADS;
            if (!okk[k]) buf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            mn = fminf(fminf(mn, fminf(buf[k].x, buf[k].y)), fminf(buf[k].z, buf[k].w));
            if (i < NQ) *reinterpret_cast<float4*>(t + 4 * i) = buf[k];
        }
        neg = mn < 0.0f;
    } else {
#pragma unroll 8
        for (int i = tid; i < ROWS * PITCH; i += NMS_THREADS) {
            const int ly = i / PITCH, lx = i - ly * PITCH;
            const int gy = gy0 + ly, gx = gx0 + lx;
            float v = 0.0f;
            if (ly < LH && lx < LW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = in[(size_t)gy * a.W + gx];
            neg |= (v < 0.0f);
            t[i] = v;
        }
    }
    for (int i = LH * EP + tid; i < ROWS * EP; i += NMS_THREADS) e[i] = 0.0f;      // pad rows only: rows < LH are rewritten by every horizontal pass
    if (tid == 0) { s_n[0] = 0; s_n[1] = 0; s_changed = 0; s_over = 0; }
    __syncthreads();
    if (neg && first) a.negflag[img] = 1;    // only the input map may not be negative; later sweeps use the sign themselves

    // write-back ownership: thread o < TH * TW / 8 owns the eight pixels (oy, ox .. ox + 7) of the tile
    static_assert(TH * TW / 8 <= NMS_THREADS && TW % 8 == 0, "write-back mapping");
    const bool owner = tid < TH * TW / 8;
    const int oy = tid / (TW / 8), ox = 8 * (tid - oy * (TW / 8));
    float orig[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) orig[k] = owner ? t[(oy + 2 * R) * PITCH + ox + 2 * R + k] : 0.0f;

    // appends the positions flagged by `hit` to maxlist with one LDS atomic per wave
    auto append = [&](bool hit, int pos, int par) {
        const unsigned long long bal = __ballot(hit);
        if (bal) {
            int base = 0;
            if (lane == __ffsll((long long)bal) - 1) base = atomicAdd(&s_n[par], __popcll(bal));
            base = __shfl(base, __ffsll((long long)bal) - 1, 64);
            if (hit) {
                const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
                if (slot < MAXLIST) maxlist[slot] = pos;
                else s_over = 1;
            }
        }
    };
    // zeroes the window of every listed maximum; no two maxima lie within R of each other, so nothing but dead or
    // doomed pixels is overwritten and no read is needed (extracter.py:81-96)
    // (one thread per window ROW: 2r+1 stores at constant offsets; a thread per window CELL spent twenty instructions of
    // index arithmetic on every store and the kill was a seventh of the sweep)
    // r06: one thread per window COLUMN (2r+1 stores at the constant offsets dy * PITCH): the lanes of a maximum store to consecutive banks.  A
    // thread per window ROW (r03 .. r05) put consecutive lanes PITCH floats apart -- a multiple of four, so eight banks for 32 lanes.
    auto kill = [&](int nmax, bool mark_centre) {
        for (int i = tid; i < nmax * KS; i += NMS_THREADS) {
            const int m = i / KS, dx = i - m * KS - R;
            const int c = maxlist[m];
            float* col = t + c + dx - R * PITCH;
            if (dx != 0) {
#pragma unroll
                for (int dy = 0; dy < KS; ++dy) col[dy * PITCH] = 0.0f;
            } else {
                const float centre = col[R * PITCH];
#pragma unroll
                for (int dy = 0; dy < KS; ++dy) if (dy != R) col[dy * PITCH] = 0.0f;
                if (mark_centre) col[R * PITCH] = -centre;
            }
        }
    };
    // appends the set bits of a lane's hit mask (positions base_pos + k * stride) to maxlist: one LDS atomic per lane that has
    // any -- maxima and undecided pixels are sparse, a ballot per bit was most of the column pass
    auto append_mask = [&](unsigned mask, int base_pos, int stride, int par) {
        if (mask) {
            int slot = atomicAdd(&s_n[par], __popc(mask));
            while (mask) {
                const int k = __ffs((int)mask) - 1;
                mask &= mask - 1;
                if (slot < MAXLIST) maxlist[slot] = base_pos + k * stride;
                else s_over = 1;
                ++slot;
            }
        }
    };

    if (!first) {   // maxima confirmed in earlier sweeps (possibly by a neighbour tile): clear what they cover here
        for (int i = tid; i < IH * IW + (NMS_THREADS - 1); i += NMS_THREADS) {
            const bool in_range = i < IH * IW;
            const int y = in_range ? i / IW + R : 0, x = in_range ? i - (y - R) * IW + R : 0;
            append(in_range && t[y * PITCH + x] < 0.0f, y * PITCH + x, 0);
        }
        __syncthreads();
        kill(min(s_n[0], MAXLIST), false);
        __syncthreads();
        if (tid == 0) { s_n[0] = 0; if (s_over) { s_over = 0; s_changed = 1; } }
        __syncthreads();
    }

    int unconverged = 1;
    for (int iter = 0; iter < a.max_local; ++iter) {
        const int par = iter & 1;
        // horizontal pass: e[y][x - R] = max t[y][x-R .. x+R] for x in [R, R + 8*NHS).  r06: a thread takes SIXTEEN pixels of a row where the row has an
        // even number of 8-pixel strips (r = 5 .. 8: 28 inputs, 78 maxima and seven 16-byte reads per 16 outputs; two threads on 8 pixels each took
        // 108 and ten) -- and the LH x NHS / 2 = 240 strip-threads of a 24 x 64 tile at r = 6 are ONE trip of the workgroup instead of 1.9.
        for (int i = tid; i < LH * NST; i += NMS_THREADS) {
            const int st = i / LH, y = i - st * LH;          // down a strip column (see PITCH)
            const float* row = t + y * PITCH + SWP * st;
            float raw[HQ2 * 4], v[HQ2 * 4];
#pragma unroll
            for (int q = 0; q < HQ2; ++q) {     // |t|: a confirmed maximum (stored negated) still outranks everything near it
                const float4 f = *reinterpret_cast<const float4*>(row + 4 * q);
                raw[4 * q] = f.x; raw[4 * q + 1] = f.y; raw[4 * q + 2] = f.z; raw[4 * q + 3] = f.w;
                v[4 * q] = fabsf(f.x); v[4 * q + 1] = fabsf(f.y); v[4 * q + 2] = fabsf(f.z); v[4 * q + 3] = fabsf(f.w);
            }
            float o[SWP], wl[SWP];
            window_max<KS, SWP, HQ2 * 4>(v, o);
            window_max<R, SWP, HQ2 * 4>(v, wl);   // wl[i] = max |t| over the R cells left of pixel i + R (shares its triples with o)
            float* er = e + y * EP + SWP * st;
#pragma unroll
            for (int q = 0; q < SWP / 4; ++q) *reinterpret_cast<float4*>(er + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
            // c > wl >= 0 makes c alive (positive, not a confirmed maximum); c == o: nothing in the row window exceeds it; the strict
            // test on the left is argmax's first-index rule (extracter.py:69-70)
#pragma unroll
            for (int j = 0; j < SW8; ++j) {
                unsigned m = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) m |= (raw[8 * j + k + R] > wl[8 * j + k] && raw[8 * j + k + R] == o[8 * j + k]) ? 1u << k : 0u;
                reinterpret_cast<unsigned char*>(hm)[(SW8 * st + j) * HMR + y + HOFF] = (unsigned char)m;
            }
        }
        __syncthreads();
        // vertical pass on VS-row strips: a pixel is a maximum iff it is a row candidate (alive, equal to its row maximum, greater than
        // the R cells to its left: hm), > every row maximum above and >= every one below (argmax = first index, extracter.py:69-70).
        // r05: the row half of the test used to be made here, per pixel behind five-way branches (a candidate's R left neighbours
        // read one by one): 700 instructions per strip, the largest phase of the sweep; now ~150, branch-free.
        for (int i0 = 0; i0 < NVS * IW; i0 += NMS_THREADS) {
            const int i = i0 + tid;
            const bool in_range = i < NVS * IW;
            const int s8 = in_range ? i / IW : 0, x = in_range ? i - s8 * IW + R : R;
            const int y0 = R + VS * s8;
            unsigned rh = 0;
            {
                const int xe = x - R, hb = (xe >> 3) * HMR + y0 + HOFF;        // byte of row y0 in the strip column of pixel x
                const unsigned* hw = hm + (hb >> 2);
#pragma unroll
                for (int w = 0; w < HNW; ++w) rh |= ((((hw[w] >> (xe & 7)) & 0x01010101u) * 0x01020408u) >> 24) << (4 * w);
                if (!HALIGNED) rh >>= hb & 3;
                const int kmax = LH - R - y0;                                   // rows y0 + k < LH - R only
                rh &= kmax >= VS ? (1u << VS) - 1u : (kmax > 0 ? (1u << kmax) - 1u : 0u);
                if (!in_range) rh = 0;
            }
            unsigned hits = 0;
            if (rh) {
                float col[VS + 2 * R];
#pragma unroll
                for (int k = 0; k < VS + 2 * R; ++k) col[k] = e[(y0 - R + k) * EP + x - R];
                float wr[VS + R + 1];
                window_max<R, VS + R + 1, VS + 2 * R>(col, wr);    // wr[j] = max col[j .. j+R-1]
#pragma unroll
                for (int k = 0; k < VS; ++k) {
                    const float v = col[k + R];                      // a row candidate IS its row maximum
                    hits |= (((rh >> k) & 1u) && v > wr[k] && v >= wr[k + R + 1]) ? 1u << k : 0u;
                }
            }
            append_mask(hits, y0 * PITCH + x, PITCH, par);
        }
        __syncthreads();
        const int nmax = min(s_n[par], MAXLIST);
        if (tid == 0) s_n[par ^ 1] = 0;
        if (nmax == 0 && !s_over) { unconverged = 0; break; }    // nothing new: the tile is at its local fixed point
        kill(nmax, true);
        __syncthreads();
        if (s_over && tid == 0) s_over = 0;   // the next round re-finds what did not fit
    }

    const int by = ty * TH, bx = tx * TW;
    int changed = 0;
    // write-back: eight pixels per owner thread, 16-byte stores where the row allows them; in sweep 0 the owner's byte of the two hand-off bitmaps (NmsArgs::ubits)
    const int gy = by + oy, gx = bx + ox;
    const bool mine = owner && gy < a.H && gx < a.W;
    const bool whole = gx + 8 <= a.W;
    if (mine) {
        unsigned umask = 0, cmask = 0, diff = 0;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t[(oy + 2 * R) * PITCH + ox + 2 * R + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) if ((whole || gx + k < a.W) && v[k] != orig[k]) diff |= 1u << k;
        if (diff) changed = 1;
        float* o = out + (size_t)gy * a.W + gx;
        if (first && whole && (a.W & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) if ((whole || gx + k < a.W) && (first || ((diff >> k) & 1u))) o[k] = v[k];
        }
        if (first && a.ubits) {
            const bool rows_in = gy >= a.border && gy < a.H - a.border;
            unsigned* hist = a.chist ? a.chist + (size_t)img * NMS_HBINS : nullptr;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool in = whole || gx + k < a.W;
                if (in && v[k] > 0.0f) umask |= 1u << k;        // what this tile could not settle goes to nms_tail
                // the maxima confirmed here that the detection could output: their scores bound what can reach the top_k
                if (hist && in && v[k] < 0.0f && -v[k] > a.cmin && rows_in && gx + k >= a.border && gx + k < a.W - a.border) cmask |= 1u << k;
            }
            while (cmask) {         // few bits a thread (maxima lie more than R apart): a loop over them, the score re-read from the tile
                const int k = __ffs((int)cmask) - 1;
                cmask &= cmask - 1;
                atomicAdd(&hist[nms_score_bin(-t[(oy + 2 * R) * PITCH + ox + 2 * R + k])], 1u);
            }
            a.ubits[((size_t)img * a.H + gy) * a.wb + (gx >> 3)] = (unsigned char)umask;
            if (a.cbits) {
                unsigned conf = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) if ((whole || gx + k < a.W) && v[k] < 0.0f) conf |= 1u << k;
                a.cbits[((size_t)img * a.H + gy) * a.wb + (gx >> 3)] = (unsigned char)conf;
            }
        }
    }
    if (changed) s_changed = 1;
    __syncthreads();
    if (tid == 128) {
        const int flag = (s_changed || unconverged) ? 1 : 0;
        tcur[tile] = flag;
        if (flag) atomicMax(&a.lastchg[img], a.sweep + 1);
    }
}

// ------------------------------------------------------------------------------------------------
// A2 + A3 + A4: one 1024-thread workgroup per image.
// ------------------------------------------------------------------------------------------------ sparse tail
// After sweep 0 a few percent of the pixels are still undecided: alive, but with a rival in a neighbouring tile whose
// fate the tile could not know.  They are settled here straight on the signed map in L2, one workgroup per image, by
// the same two monotone rules: an undecided pixel dies when a confirmed maximum lies within r, and is confirmed once
// nothing alive in its window is ahead of it (greater, or equal and earlier in raster order).  A pixel found blocked
// remembers one blocker and is not rescanned while that blocker is alive (one load instead of a window per round).
// The best undecided pixel always has a settled blocker, so every round settles something; reads of a neighbour's
// stale state only delay a decision.  If the round limit is hit, status = 1 sends the image back to the tiled sweeps.
struct TailArgs {
    float* cur; int2* wlist; int* slist; int* status;
    int ucap, H, W, r, max_rounds;
    const unsigned char* ubits; int wb;         // sweep 0's bitmap of undecided pixels (NmsArgs)
    const unsigned* chist; int top_k;           // top-K pruning (NmsArgs); chist null: none
    unsigned* cbits;                            // sweep 0's bitmap of confirmed maxima, as words (rows are wb bytes, a multiple of 4): the tail adds its own; null: not kept
};

constexpr int TAIL_THREADS = 1024;

template <int R>
__global__ __launch_bounds__(TAIL_THREADS) void nms_tail(TailArgs a)
{
    constexpr int NGRP = TAIL_THREADS / 16, U = 4;      // 16 lanes scan one window; U windows per group in flight
    static_assert(2 * R + 1 <= 17, "one 16-lane pass plus the centre-right column");
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, l = lane & 15, grp = tid >> 4;
    float* cur = a.cur + (size_t)img * a.H * a.W;
    int2* wl[2] = {a.wlist + (size_t)img * 2 * a.ucap, a.wlist + (size_t)img * 2 * a.ucap + a.ucap};
    int* sl = a.slist + (size_t)img * a.ucap;
    __shared__ int s_nw, s_ns, s_bin, s_wsum[TAIL_THREADS / 64];
    // the four bins of sweep 0's score histogram this thread will scan: on their way while the watch list is built
    uint4 hb = make_uint4(0, 0, 0, 0);
    const bool prune = a.chist && a.top_k > 0;
    if (prune) hb = reinterpret_cast<const uint4*>(a.chist + (size_t)img * NMS_HBINS)[tid];
    static_assert(NMS_HBINS == 4 * TAIL_THREADS, "four bins per thread");
    // r06: the watch list is built HERE from sweep 0's bitmap of undecided pixels, as (raster index, no blocker) entries: a 32-bit word (32 pixels of a
    // row) per thread, CH words per trip with their loads in flight together, one scan and one LDS atomic per wave and trip (the order of the list is
    // free).  A wave's set bits become raster indices in the wave's own LDS strip first, each lane writing at its scanned offset; the wave then reads the
    // strip back LINEARLY -- lane i takes entry i -- so that the stores to the list are whole 512-byte runs.  (First form: the lanes stored their own bits
    // straight to the list, every store instruction touching ~20 lines: +0.03 ms, profiles/r06_nms_knockouts.txt.)
    if (tid == 0) { s_nw = 0; s_ns = 0; s_bin = -1; }
    __syncthreads();
    {
        const int wq = a.wb / 4, nwords = a.H * wq, vb = (a.W + 7) / 8;     // wb is a multiple of 4 (nms_plan); bytes vb .. wb - 1 of a row are never written
        const unsigned char* ub = a.ubits + (size_t)img * a.H * a.wb;
        constexpr int CH = 4, WBUF = 640;
        __shared__ int s_strip[TAIL_THREADS / 64][WBUF];
        int* strip = s_strip[tid >> 6];
        for (int w0 = 0; w0 < nwords; w0 += CH * TAIL_THREADS) {
            unsigned um[CH];
            int base_idx[CH], c = 0;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int w = w0 + j * TAIL_THREADS + tid;
                const bool live = w < nwords;
                const int row = live ? w / wq : 0, q = live ? w - row * wq : 0;
                unsigned word = live ? *reinterpret_cast<const unsigned*>(ub + (size_t)row * a.wb + 4 * q) : 0u;
                const int nb = vb - 4 * q;
                if (nb < 4) word &= (1u << (8 * nb)) - 1u;       // the bytes past the row's end
                um[j] = word;
                base_idx[j] = row * a.W + 32 * q;
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) c += __popc(um[j]);
            const int incl = kpb_wave_incl_scan(c), tot = __builtin_amdgcn_readlane(incl, 63);
            if (tot == 0) continue;
            int b0 = 0;
            if (lane == 0) b0 = atomicAdd(&s_nw, tot);
            b0 = __builtin_amdgcn_readfirstlane(b0);
            const bool fits = tot <= WBUF;
            int slot = incl - c;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                unsigned mm = um[j];
                while (mm) {
                    const int k = __ffs((int)mm) - 1;
                    mm &= mm - 1;
                    if (fits) strip[slot] = base_idx[j] + k;
                    else if (b0 + slot < a.ucap) wl[0][b0 + slot] = make_int2(base_idx[j] + k, -1);       // a crowded wave: the slow way
                    ++slot;
                }
            }
            if (fits) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < tot; i += 64)
                    if (b0 + i < a.ucap) wl[0][b0 + i] = make_int2(strip[i], -1);
                __builtin_amdgcn_wave_barrier();        // the strip is free again
            }
        }
    }
    // Top-K pruning.  The detection outputs the top_k best survivors.  If sweep 0 has already CONFIRMED more than top_k maxima that qualify for the
    // output, let T be at or below the (top_k+1)-th best of their scores: at least top_k+1 survivors score >= T, so (i) N > top_k holds whatever the
    // undecided pixels turn out to be, and (ii) no pixel that scores below T can be among the top_k.  A pixel's fate depends only on pixels that
    // outrank it, so the undecided pixels >= T can be resolved exactly while those below T are dropped unresolved (they stay positive in the map;
    // select_topk takes confirmed maxima only).  T = the floor of the histogram bin that holds the (top_k+1)-th best: a block scan of the bin counts.
    int above = 0;
    if (prune) {
        const int mine = (int)(hb.x + hb.y + hb.z + hb.w), incl = kpb_wave_incl_scan(mine);
        if (lane == 63) s_wsum[tid >> 6] = incl;
        above = incl;
    }
    __threadfence_block();
    __syncthreads();
    const int total = s_nw;
    float Tprune = 0.0f;
    if (prune) {
        int before = 0, n = 0;
#pragma unroll
        for (int w = 0; w < TAIL_THREADS / 64; ++w) {
            const int v = s_wsum[w];
            before += w < (tid >> 6) ? v : 0;
            n += v;
        }
        if (n > a.top_k) {
            const int need = a.top_k + 1;
            const unsigned h[4] = {hb.x, hb.y, hb.z, hb.w};
            int ab = n - (before + above);      // confirmed maxima in the bins beyond this thread's four
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                if (ab < need && ab + (int)h[j] >= need) s_bin = 4 * tid + j;       // exactly one thread and bin
                ab += (int)h[j];
            }
        }
        __syncthreads();
        const int b = s_bin;
        if (b >= 0) Tprune = nms_bin_floor(b);
    }
    __syncthreads();
    int nw = total <= a.ucap ? total : 0;       // an overflowed list is incomplete (and partly unwritten): leave the image to the tiled sweeps
    const unsigned long long gmask = 0xFFFFull << (lane & 48);
    int round = 0;
    for (; round < a.max_rounds && nw > 0; ++round) {
        if (tid == 0) { s_nw = 0; s_ns = 0; }
        __syncthreads();
        const int2* src = wl[round & 1];
        int2* dst = wl[(round & 1) ^ 1];
        // phase A: watchers whose blocker is still alive stay parked, the rest are due for a scan
        for (int e = tid; e < nw; e += TAIL_THREADS) {
            const int2 w = src[e];
            if (round == 0 && Tprune > 0.0f && cur[w.x] < Tprune) continue;       // cannot reach the top_k: left unresolved
            if (w.y >= 0 && cur[w.y] > 0.0f) dst[atomicAdd(&s_nw, 1)] = w;
            else sl[atomicAdd(&s_ns, 1)] = w.x;
        }
        __threadfence_block();
        __syncthreads();
        const int ns = s_ns;
        // phase B: window scans (plain loads: the whole workgroup sits on one CU and shares its write-through L1, so the
        // other groups' decisions are visible after the barrier that ends a round)
        for (int e0 = 0; e0 < ns; e0 += NGRP * U) {
            int idx[U], y[U], x[U], blk[U];
            float v[U];
            bool have[U], dead[U], blocked[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * NGRP + grp;
                have[u] = e < ns;
                idx[u] = have[u] ? sl[e] : 0;
                y[u] = idx[u] / a.W; x[u] = idx[u] - y[u] * a.W;
                dead[u] = false; blocked[u] = false; blk[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = cur[idx[u]];          // written only by this group: never stale
            // lane l looks down column x - R + l of each window; for R = 8 lane 0 also takes the 17th column
#pragma unroll
            for (int c = 0; c < (2 * R + 1 + 15) / 16; ++c) {
                const int dx = -R + l + 16 * c;
                const bool col_ok = dx <= R && (c == 0 || l == 0);
#pragma unroll
                for (int dy = -R; dy <= R; ++dy) {
                    float q[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int yy = y[u] + dy, xx = x[u] + dx;
                        const bool in = have[u] && col_ok && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                        q[u] = in ? cur[(size_t)yy * a.W + xx] : 0.0f;
                    }
                    const bool earlier = dy < 0 || (dy == 0 && dx < 0);
                    const bool self = dy == 0 && dx == 0;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        dead[u] |= q[u] < 0.0f;
                        if (q[u] > 0.0f && !self && (earlier ? q[u] >= v[u] : q[u] > v[u])) { blocked[u] = true; blk[u] = idx[u] + dy * a.W + dx; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned long long bb = __ballot(blocked[u]) & gmask;
                const bool any_dead = (__ballot(dead[u]) & gmask) != 0;
                const int blocker = __shfl(blk[u], bb ? __ffsll((long long)bb) - 1 : 0, 64);
                if (have[u] && l == 0) {
                    if (any_dead) cur[idx[u]] = 0.0f;
                    else if (!bb) {
                        cur[idx[u]] = -v[u];
                        if (a.cbits) atomicOr(a.cbits + ((size_t)img * a.H + y[u]) * (a.wb / 4) + (x[u] >> 5), 1u << (x[u] & 31));
                    }
                    else dst[atomicAdd(&s_nw, 1)] = make_int2(idx[u], blocker);
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        nw = s_nw;
        __syncthreads();
    }
    if (tid == 0) a.status[img] = (nw > 0 || total > a.ucap) ? 1 : 0;
}

__device__ __forceinline__ unsigned f2key(float v)
{
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

constexpr int SEL_THREADS = 1024, SEL_WAVES = SEL_THREADS / 64;

// exclusive block scan of a pair of counters packed in 64 bits; returns exclusive value, total via ref
__device__ __forceinline__ unsigned long long block_scan(unsigned long long v, unsigned long long* wsum,
                                                         unsigned long long& total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SEL_WAVES; ++w) {
        const unsigned long long s = wsum[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

struct SelArgs {
    const float* map;            // [B][P] map after NMS (or the input map when nms_dist == 0)
    unsigned long long* cand;    // [B][P] scratch: (key << 32) | ~idx in raster order
    float* out_kps;              // [B][top_k][3]
    int* out_idx;                // [B][top_k] or null
    int* out_n;                  // [B]
    int H, W, border, top_k, kpad;
    float threshold, min_score;
    int signed_map;              // the NMS working map keeps confirmed maxima negated
    const int* lastchg; const int* negflag;      // NMS status words of the image (null without NMS) ...
    int* host_status;            // ... copied with the count to pinned host memory: [B][3] = (lastchg, negflag, n), or null
    int* chunk_cnt;              // [B][nchunks] two-phase form (small batches): select_scan has left chunk c's candidates at
    int nchunks;                 //   cand[c * SEL_CHUNK ...] and their number here; null: select_topk scans the map itself
    int lcap;                    // two-phase form: candidates that fit in LDS behind the kpad selection slots
    const unsigned* cbits; int wb;      // r06: the NMS's bitmap of confirmed maxima ([B][H][wb bytes], NmsArgs): read instead of the map for an image the tail has settled
};

__device__ __forceinline__ void emit(const SelArgs& a, int img, const unsigned long long* src, int n,
                                     unsigned long long* wsum)
{
    // ordered compaction with the min_score predicate (extracter.py:219-220), then (x, y, score) rows
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += SEL_THREADS) {
        const int i = c0 + threadIdx.x;
        unsigned long long ent = 0;
        bool keep = false;
        float s = 0.0f;
        if (i < n) {
            ent = src[i];
            s = key2f((unsigned)(ent >> 32));
            keep = !(a.min_score > 0.0f) || (s > a.min_score);
        }
        unsigned long long tot;
        const unsigned long long ex = block_scan(keep ? 1ull : 0ull, wsum, tot);
        if (keep) {
            const int pos = base + (int)ex;
            const unsigned idx = 0xFFFFFFFFu - (unsigned)(ent & 0xFFFFFFFFull);
            const int row = idx / a.W, col = idx - row * a.W;
            float* o = a.out_kps + ((size_t)img * a.top_k + pos) * 3;
            o[0] = __fdiv_rn((float)col + 0.5f, (float)a.W);   // extracter.py:149,158
            o[1] = __fdiv_rn((float)row + 0.5f, (float)a.H);
            o[2] = s;
            if (a.out_idx) a.out_idx[(size_t)img * a.top_k + pos] = (int)idx;
        }
        base += (int)tot;
    }
    if (threadIdx.x == 0) {
        a.out_n[img] = base;
        if (a.host_status) {
            a.host_status[3 * img + 0] = a.lastchg ? a.lastchg[img] : 0;
            a.host_status[3 * img + 1] = a.negflag ? a.negflag[img] : 0;
            a.host_status[3 * img + 2] = base;
        }
    }
}

constexpr int SEL_VPT = 16, SEL_CHUNK = SEL_THREADS * SEL_VPT;

// Border mask and raster-order compaction of map > threshold over pixels [p_begin, p_end) of one image, by the whole workgroup:
// candidates go to cand[out_base ...] as (key << 32) | ~index; returns their number.  Sixteen consecutive pixels per thread and
// round (four 16-byte loads in flight before the first use): 19 rounds of load latency + block scan per 480x640 image where four
// pixels per thread took 75.
__device__ __forceinline__ int scan_chunks(const SelArgs& a, const float* map, unsigned long long* cand, int p_begin, int p_end, int out_base,
                                           unsigned long long* wsum)
{
    const int tid = threadIdx.x, P = a.H * a.W;
    const int bx = min(max(a.border, 0), a.W), by = min(max(a.border, 0), a.H);
    int n = 0;
    constexpr int VPT = SEL_VPT;
    const bool vec = ((P & 3) == 0) && ((reinterpret_cast<uintptr_t>(map) & 15) == 0);
    float vn[VPT];                  // the next round's pixels are requested before this round's block scan (one load latency hidden per round)
    auto fetch = [&](int c0) {
        const int i0 = c0 + tid * VPT;
#pragma unroll
        for (int q = 0; q < VPT / 4; ++q) {
            const int iq = i0 + 4 * q;
            if (vec && iq + 3 < P) {
                const float4 f = *reinterpret_cast<const float4*>(map + iq);
                vn[4 * q] = f.x; vn[4 * q + 1] = f.y; vn[4 * q + 2] = f.z; vn[4 * q + 3] = f.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) vn[4 * q + j] = (iq + j < P) ? map[iq + j] : 0.0f;
            }
        }
    };
    fetch(p_begin);
    for (int c0 = p_begin; c0 < p_end; c0 += SEL_THREADS * VPT) {
        const int i0 = c0 + tid * VPT;
        float v[VPT];
#pragma unroll
        for (int j = 0; j < VPT; ++j) v[j] = vn[j];
        if (c0 + SEL_THREADS * VPT < p_end) fetch(c0 + SEL_THREADS * VPT);
        if (a.signed_map) {     // confirmed maxima are stored negated; at the fixed point nothing else is alive, and pixels the
                                // top-K pruning of nms_tail left unresolved (positive) are by construction not among the top_k
#pragma unroll
            for (int j = 0; j < VPT; ++j) v[j] = v[j] < 0.0f ? -v[j] : 0.0f;
        }
        unsigned pred = 0;
        int row = i0 / a.W, col = i0 - row * a.W;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const bool inside = (i0 + j < P) && col >= bx && col < a.W - bx && row >= by && row < a.H - by;
            if (inside && (v[j] > a.threshold)) pred |= 1u << j;
            if (++col == a.W) { col = 0; ++row; }
        }
        unsigned long long tot;
        int pos = out_base + n + (int)block_scan((unsigned long long)__popc(pred), wsum, tot);
#pragma unroll
        for (int j = 0; j < VPT; ++j)
            if ((pred >> j) & 1u) cand[pos++] = ((unsigned long long)f2key(v[j]) << 32) | (0xFFFFFFFFu - (unsigned)(i0 + j));
        n += (int)tot;
    }
    return n;
}

// r06: the same list from the NMS's bitmap of confirmed maxima (signed map, threshold <= 0: every confirmed maximum is > 0 >= threshold): 38 KB of an image
// instead of its 1.2 MB map -- the map scan is 19 rounds at the HBM rate of the whole launch (629 MB).  Four 32-pixel words per thread and round, the
// border masked per word, raster order kept by the block scan; the scores are gathered afterwards, every load independent.
__device__ __forceinline__ int scan_bits(const SelArgs& a, const float* map, const unsigned* bits, unsigned long long* cand, unsigned long long* wsum)
{
    const int tid = threadIdx.x, wq = a.wb / 4, nwords = a.H * wq, vb = (a.W + 7) / 8;
    const int bx = min(max(a.border, 0), a.W), by = min(max(a.border, 0), a.H);
    constexpr int G = 4;
    int n = 0;
    for (int c0 = 0; c0 < nwords; c0 += SEL_THREADS * G) {
        unsigned m[G];
        int base[G], cnt = 0;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int w = c0 + tid * G + j;
            const bool live = w < nwords;
            const int row = live ? w / wq : 0, q = live ? w - row * wq : 0;
            unsigned word = live ? bits[w] : 0u;
            const int nb = vb - 4 * q;                          // bytes of this word inside the row (the rest was never written)
            if (nb < 4) word = nb > 0 ? (word & ((1u << (8 * nb)) - 1u)) : 0u;
            // columns [bx, W - bx) of rows [by, H - by)
            const int lo = max(bx - 32 * q, 0), hi = min(a.W - bx - 32 * q, 32);
            unsigned keep = (hi > lo && row >= by && row < a.H - by) ? ((hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((lo >= 32) ? 0xFFFFFFFFu : ((1u << lo) - 1u))) : 0u;
            m[j] = word & keep;
            base[j] = row * a.W + 32 * q;
            cnt += __popc(m[j]);
        }
        unsigned long long tot;
        int pos = n + (int)block_scan((unsigned long long)cnt, wsum, tot);
#pragma unroll
        for (int j = 0; j < G; ++j) {
            unsigned mm = m[j];
            while (mm) {
                const int k = __ffs((int)mm) - 1;
                mm &= mm - 1;
                cand[pos++] = 0xFFFFFFFFu - (unsigned)(base[j] + k);          // the key follows
            }
        }
        n += (int)tot;
    }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 4 * SEL_THREADS) {
        unsigned low[4];
        float sc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int i = i0 + j * SEL_THREADS + tid; low[j] = i < n ? (unsigned)cand[i] : 0xFFFFFFFFu; }
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[j] = -map[0xFFFFFFFFu - low[j]];       // a confirmed maximum is stored negated
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int i = i0 + j * SEL_THREADS + tid; if (i < n) cand[i] = ((unsigned long long)f2key(sc[j]) << 32) | low[j]; }
    }
    return n;
}

// Two-phase form, phase 1 (r04): one workgroup per SEL_CHUNK pixels and image.  select_topk is one workgroup per image -- right
// when hundreds of images fill the chip, 129 us of one CU's latency (19 dependent rounds) for the single map `detection` hands
// over on the drop-in path (profiles/r04_single_pair_latency.txt).
__global__ __launch_bounds__(SEL_THREADS) void select_scan(SelArgs a)
{
    __shared__ unsigned long long wsum[SEL_WAVES];
    const int img = blockIdx.y, c = blockIdx.x, P = a.H * a.W;
    const int n = scan_chunks(a, a.map + (size_t)img * P, a.cand + (size_t)img * P, c * SEL_CHUNK, min((c + 1) * SEL_CHUNK, P), c * SEL_CHUNK, wsum);
    if (threadIdx.x == 0) a.chunk_cnt[(size_t)img * a.nchunks + c] = n;
}

// BITS (r06): the form for the launch right after sweep 0 + tail -- it reads the bitmap of confirmed maxima only; an image the tail did not settle gets no
// candidates here (its status sends it through more sweeps and the map-reading form afterwards).  0.256 -> 0.114 ms per 512 images; at two workgroups per CU (64 registers, 17 spilled) 0.111: not taken.
template <bool BITS>
__global__ __launch_bounds__(SEL_THREADS) void select_topk(SelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* sel = reinterpret_cast<unsigned long long*>(smem);   // [kpad]
    __shared__ unsigned long long wsum[SEL_WAVES];
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_krem;

    const int img = blockIdx.x, tid = threadIdx.x;
    const int P = a.H * a.W;
    const float* map = a.map + (size_t)img * P;
    unsigned long long* cand = a.cand + (size_t)img * P;

    // A2 + A3: border mask and raster-order compaction of map > threshold
    int n = 0;
    const unsigned long long* cl = cand;        // the raster-ordered candidate list the selection below reads
    if constexpr (BITS) {
        if (a.lastchg[img] == 0) n = scan_bits(a, map, a.cbits + (size_t)img * a.H * (a.wb / 4), cand, wsum);
        __syncthreads();
    } else if (a.chunk_cnt) {
        // two-phase form: the scan ran as select_scan on (nchunks x batch) workgroups and left chunk c's list at cand[c * SEL_CHUNK].
        const int* cc = a.chunk_cnt + (size_t)img * a.nchunks;
        __shared__ int s_off[65];
        bool local = a.nchunks <= 64 && a.lcap > 0;
        if (local) {
            if (tid < 64) {
                const int c = tid < a.nchunks ? cc[tid] : 0;
                int inc = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (tid >= o) inc += v; }
                s_off[tid + 1] = inc;
                if (tid == 0) s_off[0] = 0;
            }
            __syncthreads();
            n = s_off[a.nchunks];
            local = n <= a.lcap;
        }
        if (local) {
            // The lists are gathered into LDS behind `sel` by all threads at once (independent loads: a few round trips).  Closing
            // them up in place, chunk after chunk, was 19 dependent read-synchronise-write steps: 28 of select_topk's 40 us on the
            // single map of the drop-in path; the radix passes below then read LDS instead of global memory.
            unsigned long long* lc = sel + a.kpad;
            for (int i = tid; i < n; i += SEL_THREADS) {
                int lo = 0, hi = a.nchunks - 1;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid - 1; }
                lc[i] = cand[(size_t)lo * SEL_CHUNK + (i - s_off[lo])];
            }
            cl = lc;
        } else {
            // too many candidates (or chunks) for LDS: the lists are closed up into one at the front of cand.  Every move goes LEFT
            // (a chunk never holds more than its SEL_CHUNK pixels), so a block-wide step may read, synchronise, write: its
            // destination ends where the next step's source begins at the latest.
            n = 0;
            for (int c = 0; c < a.nchunks; ++c) {
                const int cnt = cc[c];
                const unsigned long long* src = cand + (size_t)c * SEL_CHUNK;
                if (n != c * SEL_CHUNK)
                    for (int j0 = 0; j0 < cnt; j0 += SEL_THREADS) {
                        const int j = j0 + tid;
                        const unsigned long long v = j < cnt ? src[j] : 0ull;
                        __syncthreads();
                        if (j < cnt) cand[n + j] = v;
                        __syncthreads();
                    }
                n += cnt;
            }
        }
        __syncthreads();
    } else {
        n = scan_chunks(a, map, cand, 0, P, 0, wsum);
        __syncthreads();
    }

    if (n <= a.top_k) {  // raster order kept (extracter.py:217)
        emit(a, img, cl, n, wsum);
        return;
    }

    // A4: radix select of the top_k-th largest key, 8 bits per pass
    if (tid == 0) { s_prefix = 0; s_krem = (unsigned)a.top_k; }
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < n; i += SEL_THREADS) {
            const unsigned k = (unsigned)(cl[i] >> 32);
            if ((k & himask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        // the digit d whose bin holds the krem-th largest key: the largest d with  sum(hist[d..255]) >= krem  (digit 0 if none).
        // Suffix sums by 256 threads (a wave scan and four wave totals); r03 let thread 0 walk the bins one LDS read at a time --
        // up to 256 dependent reads in each of the four passes, most of select_topk's 60 us on a single map.
        {
            const unsigned krem = s_krem;
            unsigned suf = 0, own = 0;
            if (tid < 256) {
                own = hist[255 - tid];                      // thread t looks at digit 255 - t: prefix sums over t = suffix sums over d
                suf = own;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(suf, o, 64); if ((tid & 63) >= o) suf += v; }
                if ((tid & 63) == 63) wsum[tid >> 6] = suf;
            }
            __syncthreads();
            if (tid < 256) {
                for (int w = 0; w < (tid >> 6); ++w) suf += (unsigned)wsum[w];
                const unsigned before = suf - own;          // keys in the bins above this digit
                const int d = 255 - tid;
                if ((before < krem && suf >= krem) || (d == 0 && suf < krem)) {
                    s_krem = krem - before;
                    s_prefix = prefix | ((unsigned)d << shift);
                }
            }
        }
        __syncthreads();
    }
    const unsigned T = s_prefix, krem = s_krem;   // take every key > T and the first krem keys == T

    for (int i = tid; i < a.kpad; i += SEL_THREADS) sel[i] = 0ull;
    __syncthreads();
    unsigned gt_base = 0, eq_base = 0;
    for (int c0 = 0; c0 < n; c0 += SEL_THREADS) {
        const int i = c0 + tid;
        unsigned long long ent = 0;
        bool gt = false, eq = false;
        if (i < n) {
            ent = cl[i];
            const unsigned k = (unsigned)(ent >> 32);
            gt = k > T;
            eq = k == T;
        }
        unsigned long long tot;
        const unsigned long long ex = block_scan((gt ? 1ull : 0ull) | (eq ? (1ull << 32) : 0ull), wsum, tot);
        const unsigned gtb = gt_base + (unsigned)(ex & 0xFFFFFFFFull), eqb = eq_base + (unsigned)(ex >> 32);
        if (gt || (eq && eqb < krem)) sel[gtb + min(eqb, krem)] = ent;
        gt_base += (unsigned)(tot & 0xFFFFFFFFull);
        eq_base += (unsigned)(tot >> 32);
    }
    __syncthreads();

    // bitonic sort, descending on (key, ~idx): score descending, ties ascending raster index
    for (int k = 2; k <= a.kpad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < a.kpad; i += SEL_THREADS) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long x = sel[i], y = sel[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { sel[i] = y; sel[l] = x; }
                }
            }
            __syncthreads();
        }
    }
    emit(a, img, sel, a.top_k, wsum);
}

int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

int env_int(const char* name, int dflt)
{
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
}

struct NmsPlan {
    int tiles_x, tiles_y, ntiles;
    size_t lds;
    int* tchg[2];
    int* lastchg;
    int* negflag;
    int2* ulist;        // sparse tail (r <= 8): [B][2][ucap] watch lists (ping-pong)
    int* slist;         // [B][ucap] pixels due for a window scan
    int ucap;
    unsigned* chist;    // [B][NMS_HBINS] score histogram of the confirmed maxima (top-K pruning); null when off
    int prune_k, border;
    float cmin;
    unsigned char* ubits; int wb;       // [B][H][wb] sweep 0's bitmap of undecided pixels (NmsArgs)
    unsigned char* cbits;               // [B][H][wb] confirmed maxima (NmsArgs)
    size_t nclear;      // ints at lastchg that nms_open clears: lastchg, negflag and the histograms
};

int nms_plan(kpb_ctx* ctx, int batch, int H, int W, int r, NmsPlan& p, int prune_k = 0, int border = 0, float cmin = 0.0f)
{
    p.tiles_x = cdiv(W, TW);
    p.tiles_y = cdiv(H, TH);
    p.ntiles = p.tiles_x * p.tiles_y;
    const int LH = TH + 4 * r, LW = TW + 4 * r;
    p.lds = (size_t)2 * LH * LW * sizeof(float) + (MAXLIST + 4) * sizeof(int);
    const size_t nflag = (size_t)batch * p.ntiles;
    // The tail is one workgroup per image and latency-bound (about 1.4 ms whatever the batch); four more tiled sweeps
    // cost about 8 us per 480x640 image.  It pays once the batch fills the chip.  KPB_NMS_TILED=1 / =0 force a choice.
    const int tiled = env_int("KPB_NMS_TILED", -1);
    const bool big = (size_t)batch * H * W >= (size_t)192 * 480 * 640;
    const bool tail = r >= 1 && r <= 8 && (tiled == 0 || (tiled < 0 && big));
    const bool prune = tail && prune_k > 0 && prune_k < H * W && env_int("KPB_NMS_PRUNE", 1);
    const size_t head = (2 * (size_t)batch + 3) & ~(size_t)3;        // lastchg[B], negflag[B]; the histograms start on 16 bytes (nms_tail reads uint4s)
    p.nclear = head + (prune ? (size_t)batch * NMS_HBINS : 0);
    if (int rc = kpb_reserve(ctx, ctx->ws_nms_state, (p.nclear + 2 * nflag) * sizeof(int))) return rc;
    int* base = static_cast<int*>(ctx->ws_nms_state.p);
    p.lastchg = base;
    p.negflag = base + batch;
    p.chist = prune ? reinterpret_cast<unsigned*>(base + head) : nullptr;
    p.tchg[0] = base + p.nclear;
    p.tchg[1] = p.tchg[0] + nflag;
    p.ulist = nullptr;
    p.ucap = 0;
    p.prune_k = prune ? prune_k : 0; p.border = border; p.cmin = cmin;
    p.ubits = nullptr; p.cbits = nullptr; p.wb = 0;
    if (tail) {
        p.ucap = std::max(4096, H * W / 8);
        p.wb = (cdiv(W, 8) + 3) & ~3;
        const size_t nbits = (size_t)batch * H * p.wb;       // (a multiple of 4: wb is)
        if (int rc = kpb_reserve(ctx, ctx->ws_nms_list, (size_t)batch * 5 * (size_t)p.ucap * sizeof(int) + 2 * nbits)) return rc;
        p.ulist = static_cast<int2*>(ctx->ws_nms_list.p);
        p.slist = reinterpret_cast<int*>(p.ulist + (size_t)batch * 2 * p.ucap);
        p.ubits = reinterpret_cast<unsigned char*>(p.slist + (size_t)batch * p.ucap);
        p.cbits = p.ubits + nbits;
    }
    if (!(ctx->lds_attr_done & KPB_ATTR_NMS)) {
        KPB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sweep),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        KPB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(select_topk<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        ctx->lds_attr_done |= KPB_ATTR_NMS;
    }
    return KPB_OK;
}

int nms_launch(kpb_ctx* ctx, const NmsPlan& p, const float* src, float* cur, int batch, int H, int W, int r,
               int first_sweep, int nsweeps)
{
    for (int s = first_sweep; s < first_sweep + nsweeps; ++s) {
        NmsArgs a;
        a.src = src; a.cur = cur;
        a.tchg_prev = p.tchg[(s + 1) & 1];
        a.tchg_cur = p.tchg[s & 1];
        a.lastchg = p.lastchg; a.negflag = p.negflag;
        a.ubits = p.ubits; a.cbits = p.cbits; a.wb = p.wb; a.chist = p.chist; a.border = p.border; a.cmin = p.cmin;
        a.H = H; a.W = W; a.r = r; a.tiles_y = p.tiles_y; a.tiles_x = p.tiles_x;
        a.sweep = s;
        a.xcd_map = 1;      // -1 % (profiles/r04_ab_knobs.txt)
        // with the sparse tail behind it, sweep 0 stops after three in-tile rounds (99.6 % of an ALIKE map is settled by
        // then; measured: 3 rounds 2.26 + 1.49 ms, 5 rounds 2.75 + 1.39 ms, 2 rounds 1.87 + 2.52 ms per 512 images)
        // with top-K pruning the tail is cheap and two rounds are the optimum (r02: 2 rounds 2.19 + <0.2 ms, 3 rounds 2.57 + <0.2 ms;
        // one round confirms too few maxima for the bound and the tail overflows)
        a.max_local = (p.ulist && s == 0) ? (p.chist ? 2 : 3) : 64;
        const dim3 grid(p.ntiles, batch), block(NMS_THREADS);
        switch (r) {
        case 1: KPB_LAUNCH(ctx, "nms_sweep", nms_sweep_r<1>, grid, block, 0, ctx->stream, a); break;
        case 2: KPB_LAUNCH(ctx, "nms_sweep", nms_sweep_r<2>, grid, block, 0, ctx->stream, a); break;
        case 3: KPB_LAUNCH(ctx, "nms_sweep", nms_sweep_r<3>, grid, block, 0, ctx->stream, a); break;
        case 4: KPB_LAUNCH(ctx, "nms_sweep", nms_sweep_r<4>, grid, block, 0, ctx->stream, a); break;
        case 5: KPB_LAUNCH(ctx, "nms_sweep", nms_sweep_r<5>, grid, block, 0, ctx->stream, a); break;
        case 6: KPB_LAUNCH(ctx, "nms_sweep", nms_sweep_r<6>, grid, block, 0, ctx->stream, a); break;
        case 7: KPB_LAUNCH(ctx, "nms_sweep", nms_sweep_r<7>, grid, block, 0, ctx->stream, a); break;
        case 8: KPB_LAUNCH(ctx, "nms_sweep", nms_sweep_r<8>, grid, block, 0, ctx->stream, a); break;
        default: KPB_LAUNCH(ctx, "nms_sweep", nms_sweep, grid, block, p.lds, ctx->stream, a); break;
        }
    }
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}

// The opening of every NMS run: clear the per-image state, sweep 0 and -- for the specialised radii -- the sparse tail,
// which leaves lastchg[img] = 0 (settled) or 1 (round limit or list overflow: the caller's loop of tiled sweeps takes
// over, with every tile marked as changed).  Returns the number of sweeps the status check has to account for.
int nms_open(kpb_ctx* ctx, const NmsPlan& p, const float* src, float* cur, int batch, int H, int W, int r, int chunk, int& sweeps_run)
{
    if ((size_t)H * W >= ((size_t)1 << 30))       // nms_sweep_r addresses a pixel by a 32-bit BYTE offset from its image
        return kpb_fail(ctx, KPB_E_UNSUPPORTED, "NMS: a %d x %d map is too large (2^30 pixels per image at most)", H, W);
    KPB_HIP(ctx, hipMemsetAsync(p.lastchg, 0, p.nclear * sizeof(int), ctx->stream));
    if (!p.ulist) {
        sweeps_run = chunk;
        return nms_launch(ctx, p, src, cur, batch, H, W, r, 0, chunk);
    }
    if (int rc = nms_launch(ctx, p, src, cur, batch, H, W, r, 0, 1)) return rc;
    TailArgs t{cur, p.ulist, p.slist, p.lastchg, p.ucap, H, W, r, env_int("KPB_NMS_TAIL_ROUNDS", 256), p.ubits, p.wb, p.chist, p.prune_k,
               reinterpret_cast<unsigned*>(p.cbits)};
    switch (r) {
    case 1: KPB_LAUNCH(ctx, "nms_tail", nms_tail<1>, dim3(batch), dim3(TAIL_THREADS), 0, ctx->stream, t); break;
    case 2: KPB_LAUNCH(ctx, "nms_tail", nms_tail<2>, dim3(batch), dim3(TAIL_THREADS), 0, ctx->stream, t); break;
    case 3: KPB_LAUNCH(ctx, "nms_tail", nms_tail<3>, dim3(batch), dim3(TAIL_THREADS), 0, ctx->stream, t); break;
    case 4: KPB_LAUNCH(ctx, "nms_tail", nms_tail<4>, dim3(batch), dim3(TAIL_THREADS), 0, ctx->stream, t); break;
    case 5: KPB_LAUNCH(ctx, "nms_tail", nms_tail<5>, dim3(batch), dim3(TAIL_THREADS), 0, ctx->stream, t); break;
    case 6: KPB_LAUNCH(ctx, "nms_tail", nms_tail<6>, dim3(batch), dim3(TAIL_THREADS), 0, ctx->stream, t); break;
    case 7: KPB_LAUNCH(ctx, "nms_tail", nms_tail<7>, dim3(batch), dim3(TAIL_THREADS), 0, ctx->stream, t); break;
    default: KPB_LAUNCH(ctx, "nms_tail", nms_tail<8>, dim3(batch), dim3(TAIL_THREADS), 0, ctx->stream, t); break;
    }
    KPB_HIP(ctx, hipMemsetAsync(p.tchg[0], 1, (size_t)batch * p.ntiles * sizeof(int), ctx->stream));   // "previous" flags of sweep 1
    KPB_HIP(ctx, hipGetLastError());
    sweeps_run = 1;
    return KPB_OK;
}

// reads lastchg/negflag back; returns number of images that changed in sweep (sweeps_run - 1), sets neg
int nms_status(kpb_ctx* ctx, const NmsPlan& p, int batch, int sweeps_run, int& pending, int& neg)
{
    std::vector<int> h(2 * (size_t)batch);
    KPB_HIP(ctx, hipMemcpyAsync(h.data(), p.lastchg, h.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    KPB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pending = 0; neg = 0;
    for (int b = 0; b < batch; ++b) {
        pending += (h[b] >= sweeps_run);
        neg |= h[batch + b];
    }
    return KPB_OK;
}

}  // namespace

namespace {
__global__ void abs_inplace(float* m, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) m[i] = fabsf(m[i]);
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int kpb_fast_nms(kpb_ctx* ctx, const float* score_dev, int batch, int H, int W, int nms_dist,
                            float* out_map_dev)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_fast_nms: null context");
    if (!score_dev || !out_map_dev || batch <= 0 || H <= 0 || W <= 0)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_fast_nms: bad argument");
    if (nms_dist < 0 || nms_dist > KPB_MAX_NMS_DIST)
        return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_fast_nms: nms_dist %d outside 0..%d", nms_dist, KPB_MAX_NMS_DIST);
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)batch * H * W * sizeof(float);
    if (nms_dist == 0) {  // extracter.py:40-41
        KPB_HIP(ctx, hipMemcpyAsync(out_map_dev, score_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        KPB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return KPB_OK;
    }
    NmsPlan p;
    if (int rc = nms_plan(ctx, batch, H, W, nms_dist, p)) return rc;
    const int chunk = 6;
    int run = 0, pending = 1, neg = 0;
    while (pending) {
        if (run == 0) {
            if (int rc = nms_open(ctx, p, score_dev, out_map_dev, batch, H, W, nms_dist, chunk, run)) return rc;
        } else {
            if (int rc = nms_launch(ctx, p, score_dev, out_map_dev, batch, H, W, nms_dist, run, chunk)) return rc;
            run += chunk;
        }
        if (int rc = nms_status(ctx, p, batch, run, pending, neg)) return rc;
        if (neg) return kpb_fail(ctx, KPB_E_NEGATIVE, "kpb_fast_nms: negative scores are outside this path's contract");
        if (run > 100000) return kpb_fail(ctx, KPB_E_NOT_CONVERGED, "kpb_fast_nms: no fixed point after %d sweeps", run);
    }
    if (nms_dist <= 8) {   // the specialised sweep keeps confirmed maxima negated in its working map
        const size_t n = (size_t)batch * H * W;
        KPB_LAUNCH(ctx, "abs_inplace", abs_inplace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, out_map_dev, n);
        KPB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return KPB_OK;
}

namespace {
struct DetState {
    NmsPlan plan;
    const float* score; float* cur; unsigned long long* cand;
    int batch, H, W; kpb_detect_params prm;
    float* out_kps; int* out_idx; int* out_n;
    int sweeps_run;
};

DetState& det_state(kpb_ctx* ctx)
{
    if (!ctx->det_state) {
        ctx->det_state = new DetState();
        ctx->det_state_free = [](void* p) { delete static_cast<DetState*>(p); };
    }
    return *static_cast<DetState*>(ctx->det_state);
}

int det_select(kpb_ctx* ctx, const DetState& d)
{
    SelArgs s;
    s.map = d.prm.nms_dist == 0 ? d.score : d.cur;
    s.cand = d.cand;
    s.out_kps = d.out_kps; s.out_idx = d.out_idx; s.out_n = d.out_n;
    s.H = d.H; s.W = d.W; s.border = d.prm.border_dist; s.top_k = d.prm.top_k;
    s.kpad = d.prm.top_k >= d.H * d.W ? 0 : next_pow2(d.prm.top_k);
    s.threshold = d.prm.threshold; s.min_score = d.prm.min_score;
    s.signed_map = (d.prm.nms_dist >= 1 && d.prm.nms_dist <= 8) ? 1 : 0;
    s.lastchg = d.prm.nms_dist > 0 ? d.plan.lastchg : nullptr; s.negflag = d.prm.nms_dist > 0 ? d.plan.negflag : nullptr;
    if (ctx->host_det_cap < 3 * d.batch) {
        KPB_HIP(ctx, hipStreamSynchronize(ctx->stream));        // an earlier call's select_topk may still be writing the old mirror
        if (ctx->host_det) KPB_HIP(ctx, hipHostFree(ctx->host_det));
        ctx->host_det = nullptr; ctx->host_det_cap = 0;
        KPB_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->host_det), (size_t)3 * d.batch * sizeof(int), hipHostMallocDefault));
        ctx->host_det_cap = 3 * d.batch;
    }
    s.host_status = ctx->host_det;
    s.chunk_cnt = nullptr; s.nchunks = cdiv(d.H * d.W, SEL_CHUNK); s.lcap = 0;
    // the bitmap of confirmed maxima stands for the map right after sweep 0 + tail only (later sweeps do not keep it), and only for thresholds every maximum passes
    const bool bits = d.plan.cbits && d.sweeps_run == 1 && !(d.prm.threshold > 0.0f) && s.signed_map;
    s.cbits = bits ? reinterpret_cast<const unsigned*>(d.plan.cbits) : nullptr; s.wb = d.plan.wb;
    size_t lds = (size_t)s.kpad * sizeof(unsigned long long);
    if (d.batch < 64 && s.nchunks > 1) {       // too few images to fill the chip with one workgroup each: scan in (chunks x batch) workgroups first
        if (int rc = kpb_reserve(ctx, ctx->ws_sel, (size_t)d.batch * s.nchunks * sizeof(int))) return rc;
        s.chunk_cnt = static_cast<int*>(ctx->ws_sel.p);
        s.lcap = (int)((64 * 1024 - lds) / sizeof(unsigned long long));        // (the attribute set in nms_plan allows 64 KB)
        lds = 64 * 1024;
        KPB_LAUNCH(ctx, "select_scan", select_scan, dim3(s.nchunks, d.batch), dim3(SEL_THREADS), 0, ctx->stream, s);
    }
    if (bits && !s.chunk_cnt) KPB_LAUNCH(ctx, "select_topk", select_topk<true>, dim3(d.batch), dim3(SEL_THREADS), lds, ctx->stream, s);
    else KPB_LAUNCH(ctx, "select_topk", select_topk<false>, dim3(d.batch), dim3(SEL_THREADS), lds, ctx->stream, s);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int kpb_detect(kpb_ctx* ctx, const float* score_dev, int batch, int H, int W,
                          const kpb_detect_params* prm, float* out_kps_dev, int32_t* out_idx_dev,
                          int32_t* out_n_dev, int sync)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_detect: null context");
    ctx->det_counts_valid = 0;          // whatever happens below, the mirror no longer belongs to a completed detection
    if (!score_dev || !prm || !out_kps_dev || !out_n_dev || batch <= 0 || H <= 0 || W <= 0)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_detect: bad argument");
    if (prm->nms_dist < 0 || prm->nms_dist > KPB_MAX_NMS_DIST)
        return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_detect: nms_dist %d outside 0..%d", prm->nms_dist, KPB_MAX_NMS_DIST);
    if ((size_t)H * W >= (1u << 31)) return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_detect: map too large");
    if (prm->top_k <= 0 || (prm->top_k > KPB_MAX_TOPK && prm->top_k < H * W))
        return kpb_fail(ctx, KPB_E_UNSUPPORTED, "kpb_detect: top_k %d outside 1..%d (or >= H*W)", prm->top_k, KPB_MAX_TOPK);
    if (ctx->det_pending)   // one DetState per context: a second enqueue would drop the first call's convergence / sign check
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_detect: the previous kpb_detect(sync=0) has not been completed by kpb_detect_check "
                                            "(its score map and outputs must stay alive until then)");
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)H * W;
    DetState& d = det_state(ctx);
    d.score = score_dev; d.batch = batch; d.H = H; d.W = W; d.prm = *prm;
    if (d.prm.top_k > H * W) d.prm.top_k = H * W;   // N can never exceed H*W: same rows, smaller buffers
    d.out_kps = out_kps_dev; d.out_idx = out_idx_dev; d.out_n = out_n_dev;
    if (int rc = kpb_reserve(ctx, ctx->ws_cand, (size_t)batch * P * sizeof(unsigned long long))) return rc;
    d.cand = static_cast<unsigned long long*>(ctx->ws_cand.p);
    {   // a confirmed maximum counts toward the pruning bound only if the detection could output it
        const float cmin = std::max(prm->threshold, prm->min_score > 0.0f ? prm->min_score : prm->threshold);
        const int border = std::max(prm->border_dist, 0);
        if (int rc = nms_plan(ctx, batch, H, W, prm->nms_dist, d.plan, d.prm.top_k, border, cmin)) return rc;
    }
    d.sweeps_run = 0;
    d.cur = nullptr;
    if (prm->nms_dist > 0) {
        if (int rc = kpb_reserve(ctx, ctx->ws_nms_map, (size_t)batch * P * sizeof(float))) return rc;
        d.cur = static_cast<float*>(ctx->ws_nms_map.p);
        // tiled sweeps enqueued before the first look at the status: an ALIKE map is at its fixed point after 3-4 of them; a sweep
        // with nothing to do still costs its launch (5 us each on the single map of the drop-in path), an early look costs a
        // synchronisation -- 6 for batches (amortised), 4 when a few images are all there is
        // (r04, 96 single maps -- synthetic pairs and warped views: none is confirmed at its fixed point by three sweeps, 86 are by
        //  four, the other 10 take a second chunk)
        const int chunk = batch >= 16 ? 6 : 4;
        if (int rc = nms_open(ctx, d.plan, score_dev, d.cur, batch, H, W, prm->nms_dist, chunk, d.sweeps_run)) return rc;
    }
    if (int rc = det_select(ctx, d)) return rc;
    ctx->det_pending = 1;
    if (!sync) return KPB_OK;
    const int rc = kpb_detect_check(ctx);
    return rc > 0 ? KPB_OK : rc;
}

extern "C" __attribute__((visibility("default"))) int kpb_detect_check(kpb_ctx* ctx)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_detect_check: null context");
    if (!ctx->det_pending) return KPB_OK;
    DetState& d = det_state(ctx);
    KPB_HIP(ctx, hipSetDevice(ctx->device));
    if (d.prm.nms_dist == 0) {
        KPB_HIP(ctx, kpb_wait_stream(ctx, d.batch < 16));
        ctx->det_pending = 0;
        ctx->det_counts_valid = 1;
        return KPB_OK;
    }
    const int chunk = 6;
    int rerun = 0;
    for (;;) {
        int pending = 0, neg = 0;
        KPB_HIP(ctx, kpb_wait_stream(ctx, d.batch < 16));        // select_topk has written every image's status to pinned host memory
        for (int b = 0; b < d.batch; ++b) {
            pending += (ctx->host_det[3 * b] >= d.sweeps_run);
            neg |= ctx->host_det[3 * b + 1];
        }
        if (neg) {
            ctx->det_pending = 0;
            return kpb_fail(ctx, KPB_E_NEGATIVE, "kpb_detect: negative scores are outside this path's contract "
                                                 "(the reference's zero padding makes them data dependent)");
        }
        if (!pending) break;
        rerun = 1;
        if (d.sweeps_run > 100000) {
            ctx->det_pending = 0;
            return kpb_fail(ctx, KPB_E_NOT_CONVERGED, "kpb_detect: NMS did not converge");
        }
        if (int rc = nms_launch(ctx, d.plan, d.score, d.cur, d.batch, d.H, d.W, d.prm.nms_dist, d.sweeps_run, chunk)) return rc;
        d.sweeps_run += chunk;
        if (int rc = det_select(ctx, d)) return rc;
    }
    ctx->det_pending = 0;
    ctx->det_counts_valid = 1;
    return rerun;   // 1: the outputs were rewritten after extra sweeps (the loop's last synchronisation covers them)
}

extern "C" __attribute__((visibility("default"))) int kpb_detect_counts(kpb_ctx* ctx, int32_t* out_n_host, int batch)
{
    if (!ctx) return kpb_fail(nullptr, KPB_E_INVALID, "kpb_detect_counts: null context");
    if (!out_n_host || batch <= 0) return kpb_fail(ctx, KPB_E_INVALID, "kpb_detect_counts: bad argument");
    if (ctx->det_pending) return kpb_fail(ctx, KPB_E_INVALID, "kpb_detect_counts: the last kpb_detect has not been completed (kpb_detect_check)");
    if (!ctx->det_state || !ctx->host_det || !ctx->det_counts_valid)
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_detect_counts: no completed detection (the last kpb_detect failed, was rejected, or none has run)");
    const DetState& d = det_state(ctx);
    if (batch != d.batch) return kpb_fail(ctx, KPB_E_INVALID, "kpb_detect_counts: the last kpb_detect had %d images, not %d", d.batch, batch);
    for (int b = 0; b < batch; ++b) out_n_host[b] = ctx->host_det[3 * b + 2];
    return KPB_OK;
}
