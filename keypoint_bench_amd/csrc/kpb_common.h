// kpb_common.h -- context, error handling and workspace arena shared by the libkpb.so translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>
#include <type_traits>

#include "../../include/kpb.h"
#include <chrono>

// The two 32-lane halves of a wave exchanged on the vector ALU (gfx950's v_permlane32_swap): `lo` = the value of lane & 31 in
// every lane, `hi` = that of lane | 32.  A commutative op(lo, hi) equals op(v, __shfl_xor(v, 32)) bit for bit in every lane,
// without the ds_bpermute round trip through the LDS crossbar.
__device__ __forceinline__ void kpb_halves32(float v, float& lo, float& hi)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
// XCD-aware tile map (r04).  Workgroups are dealt round-robin over the 8 XCDs in dispatch order (x fastest, then y, then z), so the
// linear ids L and L + 8 share an L2 and L, L + 1 never do: two tiles that are neighbours in the image -- and re-read each other's
// halo lines -- meet in eight DIFFERENT L2s under the plain blockIdx map.  Here XCD k works through the k-th contiguous eighth of the
// tile space in raster order, so a tile's neighbours run on the same XCD at about the same time and the halo lines are L2 hits.
// Placement is speed only (HIP promises none): any bijection of the tile space is correct.  Identity when the tile count is not a
// multiple of 8.
struct kpb_tile3 { int x, y, z; };
__device__ __forceinline__ kpb_tile3 kpb_xcd_tile(int enable)
{
    const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if (enable && (total & 7u) == 0u) L = (L & 7u) * (total >> 3) + (L >> 3);
    kpb_tile3 t;
    t.x = (int)(L % gx); t.y = (int)((L / gx) % gy); t.z = (int)(L / (gx * gy));
    return t;
}

// maximum of two NON-NEGATIVE floats (ReLU outputs, magnitudes) on the integer ALU: their bit patterns order like the values, and no
// canonicalising `v_max_f32 x, x` is spent on operands the compiler cannot prove quiet (8 % of block 1's vector instructions).
// PRECONDITION: sign bit clear (callers pass fabsf(x) or a ReLU output: v_max_f32(-0, +0) = +0) -- a -0.0 or a negative value would
// win against every positive one.  NaN and +inf order ABOVE every finite value (fmaxf would drop a NaN): where the result feeds a
// split scale (cm_exp_of) the caller checks it for finiteness and takes the masked path (alike_block1_h) or skips it (amax_reduce).
__device__ __forceinline__ float kpb_pmax(float a, float b) { return __uint_as_float(max(__float_as_uint(a), __float_as_uint(b))); }
__device__ __forceinline__ float kpb_pmax32(float v) { float a, b; kpb_halves32(v, a, b); return kpb_pmax(a, b); }
// likewise for lanes 16 apart (v_permlane16_swap: rows 1 / 3 of one operand against rows 0 / 2 of the other) and, for the quad
// neighbours lane ^ 1 and lane ^ 2, DPP quad permutes -- none of which the compiler derives from __shfl_xor (it emits ds_bpermute)
__device__ __forceinline__ float kpb_sum16(float v)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float kpb_xor1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float kpb_xor2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
// __shfl_xor(v, N, 64) for the six butterfly distances without the LDS crossbar: the value of lane ^ N in every lane (bit-identical
// to the shuffle, so any reduction written as `for (o = 32; o; o >>= 1) v = op(v, kpb_shfl_xor<o>(v))` keeps its result).
//   1, 2: DPP quad permutes;  4: two row shifts, each written into the quads (banks) it is valid for;  8: row rotate by 8;
//   16, 32: v_permlane16_swap / v_permlane32_swap give both halves, the lane keeps the one that is not its own.
template <int N>
__device__ __forceinline__ float kpb_shfl_xor(float v);
template <int N>
__device__ __forceinline__ int kpb_shfl_xor(int v) { return __float_as_int(kpb_shfl_xor<N>(__int_as_float(v))); }
template <int N>
__device__ __forceinline__ float kpb_shfl_xor(float v)
{
    static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16 || N == 32, "butterfly distances only");
    const int x = __float_as_int(v);
    if constexpr (N == 1) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));
    else if constexpr (N == 2) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));
    else if constexpr (N == 4) {
        const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);      // row_shl:4 into quads 0 and 2: lane i reads lane i + 4
        return __int_as_float(__builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false));     // row_shr:4 into quads 1 and 3: lane i reads lane i - 4
    } else if constexpr (N == 8) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false));   // row_ror:8
    else if constexpr (N == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);      // r[0] = rows (0, 0, 2, 2), r[1] = rows (1, 1, 3, 3)
        return __uint_as_float((__lane_id() & 16) ? r[0] : r[1]);
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);      // r[0] = low half twice, r[1] = high half twice
        return __uint_as_float((__lane_id() & 32) ? r[0] : r[1]);
    }
}
// f(integral_constant<int, o>) for o = 32, 16, 8, 4, 2, 1: the loop `for (o = 32; o; o >>= 1)` with o a compile-time constant
template <class F>
__device__ __forceinline__ void kpb_butterfly(F&& f)
{
    f(std::integral_constant<int, 32>{}); f(std::integral_constant<int, 16>{}); f(std::integral_constant<int, 8>{});
    f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 2>{}); f(std::integral_constant<int, 1>{});
}
// the butterflies themselves (same pairing and order as the __shfl_xor loops they replace)
__device__ __forceinline__ float kpb_wave_sum(float v)
{
    v += kpb_shfl_xor<32>(v); v += kpb_shfl_xor<16>(v); v += kpb_shfl_xor<8>(v);
    v += kpb_shfl_xor<4>(v); v += kpb_shfl_xor<2>(v); v += kpb_shfl_xor<1>(v);
    return v;
}
__device__ __forceinline__ float kpb_wave_fmax(float v)
{
    v = fmaxf(v, kpb_shfl_xor<32>(v)); v = fmaxf(v, kpb_shfl_xor<16>(v)); v = fmaxf(v, kpb_shfl_xor<8>(v));
    v = fmaxf(v, kpb_shfl_xor<4>(v)); v = fmaxf(v, kpb_shfl_xor<2>(v)); v = fmaxf(v, kpb_shfl_xor<1>(v));
    return v;
}
__device__ __forceinline__ float kpb_wave_fmin(float v)
{
    v = fminf(v, kpb_shfl_xor<32>(v)); v = fminf(v, kpb_shfl_xor<16>(v)); v = fminf(v, kpb_shfl_xor<8>(v));
    v = fminf(v, kpb_shfl_xor<4>(v)); v = fminf(v, kpb_shfl_xor<2>(v)); v = fminf(v, kpb_shfl_xor<1>(v));
    return v;
}
// inclusive prefix sum over the 64 lanes of a wave on the vector ALU (DPP row shifts inside the rows of 16, then the two row broadcasts): lane i
// gets v[0] + ... + v[i]; lane 63's value is the wave total.  Twelve instructions, no LDS.
__device__ __forceinline__ int kpb_wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);      // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ float kpb_max32(float v) { float a, b; kpb_halves32(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float kpb_min32(float v) { float a, b; kpb_halves32(v, a, b); return fminf(a, b); }
__device__ __forceinline__ float kpb_sum32(float v) { float a, b; kpb_halves32(v, a, b); return a + b; }

struct kpb_buf {
    void* p = nullptr;
    size_t cap = 0;
};

struct kpb_prof_rec {
    const char* name;
    hipEvent_t e0, e1;
};

struct kpb_ctx {
    int device = 0;
    // per-kernel timing with HIP events on the launch stream (bench.py's roofline leg); off by default
    bool prof = false;
    std::vector<kpb_prof_rec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    std::set<std::string> prof_names;   // interned: record names outlive the (possibly temporary) strings of the callers
    hipStream_t stream = nullptr;
    bool own_stream = false;
    char err[512] = {0};
    // named workspaces, grown on demand (first call of a shape = warm-up), never freed before destroy
    kpb_buf ws_nms_state;   // per-image / per-tile sweep flags
    kpb_buf ws_nms_map;     // [batch][H*W] working map
    kpb_buf ws_nms_list;    // [batch][2][cap] undecided pixels handed from sweep 0 to the sparse tail
    kpb_buf ws_cand;        // [batch][H*W] uint64 candidate list (key<<32 | ~idx)
    kpb_buf ws_match;       // per-tile row/column minima
    kpb_buf ws_misc;
    kpb_buf ws_sel;         // [batch][chunks] candidate counts of the two-phase selection (small batches)
    int* host_flags = nullptr;  // pinned, for status read-back
    // kpb_detect: select_topk leaves (last sweep that changed, negative flag, keypoint count) of every image HERE -- pinned host
    // memory the kernel writes directly -- so that completing a detection is one stream synchronisation, not two copies back
    int* host_det = nullptr;
    int host_det_cap = 0;
    hipEvent_t wait_ev = nullptr;   // kpb_wait_stream's completion marker
    int* host_match = nullptr;  // likewise: match_finalize leaves every pair's match count here (kpb_match_counts)
    int host_match_cap = 0, host_match_n = 0;
    // state of the last kpb_detect(sync=0), owned by detect.hip
    void* det_state = nullptr;
    void (*det_state_free)(void*) = nullptr;
    int det_pending = 0;
    // the pinned count mirrors hold the counts of a COMPLETED call: set only when kpb_detect(_check) / kpb_match has enqueued (and, for a
    // detection, confirmed) everything, cleared at entry to the next call and on every error path; kpb_*_counts refuse otherwise
    int det_counts_valid = 0, match_counts_valid = 0;
    // hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to the (function, device) pair: remembered per CONTEXT (= per device),
    // not in a process-wide static that a second device would find already set
    unsigned lds_attr_done = 0;
    size_t covis_store_bytes = (size_t)4 << 30;     // KPB_OPT_COVIS_STORE_BYTES (kpb_ctx_set_option)
};
enum { KPB_ATTR_NMS = 1u, KPB_ATTR_HOMOGRAPHY = 2u, KPB_ATTR_ESSENTIAL = 4u, KPB_ATTR_FUNDAMENTAL = 8u };

extern char g_kpb_err[512];

inline int kpb_fail(kpb_ctx* ctx, int code, const char* fmt, ...)
{
    char* dst = ctx ? ctx->err : g_kpb_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define KPB_HIP(ctx, call)                                                                             \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return kpb_fail(ctx, KPB_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                            __FILE__, __LINE__);                                                       \
    } while (0)

// Waits for everything enqueued on the context's stream.  small = the caller waits for a handful of images / one pair (the drop-in path):
// a completion event is POLLED -- hipStreamSynchronize parks the thread and is woken tens of microseconds late, which is 5 % of a
// single pair (0.73 -> 0.70 ms).  Batches keep hipStreamSynchronize: measured on one box (scripts/ab_lib.sh, three interleaved runs
// each, profiles/r04_ab_knobs.txt), polling at the end of every 256-pair step held the chip at 18.00 ms per step where the parked
// wait gives 17.7 -- with no gap at all between steps the chip sits at its sustained clock (the head 10.17 ms instead of 9.86).
inline hipError_t kpb_wait_stream(kpb_ctx* ctx, bool small)
{
    if (!small) return hipStreamSynchronize(ctx->stream);
    if (!ctx->wait_ev) {
        const hipError_t e = hipEventCreateWithFlags(&ctx->wait_ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipEventRecord(ctx->wait_ev, ctx->stream);
    if (e != hipSuccess) return e;
    // a bounded spin (a single pair is ~0.6 ms of kernels), then the parked wait: a kernel that hangs must not burn a host core for ever
    // (ADVICE r04).  A context -- its stream, this event, its workspaces -- is used from ONE thread at a time (include/kpb.h).
    const auto t0 = std::chrono::steady_clock::now();
    while ((e = hipEventQuery(ctx->wait_ev)) == hipErrorNotReady)
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) return hipStreamSynchronize(ctx->stream);
    return e;
}

inline int kpb_reserve(kpb_ctx* ctx, kpb_buf& b, size_t bytes)
{
    if (bytes <= b.cap) return KPB_OK;
    if (b.p) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        b.p = nullptr;
        return kpb_fail(ctx, KPB_E_NOMEM, "workspace allocation of %zu bytes failed", want);
    }
    b.cap = want;
    // KPB_LOG_ALLOC=1: where each workspace landed (scripts/head_modes.py reads these lines: placement against the caller's buffers)
    static const bool log_alloc = getenv("KPB_LOG_ALLOC") && *getenv("KPB_LOG_ALLOC") == '1';
    if (log_alloc) fprintf(stderr, "kpb_alloc %p %zu\n", b.p, want);
    return KPB_OK;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// integer tuning knob from the environment (experiments only; the defaults are the measured choices)
static inline int kpb_env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Brackets one kernel launch with two events on the context's stream when profiling is enabled.
struct ProfScope {
    kpb_ctx* ctx;
    hipEvent_t e1 = nullptr;
    ProfScope(kpb_ctx* c, const char* name) : ctx(c)
    {
        if (!c->prof) return;
        hipEvent_t ev[2];
        for (int i = 0; i < 2; ++i) {
            if (!c->prof_pool.empty()) { ev[i] = c->prof_pool.back(); c->prof_pool.pop_back(); }
            else if (hipEventCreate(&ev[i]) != hipSuccess) return;
        }
        c->prof_recs.push_back({c->prof_names.insert(name).first->c_str(), ev[0], ev[1]});
        e1 = ev[1];
        (void)hipEventRecord(ev[0], c->stream);
    }
    ~ProfScope() { if (e1) (void)hipEventRecord(e1, ctx->stream); }
};

#define KPB_LAUNCH(ctx_, name_, ...)            \
    do {                                        \
        ProfScope ps_(ctx_, name_);             \
        hipLaunchKernelGGL(__VA_ARGS__);        \
    } while (0)
