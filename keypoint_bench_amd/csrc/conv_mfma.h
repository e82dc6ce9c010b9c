// conv_mfma.h -- the generic fp32 NHWC convolution / linear layer on v_mfma_f32_32x32x2_f32 shared by the extractor
// nets (convnet.hip) and the LightGlue matcher (lightglue.hip, where a Linear over tokens is a 1x1 convolution).
#pragma once
#include <type_traits>
#include "net.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float relu(float v) { return fmaxf(v, 0.0f); }

struct ConvM {
    const float* in;    // [B][Hi][Wi][CIN]
    float* out;         // [B][Ho][Wo][COUT]   (Ho, Wo) = (H, W) or (H/2, W/2) with POOL_OUT
    const float* wp;    // packed: [ntile][tap][chunk][h][32][KC]
    const float* bias;  // [COUTP] (zero padded)
    const float* xf;    // XF: [B][CIN][4] = (scale, shift, prelu slope, -) applied to the input before zero padding
    const int* active;  // optional [B]: batch items with 0 are skipped (LightGlue pairs that stopped early)
    const float* res;   // optional [B][H][W][COUT] added before the ReLU (ALIKE ResBlock identity branch, ALike.py:76-79)
    int Hi, Wi, H, W, CIN, COUT, NCH, relu, nblk;
    int istride, ostride, ooff;   // floats between consecutive input / output pixels, channel offset of the output
    float unscale = 1.0f;         // conv_mfma_h / gemm_h: 1 / (the layer's power-of-two weight scale)
    int relu_nt = 0;              // conv_mfma_h with relu == 2: only the first relu_nt 32-wide output tiles get the ReLU
    int rstride = 0;              // floats between consecutive pixels of `res` (0: COUT)
    const float* xw = nullptr;    // conv_mfma_h<XC>: weights of ONE extra output channel (index xco) taken on the VALU: [tap][CIN / 2] x (hi pair, lo pair) of
                                  // halves (pack_xc_pairs), scaled by 1 / xun
    float xun = 1.0f;
    // conv_mfma_h<.., PRE, .., GEN> (r05): the input is GENERATED while the tile is staged -- `in` is a one-channel image and the slab's 32 channels are
    // relu(3 x 3 convolution + bias) of it (SuperPoint conv1a: gen_w [9][64] tap-major, gen_b [64]), scaled and split as the pre-split hand-off would be
    const float* gen_w = nullptr; const float* gen_b = nullptr;
    float xb = 0.0f;              // its bias
    int xco = 0;
    // gemm_h only (the LightGlue linears): rows >= rowcnt[b] are staged as zeros and never written -- what a padded row holds cannot
    // reach a valid row's operand scale; aux0 / aux1 / out1 / out2 belong to the epilogue forms GE_* below
    const int* rowcnt = nullptr;
    const float* aux0 = nullptr;
    const float* aux1 = nullptr;
    float* out1 = nullptr;
    float* out2 = nullptr;
    // conv_mfma_h<.., PRE, .., GEN>: the tile holds the generated activations ALREADY SPLIT -- per pixel and 32-channel slab 128 bytes = [4 x 16 B of hi
    // halves | 4 x 16 B of lo halves], taken at the per-image power-of-two scale of cm_exp_of(amax_in l1 + bmax), a rigorous bound of the generated
    // layer's output (amax_in: the maximum of ITS input, measured; l1 / bmax: its largest row L1 norm / |bias|): no maximum pass over the slab, one
    // barrier per slab.  (r05 also had the producer as its own kernel writing this format to HBM and the consumer landing it by LDS-DMA: 2.10 -> 2.07 ms,
    // profiles/r05_presplit_conv1b_ab.txt -- superseded by GEN and removed in r06, together with KPB_PRESPLIT.)
    const unsigned* pre_amax = nullptr; float pre_l1 = 0.0f, pre_bmax = 0.0f;
    // conv_mfma_h<.., UP> (r06, DISK's decoder): the first up_c input channels are NOT in `in` -- they are the 2 x bilinear upsampling (align_corners = False,
    // disk.py:126-127) of up_src [B][Hi / 2][Wi / 2][up_c], evaluated while the tile is staged; `in` holds the remaining CIN - up_c channels (istride floats
    // per pixel).  The concatenated map [up(bottom) | horizontal] is never written.
    const float* up_src = nullptr; int up_c = 0;
    int unfold_w = 0;             // gemm_h<.., UNFOLD>: `in` is a single-channel [B][8 H][unfold_w] image and row r, channel c stand for pixel
                                  // (8 (r / W) + c / 8, 8 (r % W) + c % 8): XFeat's _unfold2d(x, 8) read in place (XFeat.py:96-103, 138)
};

// gemm_h epilogues.  GE_RESIDUAL: out = res + (W x + b), rows of `res` rstride floats apart (lightglue.py:185 / 242, x + ffn(...)).
// GE_ROTARY: the 768 columns of Wqkv arrive permuted (lg_pack_qkv_rows): workgroup nb = 3 head + which (q, k, v) owns the 32 even
// elements of that head in its first tile and the 32 odd ones in its second, so a lane holds one rotary pair; q and k are rotated by
// (aux0, aux1) = (cos, sin) [row][32] and all three go out head-major as float2 (lightglue.py:71-78, 179-183).
// GE_SPLIT2: output tiles from column 256 on go to out1 (two Linears over the same input as one product).
// GE_L2NORM: COUT = 64 = the workgroup's two tiles, so a wave holds whole rows: out = v / max(||v||_2, xb) (F.normalize, XFeat.py:136).
enum { GE_PLAIN = 0, GE_RESIDUAL = 1, GE_ROTARY = 2, GE_SPLIT2 = 3, GE_L2NORM = 4 };

template <int KS, int S, int CC, bool POOL_IN, bool POOL_OUT, bool XF, int NTB = 2>
__global__ __launch_bounds__(256) void conv_mfma(ConvM a)
{
    constexpr int KC = CC / 2, PITCH = CC + 4, T = KS * KS, PAD = KS / 2;
    constexpr int IH = 7 * S + KS, IW = 15 * S + KS, Q = CC / 4;
    constexpr int NLD = (IH * IW * Q + 255) / 256;
    __shared__ __attribute__((aligned(16))) float tile[IH * IW * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, p = lane & 31, h = lane >> 5;
    const int b = blockIdx.z / a.nblk, nb = blockIdx.z - b * a.nblk, nt0 = nb * NTB;
    const int ty0 = blockIdx.y * 8, tx0 = blockIdx.x * 16;
    const int Hc = POOL_IN ? a.Hi / 2 : a.Hi, Wc = POOL_IN ? a.Wi / 2 : a.Wi;   // conv input extent
    const int iy0 = ty0 * S - PAD, ix0 = tx0 * S - PAD;
    if (a.active && !a.active[b]) return;
    const float* in = a.in + (size_t)b * a.Hi * a.Wi * a.istride;
    const int orow = 2 * wv + (p >> 4), ocol = p & 15;
    const size_t ntile_stride = (size_t)T * a.NCH * 2 * 32 * KC;

    f32x16 acc[NTB];
#pragma unroll
    for (int n = 0; n < NTB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    for (int ch = 0; ch < a.NCH; ++ch) {
        __syncthreads();
        {   // stage one CC-channel slab of the input tile; every load of a thread is in flight before the first LDS store
            float4 buf[NLD];
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int idx = tid + k * 256;
                const int pix = idx / Q, q = idx - pix * Q;
                const int y = pix / IW, x = pix - y * IW;
                const int gy = iy0 + y, gx = ix0 + x;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < IH * IW * Q && gy >= 0 && gy < Hc && gx >= 0 && gx < Wc) {
                    if (POOL_IN) {
                        const float* s = in + ((size_t)(2 * gy) * a.Wi + 2 * gx) * a.istride + ch * CC + 4 * q;
                        const float4 v00 = *reinterpret_cast<const float4*>(s), v01 = *reinterpret_cast<const float4*>(s + a.istride);
                        const float4 v10 = *reinterpret_cast<const float4*>(s + (size_t)a.Wi * a.istride);
                        const float4 v11 = *reinterpret_cast<const float4*>(s + (size_t)a.Wi * a.istride + a.istride);
                        v.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x)); v.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
                        v.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z)); v.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
                    } else {
                        v = *reinterpret_cast<const float4*>(in + ((size_t)gy * a.Wi + gx) * a.istride + ch * CC + 4 * q);
                    }
                    if (XF) {   // InstanceNorm (scale, shift) + PReLU of the pre-activation Conv block (disk.py:76-97)
                        const float4* t4 = reinterpret_cast<const float4*>(a.xf + ((size_t)b * a.CIN + ch * CC + 4 * q) * 4);
                        const float4 t0 = t4[0], t1 = t4[1], t2 = t4[2], t3 = t4[3];
                        v.x = fmaf(v.x, t0.x, t0.y); v.x = v.x >= 0.0f ? v.x : v.x * t0.z;
                        v.y = fmaf(v.y, t1.x, t1.y); v.y = v.y >= 0.0f ? v.y : v.y * t1.z;
                        v.z = fmaf(v.z, t2.x, t2.y); v.z = v.z >= 0.0f ? v.z : v.z * t2.z;
                        v.w = fmaf(v.w, t3.x, t3.y); v.w = v.w >= 0.0f ? v.w : v.w * t3.z;
                    }
                }
                buf[k] = v;
            }
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int idx = tid + k * 256;
                const int pix = idx / Q, q = idx - pix * Q;
                if (idx < IH * IW * Q) *reinterpret_cast<float4*>(&tile[pix * PITCH + 4 * q]) = buf[k];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < T; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
            const float* ap = &tile[((orow * S + ky) * IW + ocol * S + kx) * PITCH + h * KC];
            const float* bp = a.wp + ((((size_t)nt0 * T + tap) * a.NCH + ch) * 2 + h) * 32 * KC + p * KC;
            float A[KC], Bw[NTB][KC];
#pragma unroll
            for (int q = 0; q < KC / 4; ++q) {
                const float4 va = *reinterpret_cast<const float4*>(ap + 4 * q);
                A[4 * q] = va.x; A[4 * q + 1] = va.y; A[4 * q + 2] = va.z; A[4 * q + 3] = va.w;
#pragma unroll
                for (int n = 0; n < NTB; ++n) {
                    const float4 v = *reinterpret_cast<const float4*>(bp + n * ntile_stride + 4 * q);
                    Bw[n][4 * q] = v.x; Bw[n][4 * q + 1] = v.y; Bw[n][4 * q + 2] = v.z; Bw[n][4 * q + 3] = v.w;
                }
            }
#pragma unroll
            for (int s = 0; s < KC; ++s)
#pragma unroll
                for (int n = 0; n < NTB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], Bw[n][s], acc[n], 0, 0, 0);
        }
    }

    // epilogue: D[row = pixel i][col = channel]; lane holds channel p of each 32-wide tile, pixels (r&3) + 8*(r>>2) + 4h
#pragma unroll
    for (int n = 0; n < NTB; ++n) {
        const int co = (nt0 + n) * 32 + p;
        const float bias = a.bias[co];
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = acc[n][r] + bias;
            if (a.relu) v[r] = relu(v[r]);
        }
        if (co >= a.COUT) continue;
        if (!POOL_OUT) {
            float* out = a.out + (size_t)b * a.H * a.W * a.ostride + a.ooff;
            const float* res = a.res ? a.res + (size_t)b * a.H * a.W * a.COUT : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int gy = ty0 + 2 * wv + (i >> 4), gx = tx0 + (i & 15);
                if (gy < a.H && gx < a.W) {
                    float o = v[r];
                    if (res) {      // residual path: v holds conv + bias (a.relu is 0), the ReLU comes after the sum
                        o += res[((size_t)gy * a.W + gx) * a.COUT + co];
                        o = relu(o);
                    }
                    out[((size_t)gy * a.W + gx) * a.ostride + co] = o;
                }
            }
        } else {
            const int Ho = a.H / 2, Wo = a.W / 2;
            float* out = a.out + (size_t)b * Ho * Wo * a.ostride + a.ooff;
            const int gy = ty0 / 2 + wv;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int c = 2 * h + (cc & 1) + 4 * (cc >> 1);       // pooled column inside the tile
                const int rA = 2 * (cc & 1) + 4 * (cc >> 1);          // = 2*(c&1) + 4*(c>>2)
                const int gx = tx0 / 2 + c;
                if (gy < Ho && gx < Wo)
                    out[((size_t)gy * Wo + gx) * a.ostride + co] = fmaxf(fmaxf(v[rA], v[rA + 1]), fmaxf(v[rA + 8], v[rA + 9]));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- split-f16 form
// The same convolution with every fp32 operand written as hi + lo, hi = f16(x) and lo = f16(x - hi), both to nearest (cm_split2)
// (x - hi is exact in fp32), and each product taken as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on v_mfma_f32_32x32x16_f16
// with fp32 accumulation: three f16 MFMAs cover K = 16 in 96 cycles where the fp32 instruction needs 512, and the
// dropped a_lo*b_lo term is <= 2^-22 of the product.  Both operands are scaled by powers of two first so that the lo
// halves stay out of the f16 subnormals and the hi halves below saturation: the weights per layer to max|w| in [2^12, 2^13)
// at pack time, the activations per workgroup and slab by the power of two that fits the largest magnitude of the slab being
// staged (cm_scale_of; the accumulator is rescaled -- exactly -- when a later slab needs a smaller scale), and the
// accumulator is scaled back in the epilogue (ConvM::unscale = 1 / weight scale, times the activation unscale).
//
// LDS: one image [IH][IW] of pixels, each [hi: CC halves | lo: CC halves | 16 bytes pad] = 4*CC + 16 bytes = an odd number
// of 16-byte slots, rows padded to a multiple of 256 bytes.  A ds_read_b128 is served in 16-lane groups (lanes {0-3, 12-15,
// 20-27} and so on); with lane = (pixel row p >> 4, column p & 15) those are 8 pixels of one row and 8 of the next, and with
// an odd pixel pitch and a row pitch of 0 mod 256 bytes their sixteen 16-byte slots are distinct (stride 1).  The first
// r02 layout (two planes, pitch 2*CC + 16, rows unpadded) measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 50 %.
// A wave owns MT 32-pixel M tiles (two output rows each) and reuses every weight fragment it loads (1 KB per wave-load,
// L2-resident) MT times.
typedef _Float16 cm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cm_h2 __attribute__((ext_vector_type(2)));

// fp32 x 2 -> (hi, lo) half-precision pairs: hi = f16(x), lo = f16(x - hi), BOTH rounded to nearest (v_cvt_pk_f16_f32, new on
// gfx950; r02-r03 used v_cvt_pkrtz).  Toward zero leaves |x - hi| < 2^-10 |x| and a lo half that is itself truncated: x is
// represented to 2^-20 |x| in the worst case, 2^-22 on average -- 21-22 significant bits against fp32's 24.  To nearest halves both
// steps: |x - hi| <= 2^-11 |x|, |x - hi - lo| <= 2^-22 |x| worst case, the dropped lo.lo term <= 2^-22 of the product.  Measured on
// 256 full-size pairs (profiles/r04_parity_sweep_256.json): max |score - fp32 CPU chain| 8.0e-6 -> see there; strict fp32 4.6e-6.
typedef float cm_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cm_cvt_pk_rtn(float x, float y)
{
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const cm_f2 v = {x, y};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2_t));
}
// The residuals x - (float)hi are plain C: v_cvt_f32_f16 + a (packed) subtraction, exact (x - hi fits fp32).
// NO VECTOR INSTRUCTION IN INLINE ASSEMBLY (scripts/isa_lint.py, rule E3).  r03-r04 computed the residuals with v_fma_mix_f32 from an asm
// statement (one issue slot less per element) and r04 took it for a trigger of the run-to-run glitches of the dense ALIKE head; r05 found
// the real cause -- gfx950 miscomputes the low lane of packed-fp32 instructions with op_sel [0,1,.] beside f16 MFMAs (DESIGN.md section 3,
// keypoint_bench_amd/isa_fixup.py) -- and this statement only a bystander that changed what the vectoriser folded.  The rule stays for
// its own reason: the compiler neither pads the wait states nor counts the memory operations of an asm string.
__device__ __forceinline__ void cm_split2(float x, float y, unsigned& hi, unsigned& lo)
{
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    hi = cm_cvt_pk_rtn(x, y);
    const h2_t hh = __builtin_bit_cast(h2_t, hi);
    lo = cm_cvt_pk_rtn(x - (float)hh[0], y - (float)hh[1]);
}

__device__ __forceinline__ void cm_split4(const float4 v, uint2& hi, uint2& lo)
{
    cm_split2(v.x, v.y, hi.x, lo.x);
    cm_split2(v.z, v.w, hi.y, lo.y);
}

// ---- dynamic operand range (r03).  A split operand is exact to ~2^-22 only while its hi half does not saturate (|x| < 65504)
// and its lo half stays a normal f16 (|x| >= 2^-3): a FIXED activation scale therefore fails silently on tensors far from
// the range it was chosen for.  Every split site now measures (or rigorously bounds) the largest magnitude `amax` of what it
// is about to split and scales by the power of two that puts amax in [2^14, 2^15); the scale is divided out of the fp32
// accumulator again, exactly.  Values down to 2^-17 amax keep full precision, smaller ones an absolute error <= 2^-38 amax.
// Non-finite data stays non-finite (the scale is clamped, never inf or zero).
__device__ __forceinline__ int cm_exp_of(float amax)           // clamped exponent field: amax < 2^(e - 126)
{
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    return min(max(e, 24), 230);
}
__device__ __forceinline__ float cm_scale_of(int e) { return __uint_as_float((unsigned)(268 - e) << 23); }     // 2^(141 - e): amax * scale < 2^15
__device__ __forceinline__ float cm_unscale_of(int e) { return __uint_as_float((unsigned)(e - 14) << 23); }    // its reciprocal
__device__ __forceinline__ float cm_pow2(int d) { return __uint_as_float((unsigned)(127 + d) << 23); }         // 2^d, |d| <= 126

__device__ __forceinline__ float cm_amax4(float m, const float4 v)
{
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// maximum of a NON-NEGATIVE value over the wave, in every lane.  Six DPP steps on the vector ALU (quad swaps, the two row mirrors,
// the two row broadcasts; lanes a step does not reach read 0, which a non-negative maximum ignores) and one v_readlane of lane 63:
// the butterfly of __shfl_xor is six ds_bpermute round trips through the LDS crossbar (1.3 k cycles in block 1's prologue).
__device__ __forceinline__ float cm_wave_max(float v)
{
    auto step = [](float x, auto ctrl, auto rows) {
        const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rows)::value, 0xF, true);
        return fmaxf(x, __int_as_float(y));
    };
    v = step(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});     // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});     // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});    // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});    // row_mirror: every lane holds its row's maximum
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});    // row_bcast:15 into rows 1 and 3
    v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});    // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// XC: one more output channel than the tiles hold (DISK's 129 = 4 x 32 + 1) is accumulated on the VALU from the same LDS
// tile, one output pixel per thread (MT = 2: 256 pixels) of the workgroup that owns the first NTB tiles, instead of spending a
// 32-wide MFMA tile on it.
// (Tried, r02: requesting the weight fragments of tap t + 1 before the MFMAs of tap t in a second register set -- 12-15 %
// slower on every layer; the extra 32-64 VGPRs cost a wave per SIMD and the other waves already cover the L2 latency.
// The same for the one-tile layer b3c2 alone, where the second set is 16 registers: 0.58 -> 0.71 ms.  Fetching a tap's fragments
// once per workgroup, a tap ahead, through a double-buffered LDS strip (one barrier per tap, 4-8 registers): +1-3 % -- the
// weight round trip is not what keeps the matrix pipe at 46 %.  Nor is the slab staging: a persistent form that fetches the next
// slab (or the next tile's first slab) into registers before the taps of the current one ran 2 % faster on conv1b at two
// waves per SIMD and 20 % slower on DISK's up_3 (the 44 registers of the slab in flight cost the third wave).)
// (Tried, r04: the next slab's input requested into registers right after this slab's split, its round trip flying under the nine taps,
// with the fragments taken one 16-deep k-block at a time so that the 44 staging registers fit beside the accumulators at three waves per
// SIMD (no spill in the loops): every SuperPoint layer 10-14 % SLOWER (conv1b 2.10 -> 2.38 ms), DISK's 5 x 5 layers 0-8 % slower -- the
// half-size fragment sets cost the tap loop more than the hidden load gives back.)
// (Tried, r04: the workgroups sharing a CU start together and take the same time per phase, so their load / split / store phases
// might coincide and idle the matrix pipe together; delaying the first generation's slot k by k x 8-48 k cycles changed no layer of
// SuperPoint by more than 1 % -- the phases are not in lockstep.)
// PF: with POOL_IN the input is max-pooled PF x PF (2 or 4) while it is staged (ALike.py:139-143).
// waves per SIMD the register allocation is held to: what r02's code reached without being told (accumulators 16 MT NTB) --
// left alone, the allocator keeps a second copy of the accumulators in VGPRs for the rare rescale below (+64 registers, a wave
// per SIMD lost on every two-tile layer)
// 16 bytes per lane from global memory to LDS at lds_wave_base + 16 * lane (wave-uniform base), no register in between (the head of
// ALIKE has the story: alike.hip hp_dma16).  M0 belongs to the compiler: saved and restored around the one instruction that reads it.
__device__ __forceinline__ void cm_dma16(const void* gsrc, const void* lds_wave_base)
{
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(lds_wave_base));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(m0v), "v"(gsrc) : "memory");
}

constexpr int conv_mfma_h_waves(int KS, int CC, bool POOL_IN, int NTB, int MT)
{
    return MT * NTB >= 4 ? ((KS == 5 && CC == 32) || POOL_IN ? 2 : 3) : (MT * NTB == 2 ? (KS == 3 && CC == 16 ? 3 : 4) : 5);
}

// WPRE (r04): the latency form for grids that cannot fill the chip (one image on the drop-in path: 2-16 workgroups per layer, each a
// chain of dependent round trips -- per tap a weight fetch, then LDS reads, then the products).  The weight fragments of ALL taps of a
// slab are requested before the slab's input is even loaded and land while it is staged: one round trip per slab instead of ten.
// Costs T x 4 NTB NKB registers (144 for a 3 x 3 kernel, one n-tile), irrelevant at one wave per SIMD; the arithmetic and its order
// are those of the throughput form.
// WN (r05): the four waves as 4 / WN row groups x WN column groups.  WN = 1: every wave multiplies its MT row tiles by ALL NTB n-tiles of the
// workgroup -- the four waves request the same weight fragments, 4 x the bytes through the CU's vector L1 (64 B/clk), and a knock-out
// put 18 % of SuperPoint's conv1b there (profiles/r05_presplit_conv1b_ab.txt).  WN = 2: a wave takes twice the rows and half the
// n-tiles: the same 64 accumulator registers and products, half the weight bytes, twice the activation reads -- which come from LDS.
// The taps of the 2 x bilinear upsampling torch's F.interpolate(scale_factor = 2, mode = 'bilinear', align_corners = False) reads for output coordinate o of
// a source of n samples: source indices i0 <= i1, weight l of i1 (1 - l of i0).  Shared by upsample2_concat (the materialised form the strict-fp32 path keeps),
// up_chan_sums (its statistics) and conv_mfma_h<UP> (the staging that evaluates it in place): the three compute the same floats.
__device__ __forceinline__ void cm_up2_taps(int o, int n, int& i0, int& i1, float& l)
{
    const float f = fmaxf(((float)o + 0.5f) * 0.5f - 0.5f, 0.0f);
    i0 = (int)f;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l = f - (float)i0;
}
__device__ __forceinline__ float4 cm_up2_mix(const float4 p00, const float4 p01, const float4 p10, const float4 p11, float ly, float lx)
{
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    return make_float4(hy * (hx * p00.x + lx * p01.x) + ly * (hx * p10.x + lx * p11.x), hy * (hx * p00.y + lx * p01.y) + ly * (hx * p10.y + lx * p11.y),
                       hy * (hx * p00.z + lx * p01.z) + ly * (hx * p10.z + lx * p11.z), hy * (hx * p00.w + lx * p01.w) + ly * (hx * p10.w + lx * p11.w));
}

// UP (r06): see ConvM::up_src.  DISK's up_3 read an 80-channel map [up(u2) | f1] that a kernel of its own had written (3.1 GB per 32 images, read once more
// for the InstanceNorm statistics and then here with a 5 x 5 halo by both workgroups of a tile).  Here a slab of upsampled channels is staged from the
// HALF-resolution source: its (IH / 2 + 2)^2 pixels land raw in 9 KB of LDS and every staged float4 is the four-tap mix of that -- the direct form (four global
// taps per staged float4, r05) spilled 95 registers and lost 3 ms.  The staging has the room: a slab's taps are 750 MFMAs per wave.
template <int KS, int S, int CC, bool POOL_IN, bool POOL_OUT, bool XF, int NTB = 2, int MT = 1, bool XC = false, int PF = 2, bool WPRE = false, bool PRE = false, int WN = 1, bool GEN = false,
          bool UP = false>
__global__ __launch_bounds__(256, WPRE ? 1 : conv_mfma_h_waves(KS, CC, POOL_IN, NTB, MT)) void conv_mfma_h(ConvM a)
{
    static_assert(!UP || (S == 1 && !POOL_IN && !PRE && !WPRE && (((8 * MT / WN - 1) * S + KS) % 2 == 0) && ((15 * S + KS) % 2 == 0) && KS / 2 % 2 == 0),
                  "conv_mfma_h<UP>: stride-1 layers whose input tile starts on an even row and column");
    static_assert(WN == 1 || (WN == 2 && MT % 2 == 0 && !WPRE), "conv_mfma_h: waves split the n-tiles two ways at most");
    static_assert(PRE == GEN && (!GEN || KS == 3), "conv_mfma_h: the pre-split tile exists as the generated one (a 3 x 3 one-channel layer computed while staging); r05's DMA-landed form was measured, superseded and removed in r06");
    static_assert(!PRE || (CC == 32 && S == 1 && !POOL_IN && !XF && !XC && !WPRE), "conv_mfma_h: the pre-split input form exists for plain stride-1 32-channel slabs");
    constexpr int KC = CC / 2, NKB = CC / 16, T = KS * KS, PAD = KS / 2, TH = 8 * MT / WN;
    // r06: a tap's weight fragments (L2 round trips: NTB NKB 2 loads of 16 bytes per lane) are requested ONE TAP AHEAD and the order is pinned with a scheduling
    // barrier -- left alone the scheduler puts every request next to its first use, the round trip exposed nine (25) times per slab.  The 3 x 3 forms unroll their
    // tap loop (the two fragment sets are then plain registers: ALIKE's b3c2 0.444 -> 0.372 ms per 512 images, SuperPoint's conv2a .. conv4b -4 .. 5 %, no spills);
    // the 5 x 5 forms unroll the kernel's columns only (ROW_AHEAD: a set per column, the row loop rolled); the generated-input form (conv1b: 46 registers spilled when
    // unrolled, 1.90 -> 2.08 ms; rolled with the next set handed over by register copies 3.5 ms -- the copies' wait also covers the slab prefetch) keeps the scheduler's
    // own order.  (r02 lost 12 % with "weights a tap ahead" written in the source alone: the scheduler undid it.  profiles/r06_weights_a_tap_ahead_ab.txt)
    constexpr int TAPU = (KS == 3 && !WPRE && !PRE) ? 9 : 1;
    constexpr bool ROW_AHEAD = TAPU == 1 && !WPRE && !PRE;
    constexpr int IH = (TH - 1) * S + KS, IW = 15 * S + KS, Q = CC / 4;
    constexpr int PITCH = 4 * CC + 16, LO = 2 * CC;                 // bytes per pixel, offset of its lo halves
    constexpr int ROWP = (IW * PITCH + 255) / 256 * 256;            // bytes per tile row
    constexpr int NLD = (IH * IW * Q + 255) / 256;
    static_assert(IH * ROWP <= 65536, "conv_mfma_h: input tile exceeds the static LDS window");
    __shared__ __attribute__((aligned(256))) unsigned char tile[IH * ROWP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, p = lane & 31, h = lane >> 5;
    const int wm = wv / WN;                  // this wave's row group; its n-tiles start at nt0
    const int b = blockIdx.z / a.nblk, nb = blockIdx.z - b * a.nblk, nt0 = (nb * WN + wv % WN) * NTB;
    const int ty0 = blockIdx.y * TH, tx0 = blockIdx.x * 16;
    const int Hc = POOL_IN ? a.Hi / PF : a.Hi, Wc = POOL_IN ? a.Wi / PF : a.Wi;
    const int iy0 = ty0 * S - PAD, ix0 = tx0 * S - PAD;
    if (a.active && !a.active[b]) return;
    const float* in = a.in + (size_t)b * a.Hi * a.Wi * a.istride;
    const int ocol = p & 15;
    const uint4* wq = reinterpret_cast<const uint4*>(a.wp);      // [ntile][tap][chunk][kb][hi/lo][h][32] x 8 halves
    const size_t ntile_stride = (size_t)T * a.NCH * NKB * 4 * 32;

    static_assert(!XC || (TH == 16 && S == 1 && !POOL_OUT), "conv_mfma_h: the extra channel maps one pixel of a 16x16 tile to each thread");
    __shared__ __attribute__((aligned(16))) float s_amax[4];       // the four waves' largest staged magnitude of the slab in flight
    float xacc = 0.0f;
    int e_cur = 24;              // exponent the accumulators' activation scale belongs to (workgroup-uniform)
    // the lane's biases are requested now: read in the epilogue, each cost an exposed L2 round trip behind the last MFMA (r03 stamps:
    // the epilogue of SuperPoint's conv1b took 11 k of the workgroup's 70 k cycles)
    float biasv[NTB];
#pragma unroll
    for (int n = 0; n < NTB; ++n) biasv[n] = a.bias[(nt0 + n) * 32 + p];
    f32x16 acc[MT][NTB];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    for (int ch = 0; ch < a.NCH; ++ch) {
        cm_h8 Wh[WPRE ? T : 1][NTB][NKB], Wl[WPRE ? T : 1][NTB][NKB];
        if constexpr (WPRE) {
#pragma unroll
            for (int tap = 0; tap < T; ++tap) {
                const uint4* bp = wq + ((((size_t)nt0 * T + tap) * a.NCH + ch) * NKB * 4 + h) * 32 + p;
#pragma unroll
                for (int n = 0; n < NTB; ++n)
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb) {
                        Wh[tap][n][kb] = __builtin_bit_cast(cm_h8, bp[n * ntile_stride + (kb * 4 + 0) * 32]);
                        Wl[tap][n][kb] = __builtin_bit_cast(cm_h8, bp[n * ntile_stride + (kb * 4 + 2) * 32]);
                    }
            }
        }
        if constexpr (PRE) {
            // the tile as DENSE 128-byte pixels t = y IW + x, slot s of pixel t at 16-byte position s ^ ((x >> 1) & 7).  ds_read_b128
            // banks are (a / 4) mod 64 -- a 256-byte row = TWO pixels -- and are arbitrated in the 16-lane groups {0-3, 12-15, 20-27},
            // {4-11, 16-19, 28-31} (+32): in a fragment read each group holds the sixteen output columns once (eight of one tile row, eight
            // of the next), i.e. sixteen consecutive x.  IW is even, so x & 1 picks the half of the bank row and (x >> 1) & 7 is distinct
            // over the eight x of either parity: sixteen lanes, sixteen positions.  (The first form keyed on t & 7 for a 128-byte bank
            // row: every position used twice, SQ_LDS_BANK_CONFLICT 50 % -- profiles/r05_pmc_per_kernel_superpoint.csv.)  A DMA unit = one
            // wave instruction = 8 pixels x 8 slots = 1 KB; lane i lands at position i & 7 of pixel 8 u + (i >> 3) and fetches the slot
            // that belongs there.
            static_assert(IW % 2 == 0, "conv_mfma_h<PRE>: the swizzle key assumes pixel parity = column parity");
            constexpr int NPIX = IH * IW, NU = (NPIX + 7) / 8;
            static_assert(NU * 1024 <= IH * ROWP, "conv_mfma_h<PRE>: the dense tile must fit the padded one's LDS");
            if (ch == 0) e_cur = cm_exp_of(fmaf(__uint_as_float(a.pre_amax[b]), a.pre_l1, a.pre_bmax));
            if constexpr (GEN) {
                // conv1a inside conv1b (r05): the producer's 64 channels are never written -- its 3 x 3 window of the one-channel image sits in a
                // (IH + 2) x (IW + 2) tile behind the slab and a thread computes pixel t's 32 channels of this slab, eight at a time against
                // scalar weights (the same fused chain per channel as conv1a_c64: taps in column order), scales, splits and writes the two
                // 16-byte slots where the DMA would have put them.  2.5 GB written and read back per 32 images, and conv1a's launch, are gone.
                constexpr int GW = IW + 2, GH = IH + 2;
                static_assert(NU * 1024 + GH * GW * 4 <= IH * ROWP, "conv_mfma_h<GEN>: the image tile lives behind the dense slab");
                float* gt = reinterpret_cast<float*>(tile + NU * 1024);
                if (ch == 0) {
                    for (int i = tid; i < GH * GW; i += 256) {
                        const int yy = i / GW, gy = iy0 - 1 + yy, gx = ix0 - 1 + (i - yy * GW);
                        gt[i] = (gy >= 0 && gy < Hc && gx >= 0 && gx < Wc) ? in[(size_t)gy * a.Wi + gx] : 0.0f;
                    }
                }
                __syncthreads();                // the previous slab's taps are done with the tile (ch == 0: the image tile is complete)
                const float sc = cm_scale_of(e_cur);
#pragma unroll 1
                for (int t0 = 0; t0 < NPIX; t0 += 256) {
                    const int t = t0 + tid, tc = min(t, NPIX - 1);
                    const int y = tc / IW, x = tc - y * IW;
                    const int gy = iy0 + y, gx = ix0 + x;
                    const bool inimg = gy >= 0 && gy < Hc && gx >= 0 && gx < Wc;      // the consumer pads ITS input with zeros
                    float g[9];
#pragma unroll
                    for (int k = 0; k < 9; ++k) g[k] = gt[(y + k / 3) * GW + x + k % 3];
                    const int key = (x >> 1) & 7;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float* w = a.gen_w + ch * 32 + 8 * s;
                        float acc[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] = a.gen_b[ch * 32 + 8 * s + e];
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                acc[e] = fmaf(g[kx], w[kx * 64 + e], acc[e]);
                                acc[e] = fmaf(g[3 + kx], w[(3 + kx) * 64 + e], acc[e]);
                                acc[e] = fmaf(g[6 + kx], w[(6 + kx) * 64 + e], acc[e]);
                            }
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] = inimg ? relu(acc[e]) * sc : 0.0f;
                        uint2 h0, l0, h1, l1;
                        cm_split4(make_float4(acc[0], acc[1], acc[2], acc[3]), h0, l0);
                        cm_split4(make_float4(acc[4], acc[5], acc[6], acc[7]), h1, l1);
                        if (t < NPIX) {
                            *reinterpret_cast<uint4*>(tile + t * 128 + 16 * (s ^ key)) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                            *reinterpret_cast<uint4*>(tile + t * 128 + 16 * ((4 + s) ^ key)) = make_uint4(l0.x, l0.y, l1.x, l1.y);
                        }
                    }
                }
            }
        } else if constexpr (UP) {
        // UP: ONE pass, no maximum taken over the staged tile -- the slab's scale comes from a rigorous bound of its transformed values over the whole image
        // (ConvM::xf, fourth float per channel: make_xf from the channel's range, which bounds the upsampled values too -- they are convex combinations), so
        // nothing is held across a barrier (the measured-maximum form needs the NLD staged float4s twice: held, 78 registers spilled at three waves per SIMD;
        // evaluated twice, up_3 15.5 -> 17.0 ms).  A per-image bound instead of a per-tile maximum moves the f16 window up by the ratio of the two -- a few
        // binary places out of the 2^-24 .. 2^15 the split's low halves have.
        static_assert(XF && !POOL_IN, "conv_mfma_h<UP>: DISK's decoder layers (InstanceNorm + PReLU while staging)");
        constexpr int RH = UP ? IH / 2 + 2 : 1, RWD = UP ? IW / 2 + 2 : 1;      // the half-resolution source tile of an upsampled slab
        __shared__ __attribute__((aligned(16))) float4 rawbuf[RH * RWD * Q];
        const int ry0 = UP ? iy0 / 2 - 1 : 0, rx0 = UP ? ix0 / 2 - 1 : 0;       // iy0, ix0 even (static_assert): source row of tile row 0 is iy0 / 2 - 1
        const bool upslab = UP && ch * CC < a.up_c;
        if constexpr (UP) {
            if (upslab) {       // workgroup-uniform.  The previous slab's reads of rawbuf lie two barriers back.
                constexpr int NR = RH * RWD * Q, PERR = (NR + 255) / 256;
                const int Hb = a.Hi / 2, Wb = a.Wi / 2;
                const float* src = a.up_src + (size_t)b * Hb * Wb * a.up_c + ch * CC;
                float4 rv[PERR];
#pragma unroll
                for (int k = 0; k < PERR; ++k) {
                    const int i = min(tid + k * 256, NR - 1);
                    const int rp = i / Q, q = i - rp * Q, r = rp / RWD, c = rp - r * RWD;
                    const int sy = min(max(ry0 + r, 0), Hb - 1), sx = min(max(rx0 + c, 0), Wb - 1);
                    rv[k] = *reinterpret_cast<const float4*>(src + ((size_t)sy * Wb + sx) * a.up_c + 4 * q);
                }
#pragma unroll
                for (int k = 0; k < PERR; ++k)
                    if (tid + k * 256 < NR) rawbuf[tid + k * 256] = rv[k];
            }
        }
        __syncthreads();        // the raw tile is in LDS, and the previous slab's taps are done with the tile
        auto staged = [&](int idx, bool& live) -> float4 {
            const int pix = idx / Q, q = idx - pix * Q;
            const int y = pix / IW, x = pix - y * IW;
            const int gy = iy0 + y, gx = ix0 + x;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            live = idx < IH * IW * Q;
            if (live && gy >= 0 && gy < Hc && gx >= 0 && gx < Wc) {
                if (upslab) {
                    int y0, y1, x0, x1;
                    float ly, lx;
                    cm_up2_taps(gy, a.Hi / 2, y0, y1, ly);
                    cm_up2_taps(gx, a.Wi / 2, x0, x1, lx);
                    const float4* r0 = rawbuf + ((y0 - ry0) * RWD - rx0) * Q + q;
                    const float4* r1 = rawbuf + ((y1 - ry0) * RWD - rx0) * Q + q;
                    v = cm_up2_mix(r0[x0 * Q], r0[x1 * Q], r1[x0 * Q], r1[x1 * Q], ly, lx);
                } else
                    v = *reinterpret_cast<const float4*>(in + ((size_t)gy * a.Wi + gx) * a.istride + (ch * CC - a.up_c) + 4 * q);
                const float4* t4 = reinterpret_cast<const float4*>(a.xf + ((size_t)b * a.CIN + ch * CC + 4 * q) * 4);
                const float4 t0 = t4[0], t1 = t4[1], t2 = t4[2], t3 = t4[3];
                v.x = fmaf(v.x, t0.x, t0.y); v.x = v.x >= 0.0f ? v.x : v.x * t0.z;
                v.y = fmaf(v.y, t1.x, t1.y); v.y = v.y >= 0.0f ? v.y : v.y * t1.z;
                v.z = fmaf(v.z, t2.x, t2.y); v.z = v.z >= 0.0f ? v.z : v.z * t2.z;
                v.w = fmaf(v.w, t3.x, t3.y); v.w = v.w >= 0.0f ? v.w : v.w * t3.z;
            }
            return v;
        };
        {
            float bnd = 0.0f;       // workgroup-uniform: sixteen scalar loads
#pragma unroll
            for (int c = 0; c < CC; ++c) bnd = fmaxf(bnd, a.xf[((size_t)b * a.CIN + ch * CC + c) * 4 + 3]);
            const int e_new = cm_exp_of(bnd);
            if (ch == 0) e_cur = e_new;
            else if (e_new > e_cur) {           // this slab needs a smaller scale: bring what has been accumulated down to it (exact)
                const float f = cm_pow2(e_cur - e_new);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTB; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] *= f;
                e_cur = e_new;
            }
            const float sc = cm_scale_of(e_cur);
#pragma unroll 2
            for (int k = 0; k < NLD; ++k) {
                const int idx = tid + k * 256;
                bool live;
                const float4 v = staged(idx, live);
                if (live) {
                    const int pix = idx / Q, q = idx - pix * Q;
                    const int y = pix / IW, x = pix - y * IW;
                    uint2 hi, lo;
                    cm_split4(make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc), hi, lo);
                    *reinterpret_cast<uint2*>(&tile[y * ROWP + x * PITCH + 8 * q]) = hi;
                    *reinterpret_cast<uint2*>(&tile[y * ROWP + x * PITCH + LO + 8 * q]) = lo;
                }
            }
        }
        } else {
        float4 buf[NLD];
        float amax = 0.0f;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + k * 256;
            const int pix = idx / Q, q = idx - pix * Q;
            const int y = pix / IW, x = pix - y * IW;
            const int gy = iy0 + y, gx = ix0 + x;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < IH * IW * Q && gy >= 0 && gy < Hc && gx >= 0 && gx < Wc) {
                if (POOL_IN) {
                    const float* s = in + ((size_t)(PF * gy) * a.Wi + PF * gx) * a.istride + ch * CC + 4 * q;
                    v = *reinterpret_cast<const float4*>(s);
#pragma unroll
                    for (int py = 0; py < PF; ++py)
#pragma unroll
                        for (int px = 0; px < PF; ++px) {
                            if (py == 0 && px == 0) continue;
                            const float4 w4 = *reinterpret_cast<const float4*>(s + ((size_t)py * a.Wi + px) * a.istride);
                            v.x = fmaxf(v.x, w4.x); v.y = fmaxf(v.y, w4.y); v.z = fmaxf(v.z, w4.z); v.w = fmaxf(v.w, w4.w);
                        }
                } else {
                    v = *reinterpret_cast<const float4*>(in + ((size_t)gy * a.Wi + gx) * a.istride + ch * CC + 4 * q);
                }
                if (XF) {
                    const float4* t4 = reinterpret_cast<const float4*>(a.xf + ((size_t)b * a.CIN + ch * CC + 4 * q) * 4);
                    const float4 t0 = t4[0], t1 = t4[1], t2 = t4[2], t3 = t4[3];
                    v.x = fmaf(v.x, t0.x, t0.y); v.x = v.x >= 0.0f ? v.x : v.x * t0.z;
                    v.y = fmaf(v.y, t1.x, t1.y); v.y = v.y >= 0.0f ? v.y : v.y * t1.z;
                    v.z = fmaf(v.z, t2.x, t2.y); v.z = v.z >= 0.0f ? v.z : v.z * t2.z;
                    v.w = fmaf(v.w, t3.x, t3.y); v.w = v.w >= 0.0f ? v.w : v.w * t3.z;
                }
            }
            buf[k] = v;
            amax = cm_amax4(amax, v);
        }
        amax = cm_wave_max(amax);
        if (lane == 0) s_amax[wv] = amax;
        // ONE barrier serves two purposes: the previous slab's taps are done with the tile, and the four wave maxima are
        // visible.  (The global loads above were issued before it: they touch registers only.)
        __syncthreads();
        {
            const float4 am = *reinterpret_cast<const float4*>(s_amax);
            const int e_new = cm_exp_of(fmaxf(fmaxf(am.x, am.y), fmaxf(am.z, am.w)));
            if (ch == 0) e_cur = e_new;
            else if (e_new > e_cur) {           // this slab needs a smaller scale: bring what has been accumulated down to it (exact)
                const float f = cm_pow2(e_cur - e_new);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTB; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] *= f;
                e_cur = e_new;
            }
            const float sc = cm_scale_of(e_cur);
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int idx = tid + k * 256;
                const int pix = idx / Q, q = idx - pix * Q;
                if (idx < IH * IW * Q) {
                    const int y = pix / IW, x = pix - y * IW;
                    uint2 hi, lo;
                    cm_split4(make_float4(buf[k].x * sc, buf[k].y * sc, buf[k].z * sc, buf[k].w * sc), hi, lo);
                    *reinterpret_cast<uint2*>(&tile[y * ROWP + x * PITCH + 8 * q]) = hi;
                    *reinterpret_cast<uint2*>(&tile[y * ROWP + x * PITCH + LO + 8 * q]) = lo;
                }
            }
        }
        }       // !PRE
        __syncthreads();
        float xpart = 0.0f;      // XC: this slab's share of the extra channel, in the slab's activation scale
        static_assert(!WPRE || !XC, "conv_mfma_h: no latency form of the extra-channel layers");
        if constexpr (WPRE) {
#pragma unroll
            for (int tap = 0; tap < T; ++tap) {
                const int ky = tap / KS, kx = tap - ky * KS;
                cm_h8 Ah[MT][NKB], Al[MT][NKB];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int orow = 2 * (wm * MT + m) + (p >> 4);
                    const unsigned char* ap = &tile[(orow * S + ky) * ROWP + (ocol * S + kx) * PITCH + h * KC * 2];
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb) {
                        Ah[m][kb] = __builtin_bit_cast(cm_h8, *reinterpret_cast<const uint4*>(ap + 16 * kb));
                        Al[m][kb] = __builtin_bit_cast(cm_h8, *reinterpret_cast<const uint4*>(ap + LO + 16 * kb));
                    }
                }
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NTB; ++n) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[m][kb], Wh[tap][n][kb], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[m][kb], Wl[tap][n][kb], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[m][kb], Wh[tap][n][kb], acc[m][n], 0, 0, 0);
                        }
            }
        } else {
        // fragment sets (two are live at a time): TAPU == 9: one per tap; ROW_AHEAD: one per kernel column (the row loop is rolled, the column loop unrolled: the
        // indices are constants either way); otherwise one
        constexpr int NSET = TAPU == 9 ? T : (ROW_AHEAD ? KS : 1);
        cm_h8 BhA[NSET][NTB][NKB], BlA[NSET][NTB][NKB];
        auto loadB = [&](int tap, int set) {
            const uint4* bp = wq + ((((size_t)nt0 * T + tap) * a.NCH + ch) * NKB * 4 + h) * 32 + p;
#pragma unroll
            for (int n = 0; n < NTB; ++n)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    BhA[set][n][kb] = __builtin_bit_cast(cm_h8, bp[n * ntile_stride + (kb * 4 + 0) * 32]);
                    BlA[set][n][kb] = __builtin_bit_cast(cm_h8, bp[n * ntile_stride + (kb * 4 + 2) * 32]);
                }
        };
        loadB(0, 0);
#pragma unroll (TAPU == 9 ? KS : 1)
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll (PRE ? 1 : KS)      // (the generated-input form with its columns unrolled: conv1b 1.83 -> 3.3 ms)
        for (int kx = 0; kx < KS; ++kx) {
            const int tap = ky * KS + kx;
            cm_h8 Ah[MT][NKB], Al[MT][NKB];
            constexpr bool AHEAD = TAPU == 9 || ROW_AHEAD;
            if constexpr (AHEAD) {
                if (tap + 1 < T) loadB(tap + 1, TAPU == 9 ? tap + 1 : (kx + 1) % KS);
                __builtin_amdgcn_sched_barrier(0);
            } else if (tap > 0) loadB(tap, 0);          // requested where they are used: the scheduler's own order
            auto& Bh = BhA[TAPU == 9 ? tap : (ROW_AHEAD ? kx : 0)];
            auto& Bl = BlA[TAPU == 9 ? tap : (ROW_AHEAD ? kx : 0)];
#pragma unroll
            for (int m = 0; m < ((MT >= 2 && NTB == 1 && !PRE && !(XC && !UP)) ? 0 : MT); ++m) {
                const int orow = 2 * (wm * MT + m) + (p >> 4);
                if constexpr (PRE) {
                    const int t = (orow + ky) * IW + ocol + kx, key = ((ocol + kx) >> 1) & 7;
                    const unsigned char* ap = &tile[t * 128];
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb) {
                        Ah[m][kb] = __builtin_bit_cast(cm_h8, *reinterpret_cast<const uint4*>(ap + 16 * ((2 * h + kb) ^ key)));
                        Al[m][kb] = __builtin_bit_cast(cm_h8, *reinterpret_cast<const uint4*>(ap + 16 * ((4 + 2 * h + kb) ^ key)));
                    }
                } else {
                const unsigned char* ap = &tile[(orow * S + ky) * ROWP + (ocol * S + kx) * PITCH + h * KC * 2];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    Ah[m][kb] = __builtin_bit_cast(cm_h8, *reinterpret_cast<const uint4*>(ap + 16 * kb));
                    Al[m][kb] = __builtin_bit_cast(cm_h8, *reinterpret_cast<const uint4*>(ap + LO + 16 * kb));
                }
                }
            }
            if constexpr (MT >= 2 && NTB == 1 && !PRE && !(XC && !UP)) {
                // r06: the A pieces of a k-block in a pinned order -- the MT lo pieces and the first hi piece in flight together, the next hi piece requested behind
                // every lo piece's MFMA (whose registers it may take: the allocator decides, the order only makes it possible).  Left alone the scheduler sent the
                // lo pieces through ONE register quad, a ds_read_b128 and an exposed wait each.  Same MFMAs per accumulator in the same order: bit-identical.
                // DISK +1.1 %, SuperPoint +0.8 % (profiles/r06_weights_a_tap_ahead_ab.txt, 6.).  (Not the extra-channel form without the fused upsampling: 4 registers spilled.)
                auto apiece = [&](int m, int kb, bool lo) {
                    const int orow = 2 * (wm * MT + m) + (p >> 4);
                    const unsigned char* ap = &tile[(orow * S + ky) * ROWP + (ocol * S + kx) * PITCH + h * KC * 2];
                    return __builtin_bit_cast(cm_h8, *reinterpret_cast<const uint4*>(ap + (lo ? LO : 0) + 16 * kb));
                };
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    cm_h8 L[MT], H[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) L[m] = apiece(m, kb, true);
                    H[0] = apiece(0, kb, false);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(L[m], Bh[0][kb], acc[m][0], 0, 0, 0);
                        if (m + 1 < MT) { H[m + 1] = apiece(m + 1, kb, false); __builtin_amdgcn_sched_barrier(0); }
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H[m], Bl[0][kb], acc[m][0], 0, 0, 0);
                        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H[m], Bh[0][kb], acc[m][0], 0, 0, 0);
                    }
                }
            } else
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTB; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[m][kb], Bh[n][kb], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[m][kb], Bl[n][kb], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[m][kb], Bh[n][kb], acc[m][n], 0, 0, 0);
                    }
            if (XC && nb == 0) {       // the extra channel at pixel (tid / 16, tid % 16): hi + lo restores the staged activation, weights are wave-uniform
                const unsigned char* xp = &tile[((tid >> 4) + ky) * ROWP + ((tid & 15) + kx) * PITCH];
                // r05: as the matrix pipe takes its products -- x w = x_lo w_hi + x_hi w_lo + x_hi w_hi on halves, fp32 sums -- two channels per
                // v_dot2c_f32_f16 with the weight pair in a scalar register: 1.5 instructions per product (it was two conversions, an add and
                // an fma against fp32 weights: 7 % of DISK's up_3 in a knock-out)
                typedef _Float16 cm_h2 __attribute__((ext_vector_type(2)));
                const uint2* xwq = reinterpret_cast<const uint2*>(a.xw) + (((size_t)tap * a.CIN + ch * CC) >> 1);
#pragma unroll
                for (int q = 0; q < CC / 8; ++q) {
                    const uint4 vh = *reinterpret_cast<const uint4*>(xp + 16 * q), vl = *reinterpret_cast<const uint4*>(xp + LO + 16 * q);
                    const unsigned hh[4] = {vh.x, vh.y, vh.z, vh.w}, ll[4] = {vl.x, vl.y, vl.z, vl.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint2 w = xwq[4 * q + j];
                        xpart = __builtin_amdgcn_fdot2(__builtin_bit_cast(cm_h2, ll[j]), __builtin_bit_cast(cm_h2, w.x), xpart, false);
                        xpart = __builtin_amdgcn_fdot2(__builtin_bit_cast(cm_h2, hh[j]), __builtin_bit_cast(cm_h2, w.y), xpart, false);
                        xpart = __builtin_amdgcn_fdot2(__builtin_bit_cast(cm_h2, hh[j]), __builtin_bit_cast(cm_h2, w.x), xpart, false);
                    }
                }
            }
        }
        }
        if (XC && nb == 0) xacc = fmaf(xpart, cm_unscale_of(e_cur), xacc);
    }
    if (XC && nb == 0) {
        const int gy = ty0 + (tid >> 4), gx = tx0 + (tid & 15);
        if (gy < a.H && gx < a.W) {
            float o = fmaf(xacc, a.xun, a.xb);
            if (a.relu) o = relu(o);
            a.out[(size_t)b * a.H * a.W * a.ostride + a.ooff + ((size_t)gy * a.W + gx) * a.ostride + a.xco] = o;
        }
    }
    const float unscale = a.unscale * cm_unscale_of(e_cur);      // weight scale and activation scale, both powers of two

    // epilogue: as conv_mfma, per M tile; the accumulator carries the layer's weight scale x the activation scale of e_cur
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int trow = 2 * (wm * MT + m);          // first of the tile's two output rows inside the workgroup tile
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
            const int co = (nt0 + n) * 32 + p;
            const float bias = biasv[n];
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = fmaf(acc[m][n][r], unscale, bias);
                if (a.relu == 1 || (a.relu == 2 && nt0 + n < a.relu_nt)) v[r] = relu(v[r]);
            }
            if (co >= a.COUT) continue;
            if (!POOL_OUT) {
                float* out = a.out + (size_t)b * a.H * a.W * a.ostride + a.ooff;
                const int rs = a.rstride ? a.rstride : a.COUT;
                const float* res = a.res ? a.res + (size_t)b * a.H * a.W * rs : nullptr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int gy = ty0 + trow + (i >> 4), gx = tx0 + (i & 15);
                    if (gy < a.H && gx < a.W) {
                        float o = v[r];
                        if (res) {
                            o += res[((size_t)gy * a.W + gx) * rs + co];
                            o = relu(o);
                        }
                        out[((size_t)gy * a.W + gx) * a.ostride + co] = o;
                    }
                }
            } else {
                const int Ho = a.H / 2, Wo = a.W / 2;
                float* out = a.out + (size_t)b * Ho * Wo * a.ostride + a.ooff;
                const int gy = (ty0 + trow) / 2;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int c = 2 * h + (cc & 1) + 4 * (cc >> 1);
                    const int rA = 2 * (cc & 1) + 4 * (cc >> 1);
                    const int gx = tx0 / 2 + c;
                    if (gy < Ho && gx < Wo)
                        out[((size_t)gy * Wo + gx) * a.ostride + co] = fmaxf(fmaxf(v[rA], v[rA + 1]), fmaxf(v[rA + 8], v[rA + 9]));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- 1x1 / Linear
// A 1x1 convolution has no taps to amortise the staging of conv_mfma_h over: per 32-channel slab it pays a global-load
// latency, two barriers and a weight-fragment latency for 24 MFMAs.  gemm_h is the same product as a software pipeline
// over the slabs: the rows (pixels / tokens) of slab c + 1 are already in registers (global loads issued before the MFMAs
// of slab c) and go into the OTHER of two LDS buffers behind them, so there is one barrier per slab and the loads fly
// under the matrix work.  Workgroup = 256 rows x 32 NTB columns; wave = 64 rows (two M tiles); weights in pack_mfma_h's
// KS = 1 order; LDS rows as in conv_mfma_h (hi | lo | pad, 144 bytes: nine 16-byte slots, conflict-free).
template <int NTB, int MT = 2, int EPI = GE_PLAIN, bool UNFOLD = false>
// (at four waves per SIMD the one-tile GE_RESIDUAL / GE_ROTARY forms spill 2 / 11 registers to scratch -- their epilogue operands
// xv sit beside the accumulators in the last slab.  r05 measured the alternative it named, three waves per SIMD for those two (132 / 144
// registers, no scratch): ffn3 + residual 0.915 -> 1.09 ms per 18 launches, three interleaved runs -- the spill is the cheaper evil; loading
// the second k-block's weight fragments behind the first one's products changed nothing, the scheduler hoists them back.
//  r06: measured per form -- the ROTARY form alone at three waves per SIMD (144 registers, its 11 spilled registers gone): lg_Wqkv 0.091 -> 0.077 ms per launch, and it
//  takes them; the RESIDUAL form stays at four.  The weight fragments through LDS, once per workgroup by LDS-DMA a slab ahead with one barrier per slab (53 KB: three
//  workgroups per CU), help only that same form and by the same amount (it is the spills, not the bytes): ffn3 +18 %, to_out / out_proj +14 %, ffn0 +3 %.
//  profiles/r06_gemm_forms_ab.txt)
__global__ __launch_bounds__(256, MT == 1 ? (EPI == GE_ROTARY ? 3 : 4) : 2) void gemm_h(ConvM a)
{
    // r03: a wave stages exactly the 32 MT rows it multiplies, so its slice of the LDS buffers is private to it (a wave's LDS
    // operations execute in order: no barrier anywhere in the kernel) and the activation scale is chosen PER WAVE from the
    // largest magnitude of the slab it is staging (see cm_scale_of).
    constexpr int CC = 32, KC = 16, NKB = 2, Q = CC / 4, PITCH = 4 * CC + 16, LO = 2 * CC, WROWS = 32 * MT, ROWS = 4 * WROWS, WBUF = WROWS * PITCH,
                  NLD = WROWS * Q / 64;
    __shared__ __attribute__((aligned(256))) unsigned char tile[2 * 4 * WBUF];     // [buffer][wave][row][hi | lo | pad]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, p = lane & 31, h = lane >> 5;
    const int b = blockIdx.z / a.nblk, nb = blockIdx.z - b * a.nblk, nt0 = nb * NTB;
    if (a.active && !a.active[b]) return;
    const int nrows_all = a.H * a.W, r0 = blockIdx.x * ROWS + WROWS * wv;      // this wave's first row
    const int nrows = a.rowcnt ? min(nrows_all, a.rowcnt[b]) : nrows_all;
    if (r0 >= nrows) return;                                     // wave-level: the kernel has no barrier
    if ((EPI == GE_RESIDUAL || EPI == GE_ROTARY) && blockIdx.x * ROWS + ROWS > nrows_all) return;     // these forms read whole row tiles ahead (LightGlue pads to 128 rows)
    const float* in = a.in + (size_t)b * nrows_all * a.istride;
    const uint4* wq = reinterpret_cast<const uint4*>(a.wp);      // [ntile][chunk][kb][hi/lo][h][32] x 8 halves
    const size_t ntile_stride = (size_t)a.NCH * NKB * 4 * 32;
    unsigned char* wt = tile + wv * WBUF;
    float biasv[NTB];            // requested now, used in the epilogue (see conv_mfma_h)
#pragma unroll
    for (int n = 0; n < NTB; ++n) biasv[n] = a.bias[(nt0 + n) * 32 + p];

    f32x16 acc[MT][NTB];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    float4 buf[NLD];
    int e_cur = 24;
    auto fetch = [&](int ch) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = lane + k * 64, row = idx / Q, q = idx - row * Q;
            buf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + row < nrows) {
                if constexpr (UNFOLD) {       // channels 32 ch + 4 q .. + 3 of cell (Y, X) = four adjacent pixels of cell row 4 ch + q / 2
                    const int r = r0 + row, Y = r / a.W, X = r - Y * a.W;
                    buf[k] = *reinterpret_cast<const float4*>(a.in + ((size_t)b * 8 * a.H + 8 * Y + 4 * ch + (q >> 1)) * a.unfold_w + 8 * X + 4 * (q & 1));
                    if (a.aux0) {       // (mean, 1 / std) of image b: the image is normalised as it is read (XFeat.py:122-123)
                        const float2 nm = reinterpret_cast<const float2*>(a.aux0)[b];
                        buf[k].x = (buf[k].x - nm.x) * nm.y; buf[k].y = (buf[k].y - nm.x) * nm.y; buf[k].z = (buf[k].z - nm.x) * nm.y; buf[k].w = (buf[k].w - nm.x) * nm.y;
                    }
                } else
                    buf[k] = *reinterpret_cast<const float4*>(in + (size_t)(r0 + row) * a.istride + ch * CC + 4 * q);
            }
        }
    };
    auto stage = [&](int which, bool first) {
        float amax = 0.0f;
#pragma unroll
        for (int k = 0; k < NLD; ++k) amax = cm_amax4(amax, buf[k]);
        const int e_new = cm_exp_of(cm_wave_max(amax));
        if (first) e_cur = e_new;
        else if (e_new > e_cur) {              // this slab needs a smaller scale: bring the accumulators down to it (exact)
            const float f = cm_pow2(e_cur - e_new);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NTB; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] *= f;
            e_cur = e_new;
        }
        const float sc = cm_scale_of(e_cur);
        unsigned char* t = wt + which * 4 * WBUF;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = lane + k * 64, row = idx / Q, q = idx - row * Q;
            uint2 hi, lo;
            cm_split4(make_float4(buf[k].x * sc, buf[k].y * sc, buf[k].z * sc, buf[k].w * sc), hi, lo);
            *reinterpret_cast<uint2*>(&t[row * PITCH + 8 * q]) = hi;
            *reinterpret_cast<uint2*>(&t[row * PITCH + LO + 8 * q]) = lo;
        }
    };
    fetch(0);
    stage(0, true);
    // (Also requesting the next slab's weight fragments a slab ahead costs 94 more VGPRs and the second wave per SIMD: 1.7x slower.)
    // what the fused epilogues read per output element is requested ahead of the LAST slab's products (in the epilogue each load
    // was an exposed round trip: ffn3 + residual 0.83 -> 1.10 ms per 18 launches)
    constexpr int NX = EPI == GE_RESIDUAL ? NTB : (EPI == GE_ROTARY ? 2 : 0);
    float xv[MT][NX ? NX : 1][16];
    auto fetch_epilogue = [&]() {
        if constexpr (EPI == GE_RESIDUAL) {
            const float* res = a.res + (size_t)b * nrows_all * a.rstride;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NTB; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = r0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;        // < nrows_all: see the check at the top
                        xv[m][n][r] = res[(size_t)row * a.rstride + (nt0 + n) * 32 + p];
                    }
        } else if constexpr (EPI == GE_ROTARY) {
            const float* cs = a.aux0 + (size_t)b * nrows_all * 32 + p;
            const float* sn = a.aux1 + (size_t)b * nrows_all * 32 + p;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
                    xv[m][0][r] = cs[(size_t)row * 32];
                    xv[m][1][r] = sn[(size_t)row * 32];
                }
        }
    };
    // the last slab is peeled off the loop: it has nothing to fetch ahead, so the registers of `buf` carry the epilogue's operands
    auto slab = [&](int ch, auto more_t) {
        constexpr bool more = decltype(more_t)::value;
        if constexpr (more) fetch(ch + 1);
        else fetch_epilogue();
        const uint4* bp = wq + (((size_t)nt0 * a.NCH + ch) * NKB * 4 + h) * 32 + p;
        cm_h8 Bh[NTB][NKB], Bl[NTB][NKB], Ah[MT][NKB], Al[MT][NKB];
#pragma unroll
        for (int n = 0; n < NTB; ++n)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                Bh[n][kb] = __builtin_bit_cast(cm_h8, bp[n * ntile_stride + (kb * 4 + 0) * 32]);
                Bl[n][kb] = __builtin_bit_cast(cm_h8, bp[n * ntile_stride + (kb * 4 + 2) * 32]);
            }
        const unsigned char* t = wt + (ch & 1) * 4 * WBUF;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const unsigned char* ap = t + (32 * m + p) * PITCH + h * KC * 2;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                Ah[m][kb] = __builtin_bit_cast(cm_h8, *reinterpret_cast<const uint4*>(ap + 16 * kb));
                Al[m][kb] = __builtin_bit_cast(cm_h8, *reinterpret_cast<const uint4*>(ap + LO + 16 * kb));
            }
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NTB; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[m][kb], Bh[n][kb], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[m][kb], Bl[n][kb], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[m][kb], Bh[n][kb], acc[m][n], 0, 0, 0);
                }
        if constexpr (more) stage((ch + 1) & 1, false);      // the wave's other buffer
    };
    for (int ch = 0; ch + 1 < a.NCH; ++ch) slab(ch, std::true_type{});
    slab(a.NCH - 1, std::false_type{});

    const float unscale = a.unscale * cm_unscale_of(e_cur);
    if constexpr (EPI == GE_ROTARY) {
        static_assert(EPI != GE_ROTARY || NTB == 2, "gemm_h: the rotary epilogue pairs the workgroup's two tiles");
        const int head = nb / 3, which = nb - 3 * head;
        float* dst = (which == 0 ? a.out : which == 1 ? a.out1 : a.out2) + (size_t)b * nrows_all * 256 + head * 64 + 2 * p;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= nrows) continue;
                const float ev = fmaf(acc[m][0][r], unscale, biasv[0]), od = fmaf(acc[m][1][r], unscale, biasv[1]);
                float2 o = make_float2(ev, od);
                if (which < 2) {
                    const float c = xv[m][0][r], s = xv[m][1][r];
                    o.x = __fadd_rn(__fmul_rn(ev, c), __fmul_rn(-od, s));
                    o.y = __fadd_rn(__fmul_rn(od, c), __fmul_rn(ev, s));
                }
                *reinterpret_cast<float2*>(dst + (size_t)row * 256) = o;
            }
        return;
    }
    if constexpr (EPI == GE_L2NORM) {
        static_assert(EPI != GE_L2NORM || NTB == 2, "gemm_h: the norm is taken over the workgroup's two tiles");
        float* o = a.out + (size_t)b * nrows_all * a.ostride + a.ooff + p;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v0 = fmaf(acc[m][0][r], unscale, biasv[0]), v1 = fmaf(acc[m][1][r], unscale, biasv[1]);
                if (a.relu) { v0 = relu(v0); v1 = relu(v1); }
                float ss = fmaf(v0, v0, v1 * v1);                    // the 32 lanes of a half hold the row's 64 columns
                ss += kpb_shfl_xor<16>(ss); ss += kpb_shfl_xor<8>(ss); ss += kpb_shfl_xor<4>(ss); ss += kpb_shfl_xor<2>(ss); ss += kpb_shfl_xor<1>(ss);
                const float nrm = fmaxf(sqrtf(ss), a.xb);
                if (row < nrows) {
                    o[(size_t)row * a.ostride] = __fdiv_rn(v0, nrm);
                    o[(size_t)row * a.ostride + 32] = __fdiv_rn(v1, nrm);
                }
            }
        return;
    }
    float* out = a.out + (size_t)b * nrows_all * a.ostride + a.ooff;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n) {
            int co = (nt0 + n) * 32 + p;
            const float bias = biasv[n];
            if (co >= a.COUT) continue;
            float* o = out;
            if (EPI == GE_SPLIT2 && co >= 256) { o = a.out1 + (size_t)b * nrows_all * a.ostride; co -= 256; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = fmaf(acc[m][n][r], unscale, bias);
                if (a.relu) v = relu(v);
                if (row < nrows) {
                    if constexpr (EPI == GE_RESIDUAL) v = xv[m][n][r] + v;
                    o[(size_t)row * a.ostride + co] = v;
                }
            }
        }
}

// power of two that brings max|w| into [2^12, 2^13); 1 for an all-zero layer
inline float weight_scale_h(const float* w, size_t n)
{
    float m = 0.0f;
    for (size_t i = 0; i < n; ++i) m = std::max(m, std::fabs(w[i]));
    if (!(m > 0.0f) || !std::isfinite(m)) return 1.0f;
    int e;
    std::frexp(m, &e);          // m = f * 2^e, f in [0.5, 1)
    return std::ldexp(1.0f, 13 - e);
}

inline uint16_t f16_bits_rtn(float x)   // fp32 -> f16 to nearest even, as the device's v_cvt_pk_f16_f32 (cm_split2)
{
    const _Float16 h = (_Float16)x;
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}

inline float f16_bits_to_float(uint16_t hb)
{
    const int e = (hb >> 10) & 0x1F;
    const int man = hb & 0x3FF;
    float v = e == 0 ? std::ldexp((float)man, -24) : std::ldexp((float)(man | 0x400), e - 25);
    return (hb & 0x8000) ? -v : v;
}

// OIHW -> conv_mfma_h fragment order [ntile][tap][chunk][kb][hi/lo][h][32] x 8 halves, scaled by `scale`; the result
// has exactly the byte count of pack_mfma's, and is returned in a float vector so that it travels the same way
std::vector<float> pack_mfma_h(const float* w, int COUT, int CIN, int KS, int CC, int NTB, float scale)
{
    const int KC = CC / 2, NKB = CC / 16, T = KS * KS, NCH = CIN / CC, NT = ((COUT + 32 * NTB - 1) / (32 * NTB)) * NTB;
    std::vector<uint16_t> out((size_t)NT * T * NCH * NKB * 4 * 32 * 8, 0);
    for (int nt = 0; nt < NT; ++nt)
        for (int tap = 0; tap < T; ++tap)
            for (int ch = 0; ch < NCH; ++ch)
                for (int kb = 0; kb < NKB; ++kb)
                    for (int h = 0; h < 2; ++h)
                        for (int j = 0; j < 32; ++j)
                            for (int e = 0; e < 8; ++e) {
                                const int o = nt * 32 + j, c = ch * CC + h * KC + 8 * kb + e;
                                if (o >= COUT) continue;
                                const float x = w[((size_t)o * CIN + c) * T + tap] * scale;
                                const uint16_t hi = f16_bits_rtn(x);
                                const uint16_t lo = f16_bits_rtn(x - f16_bits_to_float(hi));
                                const size_t base = (((((size_t)nt * T + tap) * NCH + ch) * NKB + kb) * 4) * 32;
                                out[((base + (0 + h) * 32 + j) * 8) + e] = hi;
                                out[((base + (2 + h) * 32 + j) * 8) + e] = lo;
                            }
    std::vector<float> f(out.size() / 2);
    std::memcpy(f.data(), out.data(), out.size() * 2);
    return f;
}

// One output channel's weights OIHW-row [CIN][KS][KS] -> [tap][CIN / 2] x (hi pair, lo pair) of halves, scaled: what conv_mfma_h<XC> multiplies
// two channels at a time (v_dot2c_f32_f16)
inline std::vector<float> pack_xc_pairs(const float* w1, int CIN, int T, float scale)
{
    std::vector<uint16_t> out((size_t)T * CIN * 2, 0);
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < CIN; ++c) {
            const float x = w1[(size_t)c * T + t] * scale;
            const uint16_t hi = f16_bits_rtn(x), lo = f16_bits_rtn(x - f16_bits_to_float(hi));
            const size_t pair = ((size_t)t * CIN + c) >> 1;
            out[pair * 4 + (c & 1)] = hi;
            out[pair * 4 + 2 + (c & 1)] = lo;
        }
    std::vector<float> f(out.size() / 2);
    std::memcpy(f.data(), out.data(), out.size() * 2);
    return f;
}

// The split-f16 matrix form is the product; KPB_FP32_MATRIX=1 selects the strict-fp32 kernels everywhere (fp32 MFMA / fp32 vector
// ALUs: the forms of r01) -- the companion figure of bench.py and a cross-check the tests run in a child process.  Read once
// per process.
inline bool conv_mfma_use_h16()
{
    static const int v = kpb_env_int("KPB_FP32_MATRIX", 0);
    return v == 0;
}

// OIHW [COUT][CIN][KS][KS] -> conv_mfma fragment order [ntile][tap][chunk][h][32][KC]
std::vector<float> pack_mfma(const float* w, int COUT, int CIN, int KS, int CC, int NTB)
{
    const int KC = CC / 2, T = KS * KS, NCH = CIN / CC, NT = ((COUT + 32 * NTB - 1) / (32 * NTB)) * NTB;
    std::vector<float> out((size_t)NT * T * NCH * 2 * 32 * KC, 0.0f);
    for (int nt = 0; nt < NT; ++nt)
        for (int tap = 0; tap < T; ++tap)
            for (int ch = 0; ch < NCH; ++ch)
                for (int h = 0; h < 2; ++h)
                    for (int j = 0; j < 32; ++j)
                        for (int s = 0; s < KC; ++s) {
                            const int o = nt * 32 + j, c = ch * CC + h * KC + s;
                            if (o < COUT)
                                out[(((((size_t)nt * T + tap) * NCH + ch) * 2 + h) * 32 + j) * KC + s] = w[((size_t)o * CIN + c) * T + tap];
                        }
    return out;
}

std::vector<float> pad_bias(const float* b, int COUT, int mult)
{
    std::vector<float> out(((COUT + mult - 1) / mult) * mult, 0.0f);
    if (b) for (int i = 0; i < COUT; ++i) out[i] = b[i];
    return out;
}


}  // namespace
