// convnet.hip -- N2 SuperPoint (models/SuperPoint.py:30-71) on gfx950: a generic fp32 NHWC convolution on
// v_mfma_f32_32x32x2_f32 for the wide layers, a VALU convolution for the thin ones, and the small
// head kernels (RGB sum, softmax + depth-to-space heat-map, descriptor L2 normalisation).
//
// conv_mfma: implicit GEMM, M = 128 output pixels (8x16 tile, 32 per wave), N = 64 output channels per
// workgroup (two 32-wide MFMA tiles per wave), K = taps x CIN in chunks of CC channels staged in LDS with a
// 4-float pad per pixel (16 lanes x ds_read_b128 then hit 16 distinct 4-bank groups).  Lane (p, h) owns
// pixel p and, of every chunk, channels h*CC/2 .. +CC/2: the A operand is CC/8 ds_read_b128 per tap and the
// weights are pre-packed in exactly that fragment order so B is CC/8 global float4 loads per 32-wide tile.
// The 2x2 max-pools of the VGG trunk are fused: on the input read (POOL_IN) or lane-locally on the
// accumulators (POOL_OUT: the four pixels of a pool window live in registers r, r+1, r+8, r+9 of one lane).
#include "conv_mfma.h"

namespace {

// ------------------------------------------------------------------------------------------------
// VALU convolution for thin layers (CIN or COUT not MFMA shaped): one pixel x 8 output channels per thread,
// weights [tap][cin][COUT8] wave-uniform (scalar loads), inputs straight from L1/L2.
struct ConvV {
    const float* in; float* out; const float* w; const float* bias;
    const float* xf;    // optional [B][CIN][4] input transform (see ConvM)
    int Hi, Wi, H, W, CIN, COUT, COUT8, KS, S, PAD, relu;
    // conv_valu_t only, optional: XFeat's skip connection (XFeat.py:27-28, 127) added in the epilogue of the layer it joins --
    // out[c] += skw[c] * avg_pool4(gray)(pixel) + skb[c] for c < skn; gray is [B][4 H][4 W]
    const float* gray = nullptr; const float* skw = nullptr; const float* skb = nullptr; int skn = 0;
};

__global__ __launch_bounds__(256) void conv_valu(ConvV a)
{
    const int b = blockIdx.z, cg = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= a.H * a.W) return;
    const int oy = pix / a.W, ox = pix - oy * a.W;
    const float* in = a.in + (size_t)b * a.Hi * a.Wi * a.CIN;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = a.bias[cg * 8 + j];
    for (int ky = 0; ky < a.KS; ++ky) {
        const int iy = oy * a.S + ky - a.PAD;
        if (iy < 0 || iy >= a.Hi) continue;
        for (int kx = 0; kx < a.KS; ++kx) {
            const int ix = ox * a.S + kx - a.PAD;
            if (ix < 0 || ix >= a.Wi) continue;
            const float* src = in + ((size_t)iy * a.Wi + ix) * a.CIN;
            const float* w = a.w + ((size_t)(ky * a.KS + kx) * a.CIN) * a.COUT8 + cg * 8;
            for (int c = 0; c < a.CIN; ++c) {
                float v = src[c];
                if (a.xf) {
                    const float4 t = *reinterpret_cast<const float4*>(a.xf + ((size_t)b * a.CIN + c) * 4);
                    v = fmaf(v, t.x, t.y);
                    v = v >= 0.0f ? v : v * t.z;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(v, w[(size_t)c * a.COUT8 + j], acc[j]);
            }
        }
    }
    float* o = a.out + ((size_t)b * a.H * a.W + pix) * a.COUT + cg * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (cg * 8 + j < a.COUT) o[j] = a.relu ? relu(acc[j]) : acc[j];
}

// The same layer with kernel size, input channels and stride known at compile time (XFeat's block1: 1->4, 4->8/2, 8->8,
// 8->24/2, all 3x3): taps and channels unroll, a pixel's channels arrive as float4 loads, the weights of a tap are one
// scalar burst.  Bounds are tested per tap (zero padding), as in conv_valu.
// CREAL < CPT: the last CPT - CREAL channels are padding (zero weights and bias: XFeat's 24-channel stage is stored 32 wide for the
// MFMA layers behind it) -- they are stored as the zeros they are without being multiplied.
template <int KS, int CIN, int S, int CPT = 8, int CREAL = CPT>      // CPT output channels per thread: the taps are read once per CPT
__global__ __launch_bounds__(256) void conv_valu_t(ConvV a)
{
    constexpr int PAD = KS / 2;
    const int b = blockIdx.z, cg = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= a.H * a.W) return;
    const int oy = pix / a.W, ox = pix - oy * a.W;
    const float* in = a.in + (size_t)b * a.Hi * a.Wi * CIN;
    float acc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[j] = j < CREAL ? a.bias[cg * CPT + j] : 0.0f;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
        const int iy = oy * S + ky - PAD;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const int ix = ox * S + kx - PAD;
            const bool inside = iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
            const float* src = in + ((size_t)(inside ? iy : 0) * a.Wi + (inside ? ix : 0)) * CIN;
            float v[CIN];
            if (CIN % 4 == 0) {
#pragma unroll
                for (int q = 0; q < CIN / 4; ++q) {
                    const float4 f = *reinterpret_cast<const float4*>(src + 4 * q);
                    v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
                }
            } else {
#pragma unroll
                for (int c = 0; c < CIN; ++c) v[c] = src[c];
            }
            const float* w = a.w + ((size_t)(ky * KS + kx) * CIN) * a.COUT8 + cg * CPT;
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float vc = inside ? v[c] : 0.0f;
#pragma unroll
                for (int j = 0; j < CREAL; ++j) acc[j] = fmaf(vc, w[(size_t)c * a.COUT8 + j], acc[j]);
            }
        }
    }
    float* o = a.out + ((size_t)b * a.H * a.W + pix) * a.COUT + cg * CPT;
    if (a.relu) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc[j] = relu(acc[j]);
    }
    if (a.gray) {       // the skip connection: AvgPool2d(4) of the normalised grey image -> Conv2d(1, skn, 1), added AFTER the ReLU
        const int Wg = 4 * a.W;
        const float* g = a.gray + (size_t)b * (4 * a.H) * Wg + (size_t)(4 * oy) * Wg + 4 * ox;       // W is a multiple of 32: rows are 16-byte aligned
        float sacc = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const float4 r = *reinterpret_cast<const float4*>(g + (size_t)dy * Wg);
            sacc += r.x; sacc += r.y; sacc += r.z; sacc += r.w;                        // the order avg_pool2d's sum is restated in
        }
        const float avg = sacc * (1.0f / 16.0f);
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            if (cg * CPT + j < a.skn) acc[j] += fmaf(avg, a.skw[cg * CPT + j], a.skb[cg * CPT + j]);
    }
    if (a.COUT % 4 == 0 && cg * CPT + CPT <= a.COUT) {
#pragma unroll
        for (int q = 0; q < CPT / 4; ++q)
            *reinterpret_cast<float4*>(o + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            if (cg * CPT + j < a.COUT) o[j] = acc[j];
    }
}

// XFeat block1.0 (1 -> 4, 3x3) + block1.1 (4 -> 8, 3x3, stride 2) in one kernel (r04; XFeat.py:30-31): the 4-channel full-resolution map
// (4.9 MB per image written by one kernel and read back by the next: 0.89 + 0.84 ms per 512 images) stays in LDS.  A workgroup owns
// 8 x 32 outputs of block1.1 = 16 x 64 pixels (22.7 KB of LDS: seven workgroups per CU; 16 x 32 outputs, 43.7 KB, three per CU: 1.21 ms): it stages the 19 x 67 grey pixels they depend on, evaluates block1.0 + ReLU on the
// 17 x 65 positions block1.1 taps (8 % more than it owns; positions outside the image are block1.1's zero padding, not block1.0 of
// zero-padded grey), then block1.1 + ReLU, two outputs per thread.  Every output is accumulated in conv_valu_t's order -- bias, then
// taps in (ky, kx, cin) order, an fma each, zero for a tap outside -- so the maps are bit for bit those of the two-kernel form.
constexpr int XB_TH = 8, XB_TW = 32, XB_AH = 2 * XB_TH + 1, XB_AW = 2 * XB_TW + 1, XB_GH = XB_AH + 2, XB_GW = XB_AW + 2;
__global__ __launch_bounds__(256) void xfeat_block1_01(const float* __restrict__ gray, float* __restrict__ out, const float* __restrict__ w0,
                                                       const float* __restrict__ b0, const float* __restrict__ w1, const float* __restrict__ b1,
                                                       int H, int W, const float2* __restrict__ mr /* null: gray is normalised already */)
{
    __shared__ float g[XB_GH * XB_GW];
    __shared__ __attribute__((aligned(16))) float4 a1[XB_AH * XB_AW];
    const int tid = threadIdx.x, b = blockIdx.z;
    const int H2 = H / 2, W2 = W / 2;
    const int oy0 = blockIdx.y * XB_TH, ox0 = blockIdx.x * XB_TW;        // first output of the tile (half resolution)
    const int ay0 = 2 * oy0 - 1, ax0 = 2 * ox0 - 1;                      // first block1.0 position block1.1 taps
    const int gy0 = ay0 - 1, gx0 = ax0 - 1;                              // first grey pixel block1.0 taps
    const float* gi = gray + (size_t)b * H * W;
    const float2 nm = mr ? mr[b] : make_float2(0.0f, 1.0f);
    for (int i = tid; i < XB_GH * XB_GW; i += 256) {
        const int y = i / XB_GW, x = i - y * XB_GW, gy = gy0 + y, gx = gx0 + x;
        float v = 0.0f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            v = gi[(size_t)gy * W + gx];
            if (mr) v = (v - nm.x) * nm.y;          // InstanceNorm2d(1) applied on the way in; the zero padding is of the NORMALISED image
        }
        g[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < XB_AH * XB_AW; i += 256) {
        const int y = i / XB_AW, x = i - y * XB_AW, py = ay0 + y, px = ax0 + x;
        float acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = b0[j];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float v = g[(y + ky) * XB_GW + x + kx];           // zero outside the image: block1.0's padding
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(v, w0[(ky * 3 + kx) * 8 + j], acc[j]);
            }
        const bool inside = py >= 0 && py < H && px >= 0 && px < W;
        a1[i] = inside ? make_float4(relu(acc[0]), relu(acc[1]), relu(acc[2]), relu(acc[3])) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < XB_TH * XB_TW / 256; ++k) {
        const int o = tid + 256 * k, ly = o / XB_TW, lx = o - ly * XB_TW;
        const int oy = oy0 + ly, ox = ox0 + lx;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = b1[j];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 f = a1[(2 * ly + ky) * XB_AW + 2 * lx + kx];
                const float v[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(v[c], w1[((ky * 3 + kx) * 4 + c) * 8 + j], acc[j]);
            }
        if (oy < H2 && ox < W2) {
            float* q = out + (((size_t)b * H2 + oy) * W2 + ox) * 8;
            *reinterpret_cast<float4*>(q) = make_float4(relu(acc[0]), relu(acc[1]), relu(acc[2]), relu(acc[3]));
            *reinterpret_cast<float4*>(q + 4) = make_float4(relu(acc[4]), relu(acc[5]), relu(acc[6]), relu(acc[7]));
        }
    }
}

// XFeat block1.2 (8 -> 8, 3x3) + block1.3 (8 -> 24, 3x3, stride 2) + the skip connection, on the matrix cores (r04; XFeat.py:32-33,
// 27-28, 127).  As fp32 vector kernels the two layers were bound by vector issue (576 / 1 728 FMAs per output pixel: 1.04 + 0.96 ms per
// 512 images against 0.3 ms of traffic each) and the 8-channel half-resolution map between them went through HBM.  Here both are
// v_mfma_f32_16x16x32_f16 products on split operands (x = hi + lo halves, three MFMAs per product, fp32 accumulation: the arithmetic of
// every other layer of the net from block2 on), the map between them stays in LDS:
//   conv A (8 -> 8, stride 1), alike_block1_h's conv2 form: one accumulator column stands for a PAIR of horizontally adjacent positions,
//     N = (position of the pair, output channel), K = the 3 x 4 input window the pair shares = three 32-deep k-blocks (kb = window row,
//     lane group g = window column); the weight of window cell (ky, kx) for position s is w[ky][kx - s], zero outside the kernel.
//   conv B (8 -> 24, stride 2): M = 16 output channels (two tiles: 24 real rows of 32), N = 16 adjacent outputs of a row, K = three
//     k-blocks (kb = ky; lane group = kx, the fourth group's weights are zero).
// Both read their input from LDS planes [hi | lo][column parity][row][column / 2] of 16-byte slots (8 channels of one position): the 16
// lanes of a read group hit 16 consecutive slots whether neighbouring lanes are two columns apart (pairs; stride 2) or one.
// Operand range: the tile is split at the scale of its own largest input magnitude (cm_scale_of), conv A's output at the scale of its
// bound amax l1 + bmax; both are divided out of the fp32 accumulators again, exactly.  Workgroup = 8 x 16 outputs of block1.3.
typedef float f32x4m __attribute__((ext_vector_type(4)));
struct XfB1Args {
    const float* in;        // block1.1's output [B][H2][W2][8]
    float* out;             // x1 [B][H4][W4][32]: 24 channels + 8 zeros (the MFMA layers behind it read 32)
    const uint4* wA;        // [3 kb][hi / lo][64 lanes]: conv A pair fragments, scaled by 1 / inv_wsA
    const uint4* wB;        // [2 m-tiles][3 kb][hi / lo][64 lanes]: conv B fragments, scaled by 1 / inv_wsB
    const float* bA; const float* bB;       // biases [8], [32] (zero padded)
    const float* gray; const float* skw; const float* skb;      // the skip connection: avg_pool4(gray) * skw[c] + skb[c], c < 24, added after the ReLU
    const float2* mr;       // per image (mean, 1 / std) of the grey image, applied to the patch as it is read; null: gray is normalised already
    int H2, W2;
    float inv_wsA, inv_wsB, l1A, bmaxA;
};
constexpr int XQ_TH = 8, XQ_TW = 16;                               // outputs per workgroup (quarter resolution)
constexpr int XQ_MH = 2 * XQ_TH + 1, XQ_MW = 2 * XQ_TW + 1;         // conv A positions conv B taps (half resolution)
constexpr int XQ_IH = XQ_MH + 2, XQ_IW = XQ_MW + 2;                 // input positions conv A taps
constexpr int XQ_HWM = (XQ_MW + 1) / 2, XQ_HWI = (XQ_IW + 1) / 2;   // slots per row and parity
constexpr int XQ_PLM = XQ_MH * XQ_HWM, XQ_REGM = 2 * XQ_PLM, XQ_PLI = XQ_IH * XQ_HWI, XQ_REGI = 2 * XQ_PLI;

__global__ __launch_bounds__(256, 4) void xfeat_block1_23(XfB1Args a)
{
    __shared__ __attribute__((aligned(16))) uint4 pin[2 * XQ_REGI];      // [hi | lo][parity][row][column / 2]
    __shared__ __attribute__((aligned(16))) uint4 mid[2 * XQ_REGM];
    __shared__ __attribute__((aligned(16))) float s_amax[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.z;
    const int H4 = a.H2 / 2, W4 = a.W2 / 2;
    const int oy0 = blockIdx.y * XQ_TH, ox0 = blockIdx.x * XQ_TW;        // first output of the tile
    const int my0 = 2 * oy0 - 1, mx0 = 2 * ox0 - 1;                       // first conv A position (half resolution)
    const int iy0 = my0 - 1, ix0 = mx0 - 1;                               // first input position
    const float* in = a.in + (size_t)b * a.H2 * a.W2 * 8;
    cm_h8 wAh[3], wAl[3];
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
        wAh[kb] = __builtin_bit_cast(cm_h8, a.wA[(kb * 2 + 0) * 64 + lane]);
        wAl[kb] = __builtin_bit_cast(cm_h8, a.wA[(kb * 2 + 1) * 64 + lane]);
    }
    const int pr = lane & 15, g = lane >> 4;
    const float4 biasA = *reinterpret_cast<const float4*>(a.bA + 4 * (g & 1));
    int e_in;
    {   // stage the input tile: a thread takes half positions (4 channels), all its loads in flight before the first use
        constexpr int NH = XQ_IH * XQ_IW * 2, PER = (NH + 255) / 256;
        float4 ld[PER];
        float am = 0.0f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * 256, pos = i >> 1, y = pos / XQ_IW, x = pos - y * XQ_IW;
            const int gy = iy0 + y, gx = ix0 + x;
            const bool ok = i < NH && gy >= 0 && gy < a.H2 && gx >= 0 && gx < a.W2;
            ld[k] = ok ? *reinterpret_cast<const float4*>(in + ((size_t)gy * a.W2 + gx) * 8 + 4 * (i & 1)) : make_float4(0.f, 0.f, 0.f, 0.f);
            am = cm_amax4(am, ld[k]);
        }
        am = cm_wave_max(am);
        if (lane == 0) s_amax[wv] = am;
        // the pad column (odd parity, last slot of every row) is read by the last pair's window: finite zeros
        for (int i = tid; i < 2 * XQ_IH; i += 256) pin[(i / XQ_IH) * XQ_REGI + XQ_PLI + (i % XQ_IH) * XQ_HWI + XQ_HWI - 1] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        const float4 q = *reinterpret_cast<const float4*>(s_amax);
        e_in = cm_exp_of(fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
        const float sc = cm_scale_of(e_in);
        uint2* pinh = reinterpret_cast<uint2*>(pin);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * 256, pos = i >> 1, y = pos / XQ_IW, x = pos - y * XQ_IW;
            uint2 hi, lo;
            cm_split4(make_float4(ld[k].x * sc, ld[k].y * sc, ld[k].z * sc, ld[k].w * sc), hi, lo);
            if (i < NH) {
                const int h8 = (((x & 1) * XQ_PLI + y * XQ_HWI + (x >> 1)) << 1) + (i & 1);
                pinh[h8] = hi;
                pinh[h8 + 2 * XQ_REGI] = lo;
            }
        }
    }
    __syncthreads();
    const int e_mid = cm_exp_of(fmaf(__uint_as_float((unsigned)(e_in - 126 + 127) << 23), a.l1A, a.bmaxA));
    const float unA = a.inv_wsA * cm_unscale_of(e_in), sc_mid = cm_scale_of(e_mid), unB = a.inv_wsB * cm_unscale_of(e_mid);
    const int sN = g >> 1;                            // conv A accumulator: position sN of the pair, channels 4 (g & 1) .. + 3
    {   // conv A + ReLU on the MH x MW positions, 16 pairs per MFMA group: group gi < MH = pairs 0..15 of row gi; the 17th pair of
        // every row goes to two extra groups (rows 0..15 and row 16)
        uint2* midh = reinterpret_cast<uint2*>(mid);
#pragma unroll 1
        for (int gi = wv; gi < XQ_MH + (XQ_MH + 15) / 16; gi += 4) {
            const bool extra = gi >= XQ_MH;
            const int y = extra ? 16 * (gi - XQ_MH) + pr : gi, pc = extra ? 16 : pr;
            const int my = min(y, XQ_MH - 1);
            f32x4m acc = {0.f, 0.f, 0.f, 0.f};
            const int abase = (g & 1) * XQ_PLI + my * XQ_HWI + pc + (g >> 1);
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                const cm_h8 ihi = __builtin_bit_cast(cm_h8, pin[abase + kb * XQ_HWI]);
                const cm_h8 ilo = __builtin_bit_cast(cm_h8, pin[XQ_REGI + abase + kb * XQ_HWI]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wAh[kb], ilo, acc, 0, 0, 0);      // small terms first
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wAl[kb], ihi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wAh[kb], ihi, acc, 0, 0, 0);
            }
            const int gy = my0 + y, gx = mx0 + 2 * pc + sN;
            const bool inside = gy >= 0 && gy < a.H2 && gx >= 0 && gx < a.W2;       // conv B pads its INPUT (the ReLU'd map) with zeros
            float4 v = make_float4(relu(fmaf(acc[0], unA, biasA.x)) * sc_mid, relu(fmaf(acc[1], unA, biasA.y)) * sc_mid,
                                   relu(fmaf(acc[2], unA, biasA.z)) * sc_mid, relu(fmaf(acc[3], unA, biasA.w)) * sc_mid);
            if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
            uint2 hi, lo;
            cm_split4(v, hi, lo);
            if (y < XQ_MH) {
                const int h8 = ((sN * XQ_PLM + y * XQ_HWM + pc) << 1) + (g & 1);
                midh[h8] = hi;
                midh[h8 + 2 * XQ_REGM] = lo;
            }
        }
    }
    cm_h8 wBh[2][3], wBl[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
            wBh[mt][kb] = __builtin_bit_cast(cm_h8, a.wB[((mt * 3 + kb) * 2 + 0) * 64 + lane]);
            wBl[mt][kb] = __builtin_bit_cast(cm_h8, a.wB[((mt * 3 + kb) * 2 + 1) * 64 + lane]);
        }
    float4 biasB[2], skw[2], skb[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int c = 16 * mt + 4 * g;
        biasB[mt] = *reinterpret_cast<const float4*>(a.bB + c);
        skw[mt] = c < 24 ? *reinterpret_cast<const float4*>(a.skw + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        skb[mt] = c < 24 ? *reinterpret_cast<const float4*>(a.skb + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // conv B + ReLU + skip: a wave owns two output rows; lane (n = output column, g): tap column g of row 2 ly + kb at column 2 n + g
    const int bbase = (g & 1) * XQ_PLM + pr + (g >> 1);
    const int ox = ox0 + pr, Wg = 2 * a.W2;
    const float* gimg = a.gray + (size_t)b * (2 * a.H2) * Wg;
    const float2 nm = a.mr ? a.mr[b] : make_float2(0.0f, 1.0f);
#pragma unroll
    for (int rr = 0; rr < XQ_TH / 4; ++rr) {
        const int ly = (XQ_TH / 4) * wv + rr, oy = oy0 + ly;
        const bool live = oy < H4 && ox < W4;
        // the skip term's 4 x 4 grey patch is requested before the products, by the first lane group only (all four groups of a column
        // need the same average).  Requested earlier -- at the top of the kernel, or once the input tile has landed -- the kernel was
        // 6-12 % slower.
        float4 gp[4];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
            gp[dy] = (live && g == 0) ? *reinterpret_cast<const float4*>(gimg + (size_t)(4 * oy + dy) * Wg + 4 * ox) : make_float4(0.f, 0.f, 0.f, 0.f);
        f32x4m acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
            const int at = bbase + (2 * ly + kb) * XQ_HWM;
            const cm_h8 ihi = __builtin_bit_cast(cm_h8, mid[at]);
            const cm_h8 ilo = __builtin_bit_cast(cm_h8, mid[XQ_REGM + at]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wBh[mt][kb], ilo, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wBl[mt][kb], ihi, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wBh[mt][kb], ihi, acc[mt], 0, 0, 0);
            }
        }
        float sacc = 0.0f;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {                    // avg_pool2d's order (conv_valu_t), over the normalised pixels
            if (a.mr) { gp[dy].x = (gp[dy].x - nm.x) * nm.y; gp[dy].y = (gp[dy].y - nm.x) * nm.y; gp[dy].z = (gp[dy].z - nm.x) * nm.y; gp[dy].w = (gp[dy].w - nm.x) * nm.y; }
            sacc += gp[dy].x; sacc += gp[dy].y; sacc += gp[dy].z; sacc += gp[dy].w;
        }
        const float avg = __shfl(sacc * (1.0f / 16.0f), pr, 64);
        if (live) {
            float* o = a.out + (((size_t)b * H4 + oy) * W4 + ox) * 32 + 4 * g;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float4 v = make_float4(relu(fmaf(acc[mt][0], unB, biasB[mt].x)), relu(fmaf(acc[mt][1], unB, biasB[mt].y)),
                                       relu(fmaf(acc[mt][2], unB, biasB[mt].z)), relu(fmaf(acc[mt][3], unB, biasB[mt].w)));
                if (16 * mt + 4 * g < 24) {
                    v.x += fmaf(avg, skw[mt].x, skb[mt].x); v.y += fmaf(avg, skw[mt].y, skb[mt].y);
                    v.z += fmaf(avg, skw[mt].z, skb[mt].z); v.w += fmaf(avg, skw[mt].w, skb[mt].w);
                }
                *reinterpret_cast<float4*>(o + 16 * mt) = v;
            }
        }
    }
}

// conv A of xfeat_block1_23, OIHW [8][8][3][3] -> [3 kb = ky][hi / lo][64 lanes][8 halves = cin]: lane (n = (s, cout), g = column of the
// 3 x 4 window) holds w[cout][cin][ky][g - s], zero outside the kernel (alike.hip's pack_b1c2_pairs: the same product)
std::vector<float> pack_xf_pairs(const float* w, float scale)
{
    std::vector<uint16_t> hl((size_t)3 * 2 * 64 * 8, 0);
    for (int kb = 0; kb < 3; ++kb)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int n = l & 15, g = l >> 4, s2 = n >> 3, co = n & 7, kx = g - s2;
                const float v = (kx >= 0 && kx <= 2) ? w[((size_t)co * 8 + j) * 9 + kb * 3 + kx] * scale : 0.0f;
                const uint16_t hi = f16_bits_rtn(v), lo = f16_bits_rtn(v - f16_bits_to_float(hi));
                hl[(((size_t)kb * 2 + 0) * 64 + l) * 8 + j] = hi;
                hl[(((size_t)kb * 2 + 1) * 64 + l) * 8 + j] = lo;
            }
    std::vector<float> out(hl.size() / 2);
    std::memcpy(out.data(), hl.data(), hl.size() * 2);
    return out;
}

// conv B, OIHW [COUT <= 32][8][3][3] -> [2 m-tiles][3 kb = ky][hi / lo][64 lanes][8 halves = cin]: lane (m = output channel of the tile,
// kq = kx; kq = 3 and channels >= COUT are zero)
std::vector<float> pack_xf_s2(const float* w, int cout, float scale)
{
    std::vector<uint16_t> hl((size_t)2 * 3 * 2 * 64 * 8, 0);
    for (int mt = 0; mt < 2; ++mt)
        for (int kb = 0; kb < 3; ++kb)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int m = l & 15, kq = l >> 4, co = 16 * mt + m;
                    const float v = (kq <= 2 && co < cout) ? w[((size_t)co * 8 + j) * 9 + kb * 3 + kq] * scale : 0.0f;
                    const uint16_t hi = f16_bits_rtn(v), lo = f16_bits_rtn(v - f16_bits_to_float(hi));
                    hl[((((size_t)mt * 3 + kb) * 2 + 0) * 64 + l) * 8 + j] = hi;
                    hl[((((size_t)mt * 3 + kb) * 2 + 1) * 64 + l) * 8 + j] = lo;
                }
    std::vector<float> out(hl.size() / 2);
    std::memcpy(out.data(), hl.data(), hl.size() * 2);
    return out;
}

// SuperPoint conv1a (1 -> 64, 3x3, ReLU; SuperPoint.py:44): 16 lanes share a pixel, each lane keeps the 9 taps of its
// 4 output channels in registers and walks down a column of pixels, so a wave store is 4 whole 256-byte pixels.
// (The split-f16 path never launches it: conv1b generates these channels while it stages its tile, conv_mfma_h<.., GEN>; this kernel is the
// strict-fp32 path's conv1a.)
__global__ __launch_bounds__(256) void conv1a_c64(const float* gray, float* out, const float* w /*[9][64]*/, const float* bias, int H, int W, int rows_per_block)
{
    const int b = blockIdx.z, lane16 = threadIdx.x & 15, c4 = lane16 * 4;
    const int x = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (x >= W) return;
    float wr[9][4], bv[4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[t][j] = w[t * 64 + c4 + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = bias[c4 + j];
    const float* g = gray + (size_t)b * H * W;
    const int y0 = blockIdx.y * rows_per_block, y1 = min(y0 + rows_per_block, H);
    auto ld = [&](int yy, int xx) { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? g[(size_t)yy * W + xx] : 0.0f; };
    // the row below is requested one iteration before it is used (requested and used in the same iteration, every row of the column
    // walk waited for its own L2 round trip)
    float r0[3], r1[3], r2[3], r3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { r0[k] = ld(y0 - 1, x - 1 + k); r1[k] = ld(y0, x - 1 + k); r2[k] = ld(y0 + 1, x - 1 + k); }
    for (int y = y0; y < y1; ++y) {
#pragma unroll
        for (int k = 0; k < 3; ++k) r3[k] = ld(y + 2, x - 1 + k);
        float acc[4] = {bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = fmaf(r0[k], wr[k][j], acc[j]);
                acc[j] = fmaf(r1[k], wr[3 + k][j], acc[j]);
                acc[j] = fmaf(r2[k], wr[6 + k][j], acc[j]);
            }
        *reinterpret_cast<float4*>(out + (((size_t)b * H + y) * W + x) * 64 + c4) = make_float4(relu(acc[0]), relu(acc[1]), relu(acc[2]), relu(acc[3]));
#pragma unroll
        for (int k = 0; k < 3; ++k) { r0[k] = r1[k]; r1[k] = r2[k]; r2[k] = r3[k]; }
    }
}

// ------------------------------------------------------------------------------------------------ small kernels
// SuperPoint.py:42  x = torch.sum(x, dim=1, keepdim=True)
__global__ void rgb_sum(const float* img, float* gray, size_t P)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t b = blockIdx.y;
    if (i < P) gray[b * P + i] = (img[(b * 3 + 0) * P + i] + img[(b * 3 + 1) * P + i]) + img[(b * 3 + 2) * P + i];
}

// largest |value| of each image of a single-channel map, as float bits (one workgroup per image: 1.2 MB at 480 x 640; the input of the
// bound the pre-split layers scale by).  Non-finite values are skipped, as amax_reduce does.
__global__ __launch_bounds__(1024) void plane_abs_max(const float* __restrict__ g, size_t P, unsigned* out)
{
    __shared__ float s[16];
    const float* p = g + (size_t)blockIdx.x * P;
    float m = 0.0f;
    for (size_t i = (size_t)threadIdx.x * 4; i + 3 < P; i += 4096) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
        m = fmaxf(m, fmaxf(fmaxf(a0 < INFINITY ? a0 : 0.0f, a1 < INFINITY ? a1 : 0.0f), fmaxf(a2 < INFINITY ? a2 : 0.0f, a3 < INFINITY ? a3 : 0.0f)));
    }
    for (size_t i = (P & ~(size_t)3) + threadIdx.x; i < P; i += 1024) { const float a0 = fabsf(p[i]); m = fmaxf(m, a0 < INFINITY ? a0 : 0.0f); }
    m = cm_wave_max(m);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.0f;
        for (int i = 0; i < 16; ++i) r = fmaxf(r, s[i]);
        out[blockIdx.x] = __float_as_uint(r);
    }
}

// SuperPoint.py:64-69: softmax over the 65 logits of a cell, drop the dustbin, depth-to-space 8x8.
// One wave per cell: lane c holds logit c, lane 0 also folds in logit 64.  A wave takes PXW cells in a row, all their loads in
// flight before the first reduction: with one 260-byte cell per wave the kernel had 8 KB in flight per CU and ran at a quarter of
// its HBM bound (r03).  Per cell the arithmetic is unchanged.
constexpr int PXW = 8;
__global__ __launch_bounds__(256) void softmax65_d2s(const float* __restrict__ semi, float* __restrict__ heat, int Hc, int Wc, int ncell)
{
    const int lane = threadIdx.x & 63;
    const int cell0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PXW;
    const int b = blockIdx.y;
    if (cell0 >= ncell) return;
    float v[PXW], d[PXW];
#pragma unroll
    for (int i = 0; i < PXW; ++i) {
        const float* s = semi + ((size_t)b * ncell + min(cell0 + i, ncell - 1)) * 65;
        v[i] = s[lane]; d[i] = s[64];
    }
    const int W = Wc * 8;
#pragma unroll
    for (int i = 0; i < PXW; ++i) {
        const int cell = cell0 + i;
        float m = kpb_wave_fmax(v[i]);
        m = fmaxf(m, d[i]);
        const float e = expf(v[i] - m), ed = expf(d[i] - m);
        float sum = kpb_wave_sum(e);
        sum += ed;
        const int cy = cell / Wc, cx = cell - cy * Wc;
        if (cell < ncell) heat[(size_t)b * Hc * 8 * W + (size_t)(cy * 8 + (lane >> 3)) * W + cx * 8 + (lane & 7)] = __fdiv_rn(e, sum);
    }
}

// SuperPoint.py:61-62: desc / ||desc||_2 over the C channels of each pixel (no epsilon).  One wave per pixel, PXW pixels per wave
// when C = 64 (XFeat: one value per lane; see softmax65_d2s).
__global__ __launch_bounds__(256) void l2norm_nhwc(float* desc, int C, size_t npix, float eps_clamp)
{
    const int lane = threadIdx.x & 63;
    if (C == 64) {
        const size_t pix0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PXW;
        if (pix0 >= npix) return;
        float x[PXW];
#pragma unroll
        for (int i = 0; i < PXW; ++i) x[i] = desc[min(pix0 + i, npix - 1) * 64 + lane];
#pragma unroll
        for (int i = 0; i < PXW; ++i) {
            float ss = fmaf(x[i], x[i], 0.0f);
            ss = kpb_wave_sum(ss);
            float n = sqrtf(ss);
            if (eps_clamp > 0.0f) n = fmaxf(n, eps_clamp);
            if (pix0 + i < npix) desc[(pix0 + i) * 64 + lane] = __fdiv_rn(x[i], n);
        }
        return;
    }
    const size_t pix = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= npix) return;
    float* d = desc + pix * C;
    float ss = 0.0f;
    for (int c = lane; c < C; c += 64) ss = fmaf(d[c], d[c], ss);
    ss = kpb_wave_sum(ss);
    float n = sqrtf(ss);
    if (eps_clamp > 0.0f) n = fmaxf(n, eps_clamp);     // F.normalize clamps; torch.norm + div does not
    for (int c = lane; c < C; c += 64) d[c] = __fdiv_rn(d[c], n);
}

// ------------------------------------------------------------------------------------------------ XFeat pieces
// XFeat.py:121-123: x.mean(dim=1) then InstanceNorm2d(1): (x - mean) / sqrt(var + 1e-5), biased variance per image.
// r05: every block leaves its partial (sum, sum of squares) in stats[image][block]; instnorm_params adds the XF_STAT_BLOCKS pairs in block order --
// the same fp64 sums every run (they were fp64 atomics in arrival order).
constexpr int XF_STAT_BLOCKS = 64;
__global__ __launch_bounds__(256) void gray_mean_stats(const float* img, float* gray, double* stats, size_t P)
{
    __shared__ double s1[4], s2[4];
    const size_t b = blockIdx.y;
    double a = 0.0, q = 0.0;
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < P; i += 4 * step) {          // twelve loads in flight per thread, accumulated in the order of the plain loop
        float c[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) c[k][ch] = img[(b * 3 + ch) * P + i + k * step];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g = ((c[k][0] + c[k][1]) + c[k][2]) / 3.0f;
            gray[b * P + i + k * step] = g;
            a += (double)g; q += (double)g * (double)g;
        }
    }
    for (; i < P; i += step) {
        const float g = ((img[(b * 3 + 0) * P + i] + img[(b * 3 + 1) * P + i]) + img[(b * 3 + 2) * P + i]) / 3.0f;
        gray[b * P + i] = g;
        a += (double)g; q += (double)g * (double)g;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }
    if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = a; s2[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[(b * gridDim.x + blockIdx.x) * 2] = s1[0] + s1[1] + s1[2] + s1[3];
        stats[(b * gridDim.x + blockIdx.x) * 2 + 1] = s2[0] + s2[1] + s2[2] + s2[3];
    }
}

__global__ void instnorm_apply(float* gray, const float2* mr, size_t P)       // four pixels per thread (P is a multiple of 1024: H and W are multiples of 32)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4, b = blockIdx.y;
    if (i >= P) return;
    const float m = mr[b].x, r = mr[b].y;        // instnorm_params' (mean, 1 / std)
    float4* g = reinterpret_cast<float4*>(gray + b * P + i);
    float4 v = *g;
    v.x = (v.x - m) * r; v.y = (v.y - m) * r; v.z = (v.z - m) * r; v.w = (v.w - m) * r;
    *g = v;
}

// The same two numbers per image, for the kernels that normalise the grey image as they read it (r04: xfeat_block1_01, the skip term of
// xfeat_block1_23, the keypoint head's unfolded input -- instnorm_apply's pass over the image, 0.19 ms per 512 images, is gone for
// the split-f16 form; (x - m) * r is evaluated with the same two operations, so the values are bit for bit instnorm_apply's)
__global__ void instnorm_params(const double* stats, size_t P, float2* mr, int B)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    double s = 0.0, q = 0.0;
    for (int k0 = 0; k0 < XF_STAT_BLOCKS; k0 += 16) {      // block order, sixteen partial pairs in flight
        double2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const double2*>(stats + ((size_t)b * XF_STAT_BLOCKS + k0 + k) * 2);
#pragma unroll
        for (int k = 0; k < 16; ++k) { s += v[k].x; q += v[k].y; }
    }
    static_assert(XF_STAT_BLOCKS % 16 == 0, "instnorm_params folds sixteen partials at a time");
    const double mean = s / (double)P;
    const double var = fmax(q / (double)P - mean * mean, 0.0);
    mr[b] = make_float2((float)mean, 1.0f / sqrtf((float)var + 1e-5f));
}

// XFeat.py:133-135: x3 + interpolate(x4, size(x3), bilinear) + interpolate(x5, ...) (align_corners=False), 64 channels
__device__ __forceinline__ float4 bilerp4(const float* m, int Hs, int Ws, int Hd, int Wd, int y, int x, int c4)
{
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    const float fy = fmaxf(((float)y + 0.5f) * sy - 0.5f, 0.0f), fx = fmaxf(((float)x + 0.5f) * sx - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const float4 a = *reinterpret_cast<const float4*>(m + ((size_t)y0 * Ws + x0) * 64 + c4);
    const float4 b = *reinterpret_cast<const float4*>(m + ((size_t)y0 * Ws + x1) * 64 + c4);
    const float4 c = *reinterpret_cast<const float4*>(m + ((size_t)y1 * Ws + x0) * 64 + c4);
    const float4 d = *reinterpret_cast<const float4*>(m + ((size_t)y1 * Ws + x1) * 64 + c4);
    return make_float4(hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * d.x), hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * d.y),
                       hy * (hx * a.z + lx * b.z) + ly * (hx * c.z + lx * d.z), hy * (hx * a.w + lx * b.w) + ly * (hx * c.w + lx * d.w));
}

__global__ void pyramid_sum(const float* x3, const float* x4, const float* x5, float* out, int H8, int W8, int H16, int W16, int H32, int W32)
{
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= H8 * W8 * 16) return;
    const int pix = i >> 4, c4 = (i & 15) * 4;
    const int y = pix / W8, x = pix - y * W8;
    const float4 a = *reinterpret_cast<const float4*>(x3 + ((size_t)b * H8 * W8 + pix) * 64 + c4);
    const float4 u4 = bilerp4(x4 + (size_t)b * H16 * W16 * 64, H16, W16, H8, W8, y, x, c4);
    const float4 u5 = bilerp4(x5 + (size_t)b * H32 * W32 * 64, H32, W32, H8, W8, y, x, c4);
    *reinterpret_cast<float4*>(out + ((size_t)b * H8 * W8 + pix) * 64 + c4) =
        make_float4((a.x + u4.x) + u5.x, (a.y + u4.y) + u5.y, (a.z + u4.z) + u5.z, (a.w + u4.w) + u5.w);
}

// XFeat.py:96-103 _unfold2d(x, ws=8): [B,1,H,W] -> [B,64,H/8,W/8], channel = wy*8 + wx (stored NHWC)
__global__ void unfold8(const float* gray, float* out, int H, int W)       // four channels = four adjacent pixels of a cell row per thread
{
    const int Wc = W / 8, Hc = H / 8;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4, b = blockIdx.y;
    if (i >= Hc * Wc * 64) return;
    const int cell = i >> 6, c = i & 63;
    const int cy = cell / Wc, cx = cell - cy * Wc;
    *reinterpret_cast<float4*>(out + ((size_t)b * Hc * Wc + cell) * 64 + c) =
        *reinterpret_cast<const float4*>(gray + (size_t)b * H * W + (size_t)(cy * 8 + (c >> 3)) * W + cx * 8 + (c & 7));
}

// ------------------------------------------------------------------------------------------------ host helpers
// OIHW -> [tap][cin][COUT8]
std::vector<float> pack_valu(const float* w, int COUT, int CIN, int KS)
{
    const int T = KS * KS, C8 = ((COUT + 7) / 8) * 8;
    std::vector<float> out((size_t)T * CIN * C8, 0.0f);
    for (int o = 0; o < COUT; ++o)
        for (int c = 0; c < CIN; ++c)
            for (int t = 0; t < T; ++t) out[((size_t)t * CIN + c) * C8 + o] = w[((size_t)o * CIN + c) * T + t];
    return out;
}

struct Layer {      // one convolution of a network plan
    std::string name;
    int cin, cout, ks, stride;
    bool mfma;
    int cc = 32;    // channels per LDS chunk of conv_mfma (32, or 16 for stride 2 / CIN not a multiple of 32)
    int ntb = 2;    // 32-wide output tiles per workgroup
};

struct UpSrc { const float* src; int c; };      // ConvM::up_src, up_c
struct PreSplit { const unsigned* amax = nullptr; float l1 = 0.0f, bmax = 0.0f; const float* gen_w = nullptr; const float* gen_b = nullptr; };     // ConvM::pre_*, gen_*

int launch_mfma(kpb_ctx* ctx, const char* name, kpb_net* net, const Layer& L, const float* in, float* out, int B, int Hi, int Wi,
                bool pool_in, bool pool_out, bool relu_, const float* xf = nullptr, int unfold_w = 0, float l2_eps = 0.0f, const float2* unfold_mr = nullptr,
                const PreSplit* pre = nullptr, const UpSrc* up = nullptr)
{
    const int S = L.stride, CC = L.cc, PAD = L.ks / 2;
    const int Hc = pool_in ? Hi / 2 : Hi, Wc = pool_in ? Wi / 2 : Wi;
    ConvM a;
    a.in = in; a.out = out; a.wp = net->wp((L.name + ".w").c_str()); a.bias = net->wp((L.name + ".b").c_str()); a.xf = xf; a.res = nullptr;
    a.active = nullptr; a.istride = L.cin; a.ostride = L.cout; a.ooff = 0;
    a.Hi = Hi; a.Wi = Wi; a.unfold_w = unfold_w; a.aux0 = reinterpret_cast<const float*>(unfold_mr);
    if (unfold_w && !(conv_mfma_use_h16() && L.ks == 1 && S == 1 && CC == 32 && !pool_in && !pool_out && !xf && L.cin == 64))
        return kpb_fail(ctx, KPB_E_INVALID, "launch_mfma: the unfolded input exists for 64-channel 1x1 layers of the split-f16 form only");
    a.H = (Hc + 2 * PAD - L.ks) / S + 1; a.W = (Wc + 2 * PAD - L.ks) / S + 1;
    a.CIN = L.cin; a.COUT = L.cout; a.NCH = L.cin / CC; a.relu = relu_ ? 1 : 0; a.nblk = (L.cout + 32 * L.ntb - 1) / (32 * L.ntb);
    hipStream_t st = ctx->stream;
    const bool x = xf != nullptr;
    const dim3 block(256);
    if (conv_mfma_use_h16()) {     // split-f16 form: stride-1 layers on 16-row workgroup tiles; with two n-tiles per workgroup the waves
                                   // form 2 row groups x 2 n-tiles (conv_mfma_h<.., WN = 2>: four M tiles and one n-tile per wave)
        a.unscale = 1.0f / (net->wscale.at(L.name + ".w"));
        const dim3 g2(cdiv(a.W, 16), cdiv(a.H, 16), B * a.nblk), g1(cdiv(a.W, 16), cdiv(a.H, 8), B * a.nblk);
        // one-tile layers (cout <= 32) are bound by per-workgroup latency: 8-row tiles (one M tile per wave, 28 KB of LDS, five
        // workgroups per CU) measured 8-16 % faster; layers with two output tiles lose the fragment reuse that way (+8 % time)
        if (pre) {      // the input arrives already split (ConvM::pre_amax): raw bytes land in the tile by LDS-DMA
            a.pre_amax = pre->amax; a.pre_l1 = pre->l1; a.pre_bmax = pre->bmax;
            if (pre->gen_w && L.ks == 3 && S == 1 && CC == 32 && !pool_in && pool_out && !x && L.ntb == 2) {
                a.gen_w = pre->gen_w; a.gen_b = pre->gen_b; a.istride = 1;         // `in` is the one-channel image the layer in front reads
                KPB_LAUNCH(ctx, name, (conv_mfma_h<3, 1, 32, false, true, false, 1, 4, false, 2, false, true, 2, true>), g2, block, 0, st, a);
            }
            else return kpb_fail(ctx, KPB_E_INVALID, "conv_mfma_h: no generated-input instance for %s", L.name.c_str());
            return KPB_OK;
        }
        if (L.ks == 3 && S == 1 && CC == 32 && !pool_in && !pool_out && !x && L.ntb == 1) KPB_LAUNCH(ctx, name, (conv_mfma_h<3, 1, 32, false, false, false, 1, 1>), g1, block, 0, st, a);
        else if (L.ks == 3 && S == 1 && CC == 32 && !pool_in && !pool_out && !x) KPB_LAUNCH(ctx, name, (conv_mfma_h<3, 1, 32, false, false, false, 1, 4, false, 2, false, false, 2>), g2, block, 0, st, a);
        else if (L.ks == 3 && S == 1 && CC == 32 && !pool_in && pool_out && !x) KPB_LAUNCH(ctx, name, (conv_mfma_h<3, 1, 32, false, true, false, 1, 4, false, 2, false, false, 2>), g2, block, 0, st, a);
        else if (L.ks == 3 && S == 1 && CC == 32 && pool_in && !pool_out && !x) KPB_LAUNCH(ctx, name, (conv_mfma_h<3, 1, 32, true, false, false, 1, 4, false, 2, false, false, 2>), g2, block, 0, st, a);
        else if (L.ks == 5 && S == 1 && CC == 32 && !pool_in && !pool_out && x && up) {     // [up(bottom) | horizontal] evaluated while staging (ConvM::up_src)
            if (up->c % CC || up->c >= L.cin || (Hi % 2) || (Wi % 2)) return kpb_fail(ctx, KPB_E_INVALID, "launch_mfma: bad upsampled-input split for %s", L.name.c_str());
            a.up_src = up->src; a.up_c = up->c; a.istride = L.cin - up->c;
            KPB_LAUNCH(ctx, name, (conv_mfma_h<5, 1, 32, false, false, true, 1, 4, false, 2, false, false, 2, false, true>), g2, block, 0, st, a);
        }
        else if (L.ks == 5 && S == 1 && CC == 32 && !pool_in && !pool_out && x) KPB_LAUNCH(ctx, name, (conv_mfma_h<5, 1, 32, false, false, true, 1, 4, false, 2, false, false, 2>), g2, block, 0, st, a);
        else if (L.ks == 5 && S == 1 && CC == 16 && !pool_in && !pool_out && x && L.ntb == 2) KPB_LAUNCH(ctx, name, (conv_mfma_h<5, 1, 16, false, false, true, 1, 4, false, 2, false, false, 2>), g2, block, 0, st, a);
        else if (L.ks == 5 && S == 1 && CC == 16 && !pool_in && !pool_out && x && L.ntb == 5) {
            // 129 = 4 x 32 + 1 (DISK up_3): four MFMA tiles (two workgroups of two: 64 accumulator registers, three waves per SIMD)
            // and the score channel on the VALU of the first workgroup
            a.nblk = 2; a.xw = net->wp((L.name + ".xw").c_str()); a.xb = net->wscale.at(L.name + ".xb"); a.xun = 1.0f / net->wscale.at(L.name + ".xs"); a.xco = L.cout - 1;
            if (up) {       // `in` holds the channels behind the upsampled ones: cin - up->c floats per pixel
                if (up->c % CC || up->c >= L.cin || (Hi % 2) || (Wi % 2)) return kpb_fail(ctx, KPB_E_INVALID, "launch_mfma: bad upsampled-input split for %s", L.name.c_str());
                a.up_src = up->src; a.up_c = up->c; a.istride = L.cin - up->c;
                KPB_LAUNCH(ctx, name, (conv_mfma_h<5, 1, 16, false, false, true, 1, 4, true, 2, false, false, 2, false, true>), dim3(cdiv(a.W, 16), cdiv(a.H, 16), B * 2), block, 0, st, a);
            } else
            KPB_LAUNCH(ctx, name, (conv_mfma_h<5, 1, 16, false, false, true, 1, 4, true, 2, false, false, 2>), dim3(cdiv(a.W, 16), cdiv(a.H, 16), B * 2), block, 0, st, a);
        }
        else if (L.ks == 1 && S == 1 && CC == 32 && !pool_in && !pool_out && !x && l2_eps > 0.0f) {
            if (L.cout != 64 || L.ntb != 2) return kpb_fail(ctx, KPB_E_INVALID, "launch_mfma: the fused L2 norm needs a 64-channel layer");
            a.xb = l2_eps;
            KPB_LAUNCH(ctx, name, (gemm_h<2, 1, GE_L2NORM>), dim3(cdiv(a.H * a.W, 128), 1, B * a.nblk), block, 0, st, a);
        }
        else if (L.ks == 1 && S == 1 && CC == 32 && !pool_in && !pool_out && !x && unfold_w) KPB_LAUNCH(ctx, name, (gemm_h<2, 1, GE_PLAIN, true>), dim3(cdiv(a.H * a.W, 128), 1, B * a.nblk), block, 0, st, a);
        else if (L.ks == 1 && S == 1 && CC == 32 && !pool_in && !pool_out && !x) KPB_LAUNCH(ctx, name, (gemm_h<2, 1>), dim3(cdiv(a.H * a.W, 128), 1, B * a.nblk), block, 0, st, a);
        else if (L.ks == 3 && S == 2 && CC == 16 && !pool_in && !pool_out && !x) KPB_LAUNCH(ctx, name, (conv_mfma_h<3, 2, 16, false, false, false, 1, 2, false, 2, false, false, 2>), g1, block, 0, st, a);
        else return kpb_fail(ctx, KPB_E_INVALID, "conv_mfma_h: no instance for ks=%d stride=%d cc=%d pool_in=%d pool_out=%d xf=%d", L.ks, S, CC, pool_in, pool_out, x);
        return KPB_OK;
    }
    const dim3 grid(cdiv(a.W, 16), cdiv(a.H, 8), B * a.nblk);
    if (L.ks == 3 && S == 1 && CC == 32 && !pool_in && !pool_out && !x && L.ntb == 1) KPB_LAUNCH(ctx, name, (conv_mfma<3, 1, 32, false, false, false, 1>), grid, block, 0, st, a);
    else if (L.ks == 3 && S == 1 && CC == 32 && !pool_in && !pool_out && !x) KPB_LAUNCH(ctx, name, (conv_mfma<3, 1, 32, false, false, false>), grid, block, 0, st, a);
    else if (L.ks == 3 && S == 1 && CC == 32 && !pool_in && pool_out && !x) KPB_LAUNCH(ctx, name, (conv_mfma<3, 1, 32, false, true, false>), grid, block, 0, st, a);
    else if (L.ks == 3 && S == 1 && CC == 32 && pool_in && !pool_out && !x) KPB_LAUNCH(ctx, name, (conv_mfma<3, 1, 32, true, false, false>), grid, block, 0, st, a);
    else if (L.ks == 1 && S == 1 && CC == 32 && !pool_in && !pool_out && !x) KPB_LAUNCH(ctx, name, (conv_mfma<1, 1, 32, false, false, false>), grid, block, 0, st, a);
    else if (L.ks == 3 && S == 2 && CC == 16 && !pool_in && !pool_out && !x) KPB_LAUNCH(ctx, name, (conv_mfma<3, 2, 16, false, false, false>), grid, block, 0, st, a);
    else if (L.ks == 5 && S == 1 && CC == 32 && !pool_in && !pool_out && x) KPB_LAUNCH(ctx, name, (conv_mfma<5, 1, 32, false, false, true>), grid, block, 0, st, a);
    else if (L.ks == 5 && S == 1 && CC == 16 && !pool_in && !pool_out && x && L.ntb == 2) KPB_LAUNCH(ctx, name, (conv_mfma<5, 1, 16, false, false, true>), grid, block, 0, st, a);
    else if (L.ks == 5 && S == 1 && CC == 16 && !pool_in && !pool_out && x && L.ntb == 5) KPB_LAUNCH(ctx, name, (conv_mfma<5, 1, 16, false, false, true, 5>), grid, block, 0, st, a);
    else return kpb_fail(ctx, KPB_E_INVALID, "conv_mfma: no instance for ks=%d stride=%d cc=%d pool_in=%d pool_out=%d xf=%d", L.ks, S, CC, pool_in, pool_out, x);
    return KPB_OK;
}

int launch_valu(kpb_ctx* ctx, const char* name, kpb_net* net, const Layer& L, const float* in, float* out, int B, int Hi, int Wi,
                bool relu_, const float* xf = nullptr, const float* gray = nullptr, const float* skw = nullptr, const float* skb = nullptr, int skn = 0)
{
    ConvV a;
    a.gray = gray; a.skw = skw; a.skb = skb; a.skn = skn;
    a.in = in; a.out = out; a.w = net->wp((L.name + ".w").c_str()); a.bias = net->wp((L.name + ".b").c_str()); a.xf = xf;
    a.Hi = Hi; a.Wi = Wi; a.KS = L.ks; a.S = L.stride; a.PAD = L.ks / 2;
    a.H = (Hi + 2 * a.PAD - L.ks) / L.stride + 1; a.W = (Wi + 2 * a.PAD - L.ks) / L.stride + 1;
    a.CIN = L.cin; a.COUT = L.cout; a.COUT8 = ((L.cout + 7) / 8) * 8; a.relu = relu_ ? 1 : 0;
    const dim3 grid(cdiv(a.H * a.W, 256), a.COUT8 / 8, B), block(256);
    hipStream_t st = ctx->stream;
    const bool t3 = L.ks == 3 && !xf;
    if (t3 && L.cin == 1 && L.stride == 1 && L.cout == 4)      // XFeat block1.0: four output channels per thread, one 16-byte store (the 8-wide instance computed four padding channels and stored dword by dword)
        KPB_LAUNCH(ctx, name, (conv_valu_t<3, 1, 1, 4>), dim3(grid.x, 1, grid.z), block, 0, st, a);
    else if (t3 && L.cin == 1 && L.stride == 1) KPB_LAUNCH(ctx, name, (conv_valu_t<3, 1, 1>), grid, block, 0, st, a);
    else if (t3 && L.cin == 4 && L.stride == 2) KPB_LAUNCH(ctx, name, (conv_valu_t<3, 4, 2>), grid, block, 0, st, a);
    else if (t3 && L.cin == 8 && L.stride == 1) KPB_LAUNCH(ctx, name, (conv_valu_t<3, 8, 1>), grid, block, 0, st, a);
    else if (t3 && L.cin == 8 && L.stride == 2 && a.COUT8 == 32)
    {
        if (skn == 24) KPB_LAUNCH(ctx, name, (conv_valu_t<3, 8, 2, 32, 24>), dim3(grid.x, 1, grid.z), block, 0, st, a);      // XFeat block1.3: channels 24..31 are padding
        else KPB_LAUNCH(ctx, name, (conv_valu_t<3, 8, 2, 32>), dim3(grid.x, 1, grid.z), block, 0, st, a);
    }
    else if (t3 && L.cin == 8 && L.stride == 2) KPB_LAUNCH(ctx, name, (conv_valu_t<3, 8, 2>), grid, block, 0, st, a);
    else {
        if (gray) return kpb_fail(ctx, KPB_E_INVALID, "conv_valu: the fused skip connection needs a templated instance");
        KPB_LAUNCH(ctx, name, conv_valu, grid, block, 0, st, a);
    }
    return KPB_OK;
}

void stage_layer(WeightStage& ws, const Layer& L, const float* w, const float* b)
{
    if (L.mfma) {
        if (conv_mfma_use_h16() && L.ntb == 5) {       // cout = 4 x 32 + 1: the last channel goes to the VALU side of conv_mfma_h<XC>
            const int T = L.ks * L.ks, co = L.cout - 1;
            const float sc = weight_scale_h(w, (size_t)co * L.cin * T);
            ws.put(L.name + ".w", pack_mfma_h(w, co, L.cin, L.ks, L.cc, 2, sc));
            ws.wscale[L.name + ".w"] = sc;
            const float xs = weight_scale_h(w + (size_t)co * L.cin * T, (size_t)L.cin * T);
            ws.put(L.name + ".xw", pack_xc_pairs(w + (size_t)co * L.cin * T, L.cin, T, xs));
            ws.wscale[L.name + ".xs"] = xs;
            ws.wscale[L.name + ".xb"] = b ? b[co] : 0.0f;     // a host-side scalar, carried with the scales
        } else if (conv_mfma_use_h16()) {
            const float sc = weight_scale_h(w, (size_t)L.cout * L.cin * L.ks * L.ks);
            ws.put(L.name + ".w", pack_mfma_h(w, L.cout, L.cin, L.ks, L.cc, L.ntb, sc));
            ws.wscale[L.name + ".w"] = sc;
        } else {
            ws.put(L.name + ".w", pack_mfma(w, L.cout, L.cin, L.ks, L.cc, L.ntb));
        }
        ws.put(L.name + ".b", pad_bias(b, L.cout, 32 * L.ntb));
    } else {
        ws.put(L.name + ".w", pack_valu(w, L.cout, L.cin, L.ks));
        ws.put(L.name + ".b", pad_bias(b, L.cout, 8));
    }
}

// ================================================================================================ SuperPoint
struct SuperPointNet : kpb_net {
    std::map<std::string, Layer> L;
    kpb_buf aux;                    // per-image maxima of the grey image (the scale conv1b's generated input is split at)
    ~SuperPointNet() override { if (aux.p) (void)hipFree(aux.p); }
    int forward(const float* img, int batch, int H_, int W_, float* score_out, float* desc_out) override
    {
        if ((H_ % 8) || (W_ % 8)) return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_forward: SuperPoint needs H and W multiples of 8 (got %dx%d)", H_, W_);
        if (!desc_out) return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_forward: SuperPoint writes its 256 x H/8 x W/8 descriptor map; desc_out_dev is required");
        const int H = H_, W = W_, Hc = H / 8, Wc = W / 8;
        const size_t P = (size_t)H * W, B = batch;
        // activations (floats per image)
        const size_t n_gray = P, n_1a = P * 64, n_1b = P / 4 * 64, n_2a = P / 4 * 64, n_2b = P / 16 * 64, n_3a = P / 16 * 128,
                     n_3b = P / 64 * 128, n_4a = P / 64 * 128, n_4b = P / 64 * 128, n_pa = P / 64 * 256, n_semi = P / 64 * 65, n_da = P / 64 * 256;
        const size_t total = B * (n_gray + n_1a + n_1b + n_2a + n_2b + n_3a + n_3b + n_4a + n_4b + n_pa + n_semi + n_da);
        if (int rc = kpb_reserve(ctx, act, total * sizeof(float))) return rc;
        float* p = static_cast<float*>(act.p);
        float* gray = p; p += B * n_gray;
        float* x1a = p; p += B * n_1a;
        float* x1b = p; p += B * n_1b;
        float* x2a = p; p += B * n_2a;
        float* x2b = p; p += B * n_2b;
        float* x3a = p; p += B * n_3a;
        float* x3b = p; p += B * n_3b;
        float* x4a = p; p += B * n_4a;
        float* x4b = p; p += B * n_4b;
        float* cpa = p; p += B * n_pa;
        float* semi = p; p += B * n_semi;
        float* cda = p; p += B * n_da;
        this->B = batch; this->H = H; this->W = W;
        hipStream_t st = ctx->stream;
        KPB_LAUNCH(ctx, "sp_rgb_sum", rgb_sum, dim3((unsigned)((P + 255) / 256), batch), dim3(256), 0, st, img, gray, P);
        int rc;
        // conv1a (:44) is not launched on the split-f16 path: conv1b computes its channels from the gray image while it stages its tile
        // (conv_mfma_h<.., GEN>, r05), split at the scale of the bound amax(gray) l1 + bmax of conv1a's output.  r05's two earlier forms -- conv1a as
        // its own kernel handing over fp32 or pre-split halves, KPB_PRESPLIT=0 / 1 -- were measured, superseded and removed (r06).
        const bool fused = conv_mfma_use_h16();
        PreSplit ps;
        if (fused) {
            if ((rc = kpb_reserve(ctx, aux, (size_t)batch * sizeof(unsigned)))) return rc;
            unsigned* amax_gray = static_cast<unsigned*>(aux.p);
            KPB_LAUNCH(ctx, "sp_gray_amax", plane_abs_max, dim3(batch), dim3(1024), 0, st, gray, P, amax_gray);
            ps.amax = amax_gray; ps.l1 = wscale.at("conv1a.l1"); ps.bmax = wscale.at("conv1a.bmax");
            ps.gen_w = wp("conv1a.w"); ps.gen_b = wp("conv1a.b");
        } else
            KPB_LAUNCH(ctx, "sp_conv1a", conv1a_c64, dim3(cdiv(W, 16), cdiv(H, 32), batch), dim3(256), 0, st, gray, x1a, wp("conv1a.w"), wp("conv1a.b"), H, W, 32);   // :44
        if ((rc = launch_mfma(ctx, "sp_conv1b", this, L["conv1b"], fused ? gray : x1a, x1b, batch, H, W, false, true, true, nullptr, 0, 0.0f, nullptr, fused ? &ps : nullptr))) return rc;      // :45-46 (+pool)
        if ((rc = launch_mfma(ctx, "sp_conv2a", this, L["conv2a"], x1b, x2a, batch, H / 2, W / 2, false, false, true))) return rc;
        if ((rc = launch_mfma(ctx, "sp_conv2b", this, L["conv2b"], x2a, x2b, batch, H / 2, W / 2, false, true, true))) return rc;
        if ((rc = launch_mfma(ctx, "sp_conv3a", this, L["conv3a"], x2b, x3a, batch, H / 4, W / 4, false, false, true))) return rc;
        if ((rc = launch_mfma(ctx, "sp_conv3b", this, L["conv3b"], x3a, x3b, batch, H / 4, W / 4, false, true, true))) return rc;
        if ((rc = launch_mfma(ctx, "sp_conv4a", this, L["conv4a"], x3b, x4a, batch, Hc, Wc, false, false, true))) return rc;
        if ((rc = launch_mfma(ctx, "sp_conv4b", this, L["conv4b"], x4a, x4b, batch, Hc, Wc, false, false, true))) return rc;
        if ((rc = launch_mfma(ctx, "sp_convPa", this, L["convPa"], x4b, cpa, batch, Hc, Wc, false, false, true))) return rc;    // :56
        if ((rc = launch_mfma(ctx, "sp_convPb", this, L["convPb"], cpa, semi, batch, Hc, Wc, false, false, false))) return rc;  // :57
        if ((rc = launch_mfma(ctx, "sp_convDa", this, L["convDa"], x4b, cda, batch, Hc, Wc, false, false, true))) return rc;    // :59
        if ((rc = launch_mfma(ctx, "sp_convDb", this, L["convDb"], cda, desc_out, batch, Hc, Wc, false, false, false))) return rc; // :60
        KPB_LAUNCH(ctx, "sp_l2norm", l2norm_nhwc, dim3((unsigned)((B * Hc * Wc + 3) / 4)), dim3(256), 0, st, desc_out, 256, B * Hc * Wc, 0.0f);
        KPB_LAUNCH(ctx, "sp_softmax_d2s", softmax65_d2s, dim3(cdiv(Hc * Wc, 4 * PXW), batch), dim3(256), 0, st, semi, score_out, Hc, Wc, Hc * Wc);
        KPB_HIP(ctx, hipGetLastError());
        return KPB_OK;
    }
};

}  // namespace

int superpoint_create(kpb_ctx* ctx, const KpbwBlob& bl, kpb_net** out)
{
    const Layer plan[] = {
        {"conv1a", 1, 64, 3, 1, false}, {"conv1b", 64, 64, 3, 1, true}, {"conv2a", 64, 64, 3, 1, true}, {"conv2b", 64, 64, 3, 1, true},
        {"conv3a", 64, 128, 3, 1, true}, {"conv3b", 128, 128, 3, 1, true}, {"conv4a", 128, 128, 3, 1, true}, {"conv4b", 128, 128, 3, 1, true},
        {"convPa", 128, 256, 3, 1, true}, {"convPb", 256, 65, 1, 1, true}, {"convDa", 128, 256, 3, 1, true}, {"convDb", 256, 256, 1, 1, true}};
    SuperPointNet* net = new SuperPointNet();
    net->ctx = ctx; net->arch = KPB_ARCH_SUPERPOINT; net->dim = 256; net->desc_div = 8;
    WeightStage ws;
    for (const Layer& L : plan) {
        const float* w = bl.get((L.name + ".weight").c_str(), {(uint32_t)L.cout, (uint32_t)L.cin, (uint32_t)L.ks, (uint32_t)L.ks});
        const float* b = bl.get((L.name + ".bias").c_str(), {(uint32_t)L.cout});
        if (!w || !b) {
            delete net;
            return kpb_fail(ctx, KPB_E_WEIGHTS, "kpb_net_create: SuperPoint tensor %s.weight/.bias missing or mis-shaped", L.name.c_str());
        }
        stage_layer(ws, L, w, b);
        net->L[L.name] = L;
        if (L.name == "conv1a") {       // |conv1a output| <= amax(gray) l1 + bmax (its channels' largest L1 norm, its largest |bias|)
            float l1 = 0.0f, bmax = 0.0f;
            for (int co = 0; co < L.cout; ++co) {
                float r = 0.0f;
                for (int k = 0; k < L.cin * 9; ++k) r += std::fabs(w[(size_t)co * L.cin * 9 + k]);
                l1 = std::max(l1, r);
                bmax = std::max(bmax, std::fabs(b[co]));
            }
            ws.wscale["conv1a.l1"] = l1 * 1.0001f;      // the bound is taken in fp32: a hair of slack for its own rounding
            ws.wscale["conv1a.bmax"] = bmax;
        }
    }
    if (int rc = ws.upload(net)) { delete net; return rc; }
    *out = net;
    return KPB_OK;
}

namespace {

struct XFeatPlan { const char* name; int cin, cout, ks, stride; bool relu; };
// BasicLayer = conv(no bias) + BatchNorm(affine=False) + ReLU, folded at pack time (weights.py fold_xfeat)
const XFeatPlan XF[] = {
    {"block1.0", 1, 4, 3, 1, true}, {"block1.1", 4, 8, 3, 2, true}, {"block1.2", 8, 8, 3, 1, true}, {"block1.3", 8, 24, 3, 2, true},
    {"block2.0", 24, 24, 3, 1, true}, {"block2.1", 24, 24, 3, 1, true},
    {"block3.0", 24, 64, 3, 2, true}, {"block3.1", 64, 64, 3, 1, true}, {"block3.2", 64, 64, 1, 1, true},
    {"block4.0", 64, 64, 3, 2, true}, {"block4.1", 64, 64, 3, 1, true}, {"block4.2", 64, 64, 3, 1, true},
    {"block5.0", 64, 128, 3, 2, true}, {"block5.1", 128, 128, 3, 1, true}, {"block5.2", 128, 128, 3, 1, true}, {"block5.3", 128, 64, 1, 1, true},
    {"block_fusion.0", 64, 64, 3, 1, true}, {"block_fusion.1", 64, 64, 3, 1, true}, {"block_fusion.2", 64, 64, 1, 1, false},
    {"keypoint_head.0", 64, 64, 1, 1, true}, {"keypoint_head.1", 64, 64, 1, 1, true}, {"keypoint_head.2", 64, 64, 1, 1, true},
    {"keypoint_head.3", 64, 65, 1, 1, false}};

struct XFeatNet : kpb_net {
    std::map<std::string, Layer> L;
    std::map<std::string, bool> relu_of;
    int conv(const char* n, const float* in, float* out, int batch, int Hi, int Wi)
    {
        const Layer& l = L.at(n);
        const std::string tag = std::string("xf_") + n;
        if (l.mfma) return launch_mfma(ctx, tag.c_str(), this, l, in, out, batch, Hi, Wi, false, false, relu_of.at(n));
        return launch_valu(ctx, tag.c_str(), this, l, in, out, batch, Hi, Wi, relu_of.at(n));
    }
    int forward(const float* img, int batch, int H_, int W_, float* score_out, float* desc_out) override
    {
        if ((H_ % 32) || (W_ % 32)) return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_forward: XFeat needs H and W multiples of 32 (got %dx%d)", H_, W_);
        if (!desc_out) return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_forward: XFeat writes its 64 x H/8 x W/8 feature map; desc_out_dev is required");
        const int H = H_, W = W_;
        const size_t P = (size_t)H * W, B = batch;
        const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8, H16 = H / 16, W16 = W / 16, H32 = H / 32, W32 = W / 32;
        // floats per image of each activation
        const size_t n_gray = P, n_a = P * 4, n_b = P / 4 * 8, n_c = P / 4 * 8, n_x1 = P / 16 * 32, n_t = P / 16 * 32, n_x2 = P / 16 * 32,
                     n_8 = P / 64 * 64, n_16 = P / 256 * 64, n_32a = P / 1024 * 128, n_semi = P / 64 * 65;
        const size_t total = B * (n_gray + n_a + n_b + n_c + n_x1 + n_t + n_x2 + 6 * n_8 + 3 * n_16 + 3 * n_32a + n_semi) + 64;
        if (int rc = kpb_reserve(ctx, act, total * sizeof(float) + 16 * XF_STAT_BLOCKS * B + 8 * B)) return rc;
        float* p = static_cast<float*>(act.p);
        double* stats = reinterpret_cast<double*>(p); p += 4 * XF_STAT_BLOCKS * B;          // [B][XF_STAT_BLOCKS] pairs of doubles
        float* mrbuf = p; p += 2 * B;                                                      // (mean, 1 / std) per image
        float* gray = p; p += B * n_gray;
        float* a1 = p; p += B * n_a; float* b1 = p; p += B * n_b; float* c1 = p; p += B * n_c;
        float* x1 = p; p += B * n_x1; float* t2 = p; p += B * n_t; float* x2 = p; p += B * n_x2;
        float* u8[6]; for (auto& q : u8) { q = p; p += B * n_8; }
        float* u16[3]; for (auto& q : u16) { q = p; p += B * n_16; }
        float* u32[3]; for (auto& q : u32) { q = p; p += B * n_32a; }
        float* semi = p; p += B * n_semi;
        this->B = batch; this->H = H; this->W = W;
        hipStream_t st = ctx->stream;
        KPB_LAUNCH(ctx, "xf_gray_stats", gray_mean_stats, dim3(XF_STAT_BLOCKS, batch), dim3(256), 0, st, img, gray, stats, P);
        // the split-f16 form normalises the grey image where it is read (three consumers) from per-image (mean, 1 / std)
        const bool fold_norm = conv_mfma_use_h16() && L.at("keypoint_head.0").mfma && W % 8 == 0;
        float2* mr = reinterpret_cast<float2*>(mrbuf);
        KPB_LAUNCH(ctx, "xf_instnorm_params", instnorm_params, dim3(cdiv(batch, 256)), dim3(256), 0, st, stats, P, mr, batch);
        if (!fold_norm) {
            KPB_LAUNCH(ctx, "xf_instnorm", instnorm_apply, dim3((unsigned)((P / 4 + 255) / 256), batch), dim3(256), 0, st, gray, mr, P);
            mr = nullptr;
        }
        int rc;
        // block1 (XFeat.py:30-35) and the skip connection (27-28, 127)
        {   // block1.0 + block1.1 fused: the 4-channel full-resolution map never reaches HBM (r04)
            const Layer &l0 = L.at("block1.0"), &l1 = L.at("block1.1");
            if (l0.mfma || l1.mfma || l0.cin != 1 || l0.cout != 4 || l0.ks != 3 || l0.stride != 1 || l1.cin != 4 || l1.cout != 8 || l1.ks != 3 || l1.stride != 2 ||
                !relu_of.at("block1.0") || !relu_of.at("block1.1"))
                return kpb_fail(ctx, KPB_E_INVALID, "XFeat block1.0 / block1.1: unexpected layer plan");
            KPB_LAUNCH(ctx, "xf_block1.01", xfeat_block1_01, dim3(cdiv(W2, XB_TW), cdiv(H2, XB_TH), batch), dim3(256), 0, st, gray, b1,
                       wp("block1.0.w"), wp("block1.0.b"), wp("block1.1.w"), wp("block1.1.b"), H, W, mr);
            (void)a1;
        }
        if (conv_mfma_use_h16()) {      // block1.2 + block1.3 + skip on the matrix cores, the map between them in LDS (xfeat_block1_23)
            XfB1Args xa{b1, x1, reinterpret_cast<const uint4*>(wp("block1.23.wA")), reinterpret_cast<const uint4*>(wp("block1.23.wB")),
                        wp("block1.23.bA"), wp("block1.23.bB"), gray, wp("skip1.w"), wp("skip1.b"), mr, H2, W2,
                        wscale.at("block1.23.inv_wsA"), wscale.at("block1.23.inv_wsB"), wscale.at("block1.23.l1A"), wscale.at("block1.23.bmaxA")};
            KPB_LAUNCH(ctx, "xf_block1.23", xfeat_block1_23, dim3(cdiv(W4, XQ_TW), cdiv(H4, XQ_TH), batch), dim3(256), 0, st, xa);
            (void)c1;
        } else {
        if ((rc = conv("block1.2", b1, c1, batch, H2, W2))) return rc;
        // block1's last layer with the skip connection (AvgPool2d(4) -> Conv2d(1, 24, 1), XFeat.py:27-28, 127) added in its epilogue:
        // as a kernel of its own (r02: xf_skip_add, 1.12 ms per 512 images) it read x1 back and wrote it again
        {
            const Layer& l = L.at("block1.3");
            if (l.mfma || l.ks != 3 || l.cin != 8 || l.stride != 2 || ((l.cout + 7) / 8) * 8 != 32)
                return kpb_fail(ctx, KPB_E_INVALID, "XFeat block1.3: unexpected layer plan");
            if ((rc = launch_valu(ctx, "xf_block1.3", this, l, c1, x1, batch, H2, W2, relu_of.at("block1.3"), nullptr, gray, wp("skip1.w"), wp("skip1.b"), 24))) return rc;
        }
        }
        if ((rc = conv("block2.0", x1, t2, batch, H4, W4))) return rc;
        if ((rc = conv("block2.1", t2, x2, batch, H4, W4))) return rc;
        if ((rc = conv("block3.0", x2, u8[0], batch, H4, W4))) return rc;
        if ((rc = conv("block3.1", u8[0], u8[1], batch, H8, W8))) return rc;
        if ((rc = conv("block3.2", u8[1], u8[2], batch, H8, W8))) return rc;          // x3
        if ((rc = conv("block4.0", u8[2], u16[0], batch, H8, W8))) return rc;
        if ((rc = conv("block4.1", u16[0], u16[1], batch, H16, W16))) return rc;
        if ((rc = conv("block4.2", u16[1], u16[2], batch, H16, W16))) return rc;       // x4
        if ((rc = conv("block5.0", u16[2], u32[0], batch, H16, W16))) return rc;
        if ((rc = conv("block5.1", u32[0], u32[1], batch, H32, W32))) return rc;
        if ((rc = conv("block5.2", u32[1], u32[2], batch, H32, W32))) return rc;
        float* x5 = u32[0];                                                             // 64 channels reuse the 128-channel slot
        if ((rc = conv("block5.3", u32[2], x5, batch, H32, W32))) return rc;
        KPB_LAUNCH(ctx, "xf_pyramid_sum", pyramid_sum, dim3(cdiv(H8 * W8 * 16, 256), batch), dim3(256), 0, st, u8[2], u16[2], x5, u8[3],
                   H8, W8, H16, W16, H32, W32);
        if ((rc = conv("block_fusion.0", u8[3], u8[4], batch, H8, W8))) return rc;
        if ((rc = conv("block_fusion.1", u8[4], u8[5], batch, H8, W8))) return rc;
        if (conv_mfma_use_h16() && L.at("block_fusion.2").mfma) {     // F.normalize in the product's epilogue (a wave holds whole 64-channel rows)
            if ((rc = launch_mfma(ctx, "xf_block_fusion.2", this, L.at("block_fusion.2"), u8[5], desc_out, batch, H8, W8, false, false, relu_of.at("block_fusion.2"), nullptr, 0, 1e-12f))) return rc;
        } else {
        if ((rc = conv("block_fusion.2", u8[5], desc_out, batch, H8, W8))) return rc;
        KPB_LAUNCH(ctx, "xf_l2norm", l2norm_nhwc, dim3((unsigned)((B * H8 * W8 + 4 * PXW - 1) / (4 * PXW))), dim3(256), 0, st, desc_out, 64, B * H8 * W8, 1e-12f);   // F.normalize
        }
        // keypoint head on the 8x8-unfolded normalised image (XFeat.py:138-139)
        if (conv_mfma_use_h16() && L.at("keypoint_head.0").mfma && W % 8 == 0) {
            // the first layer reads the 8 x 8 cells straight from the normalised image (ConvM::unfold_w): as a kernel of its own the
            // unfolding wrote and re-read 0.63 GB per 512 images (xf_unfold8, 0.24 ms)
            if ((rc = launch_mfma(ctx, "xf_keypoint_head.0", this, L.at("keypoint_head.0"), gray, u8[1], batch, H8, W8, false, false, relu_of.at("keypoint_head.0"), nullptr, W, 0.0f, mr))) return rc;
        } else {
            KPB_LAUNCH(ctx, "xf_unfold8", unfold8, dim3(cdiv(H8 * W8 * 16, 256), batch), dim3(256), 0, st, gray, u8[0], H, W);
            if ((rc = conv("keypoint_head.0", u8[0], u8[1], batch, H8, W8))) return rc;
        }
        if ((rc = conv("keypoint_head.1", u8[1], u8[0], batch, H8, W8))) return rc;
        if ((rc = conv("keypoint_head.2", u8[0], u8[1], batch, H8, W8))) return rc;
        if ((rc = conv("keypoint_head.3", u8[1], semi, batch, H8, W8))) return rc;
        KPB_LAUNCH(ctx, "xf_softmax_d2s", softmax65_d2s, dim3(cdiv(H8 * W8, 4 * PXW), batch), dim3(256), 0, st, semi, score_out, H8, W8, H8 * W8);
        KPB_HIP(ctx, hipGetLastError());
        return KPB_OK;
    }
};

}  // namespace

int xfeat_create(kpb_ctx* ctx, const KpbwBlob& bl, kpb_net** out)
{
    XFeatNet* net = new XFeatNet();
    net->ctx = ctx; net->arch = KPB_ARCH_XFEAT; net->dim = 64; net->desc_div = 8;
    WeightStage ws;
    for (const XFeatPlan& q : XF) {
        const float* w = bl.get((std::string(q.name) + ".w").c_str(), {(uint32_t)q.cout, (uint32_t)q.cin, (uint32_t)q.ks, (uint32_t)q.ks});
        const float* b = bl.get((std::string(q.name) + ".b").c_str(), {(uint32_t)q.cout});
        if (!w || !b) {
            delete net;
            return kpb_fail(ctx, KPB_E_WEIGHTS, "kpb_net_create: XFeat tensor %s.w/.b missing or mis-shaped", q.name);
        }
        // the 24-channel stage (block1.3 out, block2, block3.0 in) is stored with 32 channels, the extra 8 are
        // exact zeros (zero weights, zero bias, ReLU), which puts block2 and block3.0 on the MFMA kernel
        const int cin = q.cin == 24 ? 32 : q.cin, cout = q.cout == 24 ? 32 : q.cout, T = q.ks * q.ks;
        std::vector<float> wpad((size_t)cout * cin * T, 0.0f), bpad(cout, 0.0f);
        for (int o = 0; o < q.cout; ++o) {
            bpad[o] = b[o];
            for (int c = 0; c < q.cin; ++c)
                for (int t = 0; t < T; ++t) wpad[((size_t)o * cin + c) * T + t] = w[((size_t)o * q.cin + c) * T + t];
        }
        w = wpad.data(); b = bpad.data();
        Layer L{q.name, cin, cout, q.ks, q.stride, (cin % 32 == 0)};
        L.cc = q.stride == 2 ? 16 : 32;
        if (cout <= 32 && q.ks == 3 && q.stride == 1) L.ntb = 1;      // block2: one 32-wide output tile, not a half-empty pair
        stage_layer(ws, L, w, b);
        net->L[L.name] = L;
        net->relu_of[L.name] = q.relu;
    }
    if (conv_mfma_use_h16()) {      // the fused matrix form of block1.2 + block1.3 (xfeat_block1_23)
        const float* wA = bl.get("block1.2.w", {8, 8, 3, 3});
        const float* bA = bl.get("block1.2.b", {8});
        const float* wB = bl.get("block1.3.w", {24, 8, 3, 3});
        const float* bB = bl.get("block1.3.b", {24});
        if (!wA || !bA || !wB || !bB || !net->relu_of.at("block1.2") || !net->relu_of.at("block1.3")) {
            delete net;
            return kpb_fail(ctx, KPB_E_WEIGHTS, "kpb_net_create: XFeat block1.2 / block1.3: unexpected layer plan");
        }
        const float scA = weight_scale_h(wA, 8 * 8 * 9), scB = weight_scale_h(wB, 24 * 8 * 9);
        ws.put("block1.23.wA", pack_xf_pairs(wA, scA));
        ws.put("block1.23.wB", pack_xf_s2(wB, 24, scB));
        ws.put("block1.23.bA", pad_bias(bA, 8, 8));
        ws.put("block1.23.bB", pad_bias(bB, 24, 32));
        float l1 = 0.0f, bmax = 0.0f;
        for (int o = 0; o < 8; ++o) {
            float s1 = 0.0f;
            for (int i = 0; i < 8 * 9; ++i) s1 += std::fabs(wA[(size_t)o * 72 + i]);
            l1 = std::max(l1, s1);
            bmax = std::max(bmax, std::fabs(bA[o]));
        }
        ws.wscale["block1.23.inv_wsA"] = 1.0f / scA;
        ws.wscale["block1.23.inv_wsB"] = 1.0f / scB;
        ws.wscale["block1.23.l1A"] = l1 * 1.0001f;       // the bound is taken in fp32: a hair of slack for its own rounding
        ws.wscale["block1.23.bmaxA"] = bmax;
    }
    const float* sw = bl.get("skip1.w", {24});
    const float* sb = bl.get("skip1.b", {24});
    if (!sw || !sb) { delete net; return kpb_fail(ctx, KPB_E_WEIGHTS, "kpb_net_create: XFeat skip1 tensors missing"); }
    ws.put_raw("skip1.w", sw, 24);
    ws.put_raw("skip1.b", sb, 24);
    if (int rc = ws.upload(net)) { delete net; return rc; }
    *out = net;
    return KPB_OK;
}

// ================================================================================================ DISK (N4)
// models/disk.py:293-313: thin U-Net, 5x5 convs with padding, every Conv block but the first is
// InstanceNorm2d -> PReLU -> conv (pre-activation, disk.py:76-97).  The norm + PReLU are applied while the
// conv stages its input tile (ConvM::xf); their per-(image, channel) statistics come from chan_sums.
namespace {

__global__ void avgpool2_nhwc(const float* in, float* out, int H, int W, int C)   // F.avg_pool2d(x, 2), disk.py:69
{
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= (size_t)Ho * Wo * C4) return;
    const int c4 = (int)(i % C4);
    const size_t pix = i / C4;
    const int y = (int)(pix / Wo), x = (int)(pix - (size_t)y * Wo);
    const float* s = in + (((size_t)b * H + 2 * y) * W + 2 * x) * C + 4 * c4;
    const float4 a = *reinterpret_cast<const float4*>(s), bq = *reinterpret_cast<const float4*>(s + C);
    const float4 c = *reinterpret_cast<const float4*>(s + (size_t)W * C), d = *reinterpret_cast<const float4*>(s + (size_t)W * C + C);
    *reinterpret_cast<float4*>(out + (((size_t)b * Ho + y) * Wo + x) * C + 4 * c4) =
        make_float4(((a.x + bq.x) + (c.x + d.x)) * 0.25f, ((a.y + bq.y) + (c.y + d.y)) * 0.25f,
                    ((a.z + bq.z) + (c.z + d.z)) * 0.25f, ((a.w + bq.w) + (c.w + d.w)) * 0.25f);
}

// bilinear x2 (align_corners=False, disk.py:52-55) of `bot` concatenated with `hor` along channels (disk.py:137-139)
__global__ void upsample2_concat(const float* bot, const float* hor, float* out, int Hb, int Wb, int Cb, int Ch)
{
    const int H = 2 * Hb, W = 2 * Wb, C = Cb + Ch, C4 = C / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= (size_t)H * W * C4) return;
    const int c = (int)(i % C4) * 4;
    const size_t pix = i / C4;
    const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W);
    float4 v;
    if (c < Cb) {
        int y0, y1, x0, x1;
        float ly, lx;
        cm_up2_taps(y, Hb, y0, y1, ly);
        cm_up2_taps(x, Wb, x0, x1, lx);
        const float* m = bot + (size_t)b * Hb * Wb * Cb + c;
        const float4 p00 = *reinterpret_cast<const float4*>(m + ((size_t)y0 * Wb + x0) * Cb), p01 = *reinterpret_cast<const float4*>(m + ((size_t)y0 * Wb + x1) * Cb);
        const float4 p10 = *reinterpret_cast<const float4*>(m + ((size_t)y1 * Wb + x0) * Cb), p11 = *reinterpret_cast<const float4*>(m + ((size_t)y1 * Wb + x1) * Cb);
        v = cm_up2_mix(p00, p01, p10, p11, ly, lx);
    } else {
        v = *reinterpret_cast<const float4*>(hor + (((size_t)b * H + y) * W + x) * Ch + (c - Cb));
    }
    *reinterpret_cast<float4*>(out + (((size_t)b * H + y) * W + x) * C + c) = v;
}

// per (image, channel) sum and sum of squares over the pixels of an NHWC tensor; block = (C, R) threads.  Every block leaves ITS partial pair in
// part[image][block][channel] and make_xf adds the blocks up in index order: the sums are the same bits every run (r05: they were accumulated by
// fp64 atomics in arrival order -- within rounding of the float parameters they feed, but not a fixed sequence of operations).
__global__ void chan_sums(const float* in, double* part, size_t P, int C, int Ctot, int coff, float* mm)
{
    extern __shared__ double sh[];   // [R][C][2]
    const int c = threadIdx.x, r = threadIdx.y, R = blockDim.y;
    const size_t b = blockIdx.y;
    double s = 0.0, q = 0.0;
    float mn = INFINITY, mx = -INFINITY;        // mm (r06): the channel's smallest and largest value, for the bound conv_mfma_h<UP> scales its slabs by
    const size_t step = (size_t)gridDim.x * R;
    size_t pix = (size_t)blockIdx.x * R + r;
    for (; pix + 3 * step < P; pix += 4 * step) {        // four loads in flight, accumulated in the order of the plain loop
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = in[(b * P + pix + k * step) * C + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) { s += (double)v[k]; q += (double)v[k] * (double)v[k]; mn = fminf(mn, v[k]); mx = fmaxf(mx, v[k]); }
    }
    for (; pix < P; pix += step) {
        const float v = in[(b * P + pix) * C + c];
        s += (double)v; q += (double)v * (double)v; mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    sh[(r * C + c) * 2] = s; sh[(r * C + c) * 2 + 1] = q;
    __syncthreads();
    if (r == 0) {
        for (int k = 1; k < R; ++k) { s += sh[(k * C + c) * 2]; q += sh[(k * C + c) * 2 + 1]; }
        double* o = part + ((b * gridDim.x + blockIdx.x) * Ctot + coff + c) * 2;     // a row of Ctot channel pairs per block: this tensor's channels start at coff
        o[0] = s; o[1] = q;
    }
    if (mm) {       // kernel-uniform
        __syncthreads();
        sh[(r * C + c) * 2] = (double)mn; sh[(r * C + c) * 2 + 1] = (double)mx;
        __syncthreads();
        if (r == 0) {
            for (int k = 1; k < R; ++k) { mn = fminf(mn, (float)sh[(k * C + c) * 2]); mx = fmaxf(mx, (float)sh[(k * C + c) * 2 + 1]); }
            float* o = mm + ((b * gridDim.x + blockIdx.x) * Ctot + coff + c) * 2;
            o[0] = mn; o[1] = mx;
        }
    }
}

// The same partial sums for a tensor that is never materialised (r06): the 2 x bilinear upsampling of `bot` [B][Hb][Wb][Cb] that conv_mfma_h<UP> evaluates
// while it stages.  A thread owns four channels and walks source cells (i, j): the cell's 3 x 3 neighbourhood (edge-replicated: at the frame the interpolation's
// clamped taps coincide) gives its four output pixels (2 i + dy, 2 j + dx) separably -- 1/4 : 3/4 mixes down the columns, then along the rows.  These are sums of
// fp64 over 1.2 M values for a mean and a variance that are rounded to float: they need the map's values to a few ulp, not to the bit (the staging evaluates
// torch's own expression, cm_up2_mix).  Fixed order: the same bits every run.  blockDim = (Cb / 4, 256 / (Cb / 4)).
__global__ __launch_bounds__(256) void up_chan_sums(const float* __restrict__ bot, double* __restrict__ part, int Hb, int Wb, int Cb, int Ctot, int coff, float* __restrict__ mm)
{
    extern __shared__ double sh[];   // [R][Cb / 4][8]
    const int cq = threadIdx.x, r = threadIdx.y, R = blockDim.y, CQ = blockDim.x;
    const size_t b = blockIdx.y;
    const float* m = bot + b * (size_t)Hb * Wb * Cb + 4 * cq;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    float4 mn = make_float4(INFINITY, INFINITY, INFINITY, INFINITY), mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    // (the upsampled values are convex combinations of source values: the source's range bounds theirs; every source pixel is the centre of one cell)
    const int ncell = Hb * Wb;
    auto mix = [](const float4 a, const float4 c) { return make_float4(0.25f * a.x + 0.75f * c.x, 0.25f * a.y + 0.75f * c.y, 0.25f * a.z + 0.75f * c.z, 0.25f * a.w + 0.75f * c.w); };
    for (int cell = blockIdx.x * R + r; cell < ncell; cell += gridDim.x * R) {
        const int i = cell / Wb, j = cell - i * Wb;
        const int ym = max(i - 1, 0), yp = min(i + 1, Hb - 1), xm = max(j - 1, 0), xp = min(j + 1, Wb - 1);
        float4 t0[3], t1[3];        // the two output rows of the cell at source columns j - 1, j, j + 1
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = dx == 0 ? xm : (dx == 1 ? j : xp);
            const float4 u0 = *reinterpret_cast<const float4*>(m + ((size_t)ym * Wb + xx) * Cb);
            const float4 u1 = *reinterpret_cast<const float4*>(m + ((size_t)i * Wb + xx) * Cb);
            const float4 u2 = *reinterpret_cast<const float4*>(m + ((size_t)yp * Wb + xx) * Cb);
            t0[dx] = mix(u0, u1);
            t1[dx] = mix(u2, u1);
            if (dx == 1) {
                mn = make_float4(fminf(mn.x, u1.x), fminf(mn.y, u1.y), fminf(mn.z, u1.z), fminf(mn.w, u1.w));
                mx = make_float4(fmaxf(mx.x, u1.x), fmaxf(mx.y, u1.y), fmaxf(mx.z, u1.z), fmaxf(mx.w, u1.w));
            }
        }
        const float4 o[4] = {mix(t0[0], t0[1]), mix(t0[2], t0[1]), mix(t1[0], t1[1]), mix(t1[2], t1[1])};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s[0] += (double)o[k].x; q[0] += (double)o[k].x * (double)o[k].x;
            s[1] += (double)o[k].y; q[1] += (double)o[k].y * (double)o[k].y;
            s[2] += (double)o[k].z; q[2] += (double)o[k].z * (double)o[k].z;
            s[3] += (double)o[k].w; q[3] += (double)o[k].w * (double)o[k].w;
        }
    }
    double* me = sh + ((size_t)r * CQ + cq) * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) { me[2 * k] = s[k]; me[2 * k + 1] = q[k]; }
    __syncthreads();
    if (r == 0) {
        for (int k = 1; k < R; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += sh[((size_t)k * CQ + cq) * 8 + 2 * e]; q[e] += sh[((size_t)k * CQ + cq) * 8 + 2 * e + 1]; }
        double* o = part + ((b * gridDim.x + blockIdx.x) * Ctot + coff + 4 * cq) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[2 * e] = s[e]; o[2 * e + 1] = q[e]; }
    }
    __syncthreads();
    me[0] = (double)mn.x; me[1] = (double)mx.x; me[2] = (double)mn.y; me[3] = (double)mx.y; me[4] = (double)mn.z; me[5] = (double)mx.z; me[6] = (double)mn.w; me[7] = (double)mx.w;
    __syncthreads();
    if (r == 0) {
        float lo[4] = {mn.x, mn.y, mn.z, mn.w}, hi[4] = {mx.x, mx.y, mx.z, mx.w};
        for (int k = 1; k < R; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = fminf(lo[e], (float)sh[((size_t)k * CQ + cq) * 8 + 2 * e]); hi[e] = fmaxf(hi[e], (float)sh[((size_t)k * CQ + cq) * 8 + 2 * e + 1]); }
        float* o = mm + ((b * gridDim.x + blockIdx.x) * Ctot + coff + 4 * cq) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[2 * e] = lo[e]; o[2 * e + 1] = hi[e]; }
    }
}

// InstanceNorm2d (biased variance, eps 1e-5, no affine) folded with the PReLU slope into the conv's input transform
__global__ void make_xf(const double* part, int nblk, const float* slope, float* xf, size_t P, int C, int n, const float* mm)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = i / C, c = i - b * C;
    double s = 0.0, q = 0.0;
    for (int k0 = 0; k0 < nblk; k0 += 16) {          // block order: the same sum every run; sixteen partial pairs requested before the first is added
        double2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = k0 + k < nblk ? *reinterpret_cast<const double2*>(part + (((size_t)b * nblk + k0 + k) * C + c) * 2) : make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < 16; ++k) { s += v[k].x; q += v[k].y; }
    }
    const double mean = s / (double)P;
    const double var = fmax(q / (double)P - mean * mean, 0.0);
    const float rstd = 1.0f / sqrtf((float)var + 1e-5f);
    const float shift = -(float)mean * rstd, sl = slope[i % C];
    // mm (r06): the channel's range over the image -> the largest magnitude its transformed values can take.  v -> prelu(v rstd + shift) is |.|-convex
    // (V-shaped around the zero of the affine part), so the maximum over [lo, hi] sits at an end; conv_mfma_h<UP> scales a slab by the largest of its channels'
    // bounds instead of measuring the staged tile.
    float bound = 0.0f;
    if (mm) {
        float lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < nblk; ++k) { const float* p = mm + (((size_t)b * nblk + k) * C + c) * 2; lo = fminf(lo, p[0]); hi = fmaxf(hi, p[1]); }
        float a0 = fmaf(lo, rstd, shift), a1 = fmaf(hi, rstd, shift);
        a0 = a0 >= 0.0f ? a0 : a0 * sl; a1 = a1 >= 0.0f ? a1 : a1 * sl;
        bound = fmaxf(fabsf(a0), fabsf(a1));
    }
    xf[4 * i] = rstd; xf[4 * i + 1] = shift; xf[4 * i + 2] = sl; xf[4 * i + 3] = bound;
}

// disk.py:311-312: desc = F.normalize(feature[:, :128], dim=1), score = sigmoid(feature[:, 128]); feature is [.., 129]
__global__ __launch_bounds__(256) void disk_head(const float* __restrict__ feat, float* __restrict__ desc, float* __restrict__ score, size_t npix)
{
    // one wave per pixel, DHW pixels per wave with their loads in flight together (see softmax65_d2s); per pixel unchanged
    constexpr int DHW = 4;
    const int lane = threadIdx.x & 63;
    const size_t pix0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * DHW;
    if (pix0 >= npix) return;
    float a[DHW], b[DHW], sl[DHW];
#pragma unroll
    for (int i = 0; i < DHW; ++i) {
        const float* f = feat + min(pix0 + i, npix - 1) * 129;
        a[i] = f[lane]; b[i] = f[lane + 64]; sl[i] = f[128];
    }
#pragma unroll
    for (int i = 0; i < DHW; ++i) {
        const size_t pix = pix0 + i;
        float ss = fmaf(a[i], a[i], b[i] * b[i]);
        ss = kpb_wave_sum(ss);
        const float n = fmaxf(sqrtf(ss), 1e-12f);
        if (pix < npix) {
            desc[pix * 128 + lane] = __fdiv_rn(a[i], n);
            desc[pix * 128 + lane + 64] = __fdiv_rn(b[i], n);
            if (lane == 0) score[pix] = __fdiv_rn(1.0f, 1.0f + expf(-sl[i]));
        }
    }
}

struct DiskNet : kpb_net {
    std::map<std::string, Layer> L;
    static constexpr int SUM_BLOCKS = 128;
    int stats_xf(const float* t, size_t P, int C, const char* slope_name, double* sums, float* xf, int batch)
    {
        hipStream_t st = ctx->stream;
        const int R = 512 / C > 0 ? 512 / C : 1;
        KPB_LAUNCH(ctx, "disk_chan_sums", chan_sums, dim3(SUM_BLOCKS, batch), dim3(C, R), (size_t)R * C * 2 * sizeof(double), st, t, sums, P, C, C, 0, nullptr);
        KPB_LAUNCH(ctx, "disk_make_xf", make_xf, dim3(cdiv(batch * C, 256)), dim3(256), 0, st, sums, SUM_BLOCKS, wp(slope_name), xf, P, C, batch * C, nullptr);
        return KPB_OK;
    }
    // the statistics of [up2(bot) | hor] (Cb + Ch channels at 2 Hb x 2 Wb) without the map: the upsampled channels from the half-resolution source
    // (up_chan_sums), the horizontal ones from their own tensor, into one row of partial pairs per block (r06)
    int stats_xf_up(const float* bot, const float* hor, int Hb, int Wb, int Cb, int Ch, const char* slope_name, double* sums, float* mm, float* xf, int batch)
    {
        hipStream_t st = ctx->stream;
        const int C = Cb + Ch, CQ = Cb / 4, RU = 256 / CQ, R = 512 / Ch > 0 ? 512 / Ch : 1;
        const size_t P = (size_t)4 * Hb * Wb;
        KPB_LAUNCH(ctx, "disk_up_chan_sums", up_chan_sums, dim3(SUM_BLOCKS, batch), dim3(CQ, RU), (size_t)RU * CQ * 8 * sizeof(double), st, bot, sums, Hb, Wb, Cb, C, 0, mm);
        KPB_LAUNCH(ctx, "disk_chan_sums", chan_sums, dim3(SUM_BLOCKS, batch), dim3(Ch, R), (size_t)R * Ch * 2 * sizeof(double), st, hor, sums, P, Ch, C, Cb, mm);
        KPB_LAUNCH(ctx, "disk_make_xf", make_xf, dim3(cdiv(batch * C, 256)), dim3(256), 0, st, sums, SUM_BLOCKS, wp(slope_name), xf, P, C, batch * C, mm);
        return KPB_OK;
    }
    int forward(const float* img, int batch, int H_, int W_, float* score_out, float* desc_out) override
    {
        if ((H_ % 16) || (W_ % 16)) return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_forward: DISK needs H and W multiples of 16 (got %dx%d)", H_, W_);
        if (!desc_out) return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_forward: DISK writes its 128 x H x W descriptor map; desc_out_dev is required");
        const int H = H_, W = W_;
        const size_t P = (size_t)H * W, B = batch;
        // floats per image
        const size_t n_in = P * 4, n_f1 = P * 16, n_p1 = P / 4 * 16, n_f2 = P / 4 * 32, n_p2 = P / 16 * 32, n_f3 = P / 16 * 64, n_p3 = P / 64 * 64,
                     n_f4 = P / 64 * 64, n_p4 = P / 256 * 64, n_f5 = P / 256 * 64, n_c0 = P / 64 * 128, n_u0 = P / 64 * 64, n_c1 = P / 16 * 128,
                     n_u1 = P / 16 * 64, n_c2 = P / 4 * 96, n_u2 = P / 4 * 64, n_c3 = conv_mfma_use_h16() ? 0 : P * 80 /* only the strict-fp32 path materialises [up(u2) | f1] */, n_lg = P * 129;
        const size_t total = B * (n_in + n_f1 + n_p1 + n_f2 + n_p2 + n_f3 + n_p3 + n_f4 + n_p4 + n_f5 + n_c0 + n_u0 + n_c1 + n_u1 + n_c2 + n_u2 + n_c3 + n_lg)
                             + B * SUM_BLOCKS * 128 * 4 + B * SUM_BLOCKS * 128 * 2 + B * 128 * 4 + 64;
        if (int rc = kpb_reserve(ctx, act, total * sizeof(float))) return rc;
        float* p = static_cast<float*>(act.p);
        double* sums = reinterpret_cast<double*>(p); p += B * SUM_BLOCKS * 128 * 4;     // [B][SUM_BLOCKS][<= 128 channels] pairs of doubles: per-block partial sums
        float* mm = p; p += B * SUM_BLOCKS * 128 * 2;                                      // [B][SUM_BLOCKS][<= 128 channels] (min, max): per-block ranges (stats_xf_up)
        float* xf = p; p += B * 128 * 4;
        auto take = [&](size_t n) { float* q = p; p += B * n; return q; };
        float *in4 = take(n_in), *f1 = take(n_f1), *p1 = take(n_p1), *f2 = take(n_f2), *p2 = take(n_p2), *f3 = take(n_f3), *p3 = take(n_p3),
              *f4 = take(n_f4), *p4 = take(n_p4), *f5 = take(n_f5), *c0 = take(n_c0), *u0 = take(n_u0), *c1 = take(n_c1), *u1 = take(n_u1),
              *c2 = take(n_c2), *u2 = take(n_u2), *c3 = take(n_c3), *lg = take(n_lg);
        (void)in4;
        this->B = batch; this->H = H; this->W = W;
        hipStream_t st = ctx->stream;
        int rc;
        auto pool = [&](const float* a, float* o, int h, int w, int c) {
            KPB_LAUNCH(ctx, "disk_avgpool2", avgpool2_nhwc, dim3((unsigned)(((size_t)(h / 2) * (w / 2) * (c / 4) + 255) / 256), batch), dim3(256), 0, st, a, o, h, w, c);
        };
        auto upcat = [&](const float* bot, const float* hor, float* o, int hb, int wb, int cb, int chh) {
            KPB_LAUNCH(ctx, "disk_upsample_concat", upsample2_concat, dim3((unsigned)(((size_t)4 * hb * wb * ((cb + chh) / 4) + 255) / 256), batch), dim3(256), 0, st,
                       bot, hor, o, hb, wb, cb, chh);
        };
        // down path (disk.py:233-250, 274-282).  down_0 has no norm / gate and reads the planar RGB image through a
        // 3-channel NHWC repack done by conv_valu's generic addressing: repack first.
        {
            // [B,3,H,W] -> [B,H,W,3] is avoided: conv_valu reads NHWC, so down_0 uses the planar-input variant below
        }
        if ((rc = launch_valu_planar(img, f1, batch, H, W))) return rc;
        pool(f1, p1, H, W, 16);
        if ((rc = stats_xf(p1, P / 4, 16, "down1.slope", sums, xf, batch))) return rc;
        if ((rc = launch_mfma(ctx, "disk_down1", this, L["down1"], p1, f2, batch, H / 2, W / 2, false, false, false, xf))) return rc;
        pool(f2, p2, H / 2, W / 2, 32);
        if ((rc = stats_xf(p2, P / 16, 32, "down2.slope", sums, xf, batch))) return rc;
        if ((rc = launch_mfma(ctx, "disk_down2", this, L["down2"], p2, f3, batch, H / 4, W / 4, false, false, false, xf))) return rc;
        pool(f3, p3, H / 4, W / 4, 64);
        if ((rc = stats_xf(p3, P / 64, 64, "down3.slope", sums, xf, batch))) return rc;
        if ((rc = launch_mfma(ctx, "disk_down3", this, L["down3"], p3, f4, batch, H / 8, W / 8, false, false, false, xf))) return rc;
        pool(f4, p4, H / 8, W / 8, 64);
        if ((rc = stats_xf(p4, P / 256, 64, "down4.slope", sums, xf, batch))) return rc;
        if ((rc = launch_mfma(ctx, "disk_down4", this, L["down4"], p4, f5, batch, H / 16, W / 16, false, false, false, xf))) return rc;
        // up path (disk.py:114-141, 284-288)
        upcat(f5, f4, c0, H / 16, W / 16, 64, 64);
        if ((rc = stats_xf(c0, P / 64, 128, "up0.slope", sums, xf, batch))) return rc;
        if ((rc = launch_mfma(ctx, "disk_up0", this, L["up0"], c0, u0, batch, H / 8, W / 8, false, false, false, xf))) return rc;
        upcat(u0, f3, c1, H / 8, W / 8, 64, 64);
        if ((rc = stats_xf(c1, P / 16, 128, "up1.slope", sums, xf, batch))) return rc;
        if ((rc = launch_mfma(ctx, "disk_up1", this, L["up1"], c1, u1, batch, H / 4, W / 4, false, false, false, xf))) return rc;
        const bool fuse_up = conv_mfma_use_h16();        // r06: the concatenated decoder inputs of up_2 and up_3 are not written (profiles/r06_disk_fused_upsample_ab.txt)
        if (fuse_up) {
            if ((rc = stats_xf_up(u1, f2, H / 4, W / 4, 64, 32, "up2.slope", sums, mm, xf, batch))) return rc;
            const UpSrc up{u1, 64};
            if ((rc = launch_mfma(ctx, "disk_up2", this, L["up2"], f2, u2, batch, H / 2, W / 2, false, false, false, xf, 0, 0.0f, nullptr, nullptr, &up))) return rc;
        } else {
            upcat(u1, f2, c2, H / 4, W / 4, 64, 32);
            if ((rc = stats_xf(c2, P / 4, 96, "up2.slope", sums, xf, batch))) return rc;
            if ((rc = launch_mfma(ctx, "disk_up2", this, L["up2"], c2, u2, batch, H / 2, W / 2, false, false, false, xf))) return rc;
        }
        if (fuse_up) {      // r06: [up(u2) | f1] is not written -- up_3 evaluates the upsampling while it stages, the statistics come from u2 and f1
            if ((rc = stats_xf_up(u2, f1, H / 2, W / 2, 64, 16, "up3.slope", sums, mm, xf, batch))) return rc;
            const UpSrc up{u2, 64};
            if ((rc = launch_mfma(ctx, "disk_up3", this, L["up3"], f1, lg, batch, H, W, false, false, false, xf, 0, 0.0f, nullptr, nullptr, &up))) return rc;
        } else {
            upcat(u2, f1, c3, H / 2, W / 2, 64, 16);
            if ((rc = stats_xf(c3, P, 80, "up3.slope", sums, xf, batch))) return rc;
            if ((rc = launch_mfma(ctx, "disk_up3", this, L["up3"], c3, lg, batch, H, W, false, false, false, xf))) return rc;
        }
        KPB_LAUNCH(ctx, "disk_head", disk_head, dim3((unsigned)((B * P + 15) / 16)), dim3(256), 0, st, lg, desc_out, score_out, B * P);
        KPB_HIP(ctx, hipGetLastError());
        return KPB_OK;
    }
    int launch_valu_planar(const float* img, float* f1, int batch, int H, int W);
};

// down_0: 3 -> 16, 5x5, padding 2, no norm / gate (disk.py:106-108), reading the planar [B,3,H,W] image
__global__ __launch_bounds__(256) void disk_down0(const float* img, float* out, const float* w /*[75][16]*/, const float* bias, int H, int W)
{
    const int b = blockIdx.z;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int oy = pix / W, ox = pix - oy * W;
    const size_t P = (size_t)H * W;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias[j];
    for (int c = 0; c < 3; ++c)
        for (int ky = 0; ky < 5; ++ky) {
            const int iy = oy + ky - 2;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const int ix = ox + kx - 2;
                const float v = (ix >= 0 && ix < W) ? img[((size_t)b * 3 + c) * P + (size_t)iy * W + ix] : 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = fmaf(v, w[((c * 5 + ky) * 5 + kx) * 16 + j], acc[j]);
            }
        }
    float* o = out + ((size_t)b * P + pix) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(o + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
}

int DiskNet::launch_valu_planar(const float* img, float* f1, int batch, int H, int W)
{
    KPB_LAUNCH(ctx, "disk_down0", disk_down0, dim3(cdiv(H * W, 256), 1, batch), dim3(256), 0, ctx->stream, img, f1, wp("down0.w"), wp("down0.b"), H, W);
    return KPB_OK;
}

}  // namespace

int disk_create(kpb_ctx* ctx, const KpbwBlob& bl, kpb_net** out)
{
    DiskNet* net = new DiskNet();
    net->ctx = ctx; net->arch = KPB_ARCH_DISK; net->dim = 128; net->desc_div = 1;
    WeightStage ws;
    auto fail = [&](const char* n) { delete net; return kpb_fail(ctx, KPB_E_WEIGHTS, "kpb_net_create: DISK tensor %s missing or mis-shaped", n); };
    {   // down_0
        const float* w = bl.get("down0.w", {16, 3, 5, 5});
        const float* b = bl.get("down0.b", {16});
        if (!w || !b) return fail("down0");
        std::vector<float> t(75 * 16);
        for (int o = 0; o < 16; ++o) for (int k = 0; k < 75; ++k) t[k * 16 + o] = w[o * 75 + k];
        ws.put("down0.w", t);
        ws.put_raw("down0.b", b, 16);
    }
    struct { const char* n; int cin, cout; } plan[] = {{"down1", 16, 32}, {"down2", 32, 64}, {"down3", 64, 64}, {"down4", 64, 64},
                                                       {"up0", 128, 64}, {"up1", 128, 64}, {"up2", 96, 64}};
    for (auto& q : plan) {
        Layer L{q.n, q.cin, q.cout, 5, 1, true};
        L.cc = (q.cin % 32 == 0) ? 32 : 16;
        const float* w = bl.get((L.name + ".w").c_str(), {(uint32_t)q.cout, (uint32_t)q.cin, 5, 5});
        const float* b = bl.get((L.name + ".b").c_str(), {(uint32_t)q.cout});
        const float* sl = bl.get((L.name + ".slope").c_str(), {(uint32_t)q.cin});
        if (!w || !b || !sl) return fail(q.n);
        stage_layer(ws, L, w, b);
        ws.put_raw(L.name + ".slope", sl, q.cin);
        net->L[L.name] = L;
    }
    {   // up_3: 80 -> 129 = 128 descriptor channels + the score logit, five 32-wide tiles in one workgroup
        const float* w = bl.get("up3.w", {129, 80, 5, 5});
        const float* b = bl.get("up3.b", {129});
        const float* sl = bl.get("up3.slope", {80});
        if (!w || !b || !sl) return fail("up3");
        Layer L3{"up3", 80, 129, 5, 1, true}; L3.cc = 16; L3.ntb = 5;
        stage_layer(ws, L3, w, b);
        ws.put_raw("up3.slope", sl, 80);
        net->L["up3"] = L3;
    }
    if (int rc = ws.upload(net)) { delete net; return rc; }
    *out = net;
    return KPB_OK;
}
