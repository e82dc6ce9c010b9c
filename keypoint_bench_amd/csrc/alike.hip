// alike.hip -- N1: ALIKE forward (models/ALike.py:136-164 ALNet.forward, ConvBlock 25-28, ResBlock 65-81)
// as hand-written gfx950 kernels.  fp32 throughout (the reference's CPU path is fp32 and the parity
// contract is 1e-4 on descriptors / bit-stable keypoints, so no bf16 shortcuts).
//
// Data layout in HBM: every activation is NHWC fp32 ([B][H][W][C], channel minor) so that one
// pixel's channels are one contiguous 32..256-byte segment: float4 loads in the conv kernels, whole
// 128-byte lines in the head's MFMA epilogue, and a channels-last [B,C,H,W] torch view for the dense
// descriptor map.  The input image stays in the reference's planar [B,3,H,W].
//
//   alike_block1   3->8->8 3x3 (+folded BN, ReLU) fused through LDS; VALU fp32 (K=27/72, N=8 is
//                  too thin for a 32x32 MFMA tile; fp32 MFMA peak equals the VALU peak on gfx950)
//   conv3x3_k      generic pooled-input 3x3 conv (+bias, +1x1 residual branch, ReLU): block2..4
//   conv1x1_relu   the aggregation 1x1 convs 2..4 (+ each group's share of the head rows, projected at low resolution)
//   alike_head_hyb dense mode: head = [relu(agg1 x1) | up2 a2] x W on v_mfma_f32_32x32x2_f32 (A operand straight
//                  from the registers that built the features) + the x interpolation of the projected coarse groups
//                  as extra K steps; score = sigmoid of row 64
//   alike_score_lin score-only mode: the score row in its fully linear form
//   alike_desc_at  descriptors at keypoints only: bilinear taps on f, then one 64x64 mat-vec
#include "conv_mfma.h"

namespace {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));     // one f16 MFMA operand (four VGPRs)
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu(float v);

// fp32 x 4 -> (hi, lo) half-precision pairs: hi = f16(x), lo = f16(x - hi), both to nearest (x - hi is exact in fp32)
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) { cm_split4(v, hi, lo); }

// Largest value of a NON-NEGATIVE tensor, per image, without atomics: every wave of the producing kernel leaves its maximum in a
// slot of its own ([image][workgroup][wave]), and amax_reduce -- one small workgroup per image, launched between producer and
// consumer -- folds an image's slots into one float.  (r03 first tried one atomicMax per wave on a per-image word: the waves of
// an image run together, all see the initial zero, and 1.2 M same-line atomics per launch cost block 1 1.8 ms.)
__device__ __forceinline__ void amax_commit(float lane_max, float* slots, int slot, int lane)
{
    const float m = cm_wave_max(lane_max);
    if (lane == 0) slots[slot] = m;
}

__global__ __launch_bounds__(256) void amax_reduce(const float* slots, int n, unsigned* out)
{
    __shared__ float s[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float m = 0.0f;
    for (int i = tid; i < n; i += 256) {
        // a wave that met an infinite activation leaves +inf: skipped, so that the rest of the image keeps a usable scale and the
        // fault stays where the reference's would (values beyond the window saturate locally)
        const float v = slots[(size_t)b * n + i];
        m = fmaxf(m, v < INFINITY ? v : 0.0f);
    }
    m = cm_wave_max(m);
    if ((tid & 63) == 0) s[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) out[b] = __float_as_uint(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
}

// ------------------------------------------------------------------------------------------------ block1
constexpr int B1_TH = 32, B1_TW = 32;

struct Block1Args {
    const float* img;   // [B][3][H][W]
    float* x1;          // [B][H][W][8]
    float* p1;          // [B][H/2][W/2][8] = max_pool2d(x1, 2, 2) (ALike.py:139), what block 2 reads
    const float* w1;    // [27][8]  (cin, ky, kx) major, cout minor
    const float* b1;    // [8]
    const float* w2;    // [9][8][8] (tap, cin, cout)
    const float* b2;    // [8]
    int H, W;
};

// 32x32 output tile per workgroup.  conv1 is evaluated on the 34x34 halo'd positions as 1x2 strips, conv2 on four
// pixels per thread, so every scalar-loaded weight feeds 2 / 4 FMAs.
//
// LDS layout of the intermediate map (r02): two PLANES of float4, mid_lo[34][34] = channels 0..3 and mid_hi = 4..7, and
// conv2's thread (row r0 = tid / 32, column tid % 32) owns the pixels of rows r0, r0 + 8, r0 + 16, r0 + 24 of its column.
// A wave's ds_read_b128 then finds 32 consecutive lanes on 32 consecutive 16-byte slots of one map row, so each of the
// instruction's 16-lane groups ({0-3, 12-15, 20-27}, ...) lands on 16 distinct 4-bank sets at ANY row pitch.  The r01
// layout ([34][34][8] with 4 consecutive pixels per thread: lanes 128 bytes apart, rows 16 banks apart) put lanes 0/2 and
// 1/3 of every group on the same banks: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 70 %.
__global__ __launch_bounds__(256) void alike_block1(Block1Args a)
{
    constexpr int IH = B1_TH + 4, IW = B1_TW + 4, MH = B1_TH + 2, MW = B1_TW + 2;
    __shared__ float in[3][IH][IW];
    __shared__ __attribute__((aligned(16))) float4 mid[2][MH][MW];          // [lo / hi][row][column]
    const int tid = threadIdx.x, b = blockIdx.z;
    const int ty0 = blockIdx.y * B1_TH, tx0 = blockIdx.x * B1_TW;
    const size_t P = (size_t)a.H * a.W;
    const float* img = a.img + (size_t)b * 3 * P;
    {   // all of a thread's loads are issued before the first LDS store (one memory latency, not sixteen)
        constexpr int N = 3 * IH * IW, PER = (N + 255) / 256;
        float buf[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * 256;
            const int c = i / (IH * IW), rem = i - c * IH * IW;
            const int y = rem / IW, x = rem - y * IW;
            const int gy = ty0 - 2 + y, gx = tx0 - 2 + x;
            buf[k] = 0.0f;
            if (i < N && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) buf[k] = img[c * P + (size_t)gy * a.W + gx];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * 256;
            if (i < N) (&in[0][0][0])[i] = buf[k];
        }
    }
    __syncthreads();
    // conv1 + ReLU on MH x MW positions, two per item
    for (int it = tid; it < MH * (MW / 2); it += 256) {
        const int my = it / (MW / 2), mx = (it - my * (MW / 2)) * 2;
        float acc[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[q][j] = a.b1[j];
#pragma unroll 1
        for (int cky = 0; cky < 9; ++cky) {   // rolled: 24 scalar-loaded weights live per trip, no SGPR spills
                const int c = cky / 3, ky = cky - 3 * c;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = in[c][my + ky][mx + k];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float w = a.w1[((c * 3 + ky) * 3 + kx) * 8 + j];
                        acc[0][j] = fmaf(v[kx], w, acc[0][j]);
                        acc[1][j] = fmaf(v[kx + 1], w, acc[1][j]);
                    }
            }
        const int gy = ty0 - 1 + my;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int gx = tx0 - 1 + mx + q;
            // conv2 pads its INPUT (the ReLU'd map) with zeros outside the image
            const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
            if (inside) {
                lo = make_float4(relu(acc[q][0]), relu(acc[q][1]), relu(acc[q][2]), relu(acc[q][3]));
                hi = make_float4(relu(acc[q][4]), relu(acc[q][5]), relu(acc[q][6]), relu(acc[q][7]));
            }
            mid[0][my][mx + q] = lo;
            mid[1][my][mx + q] = hi;
        }
    }
    __syncthreads();
    // conv2 + ReLU: thread = column tid % 32, rows tid / 32 + 8 q
    const int oy = tid >> 5, ox = tid & 31;
    float acc[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[q][j] = a.b2[j];
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {   // rolled: 64 scalar-loaded weights live per trip; no runtime-indexed registers
        const int ky = tap / 3, kx = tap - 3 * ky;
        float v[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 lo = mid[0][oy + 8 * q + ky][ox + kx];
            const float4 hi = mid[1][oy + 8 * q + ky][ox + kx];
            v[q][0] = lo.x; v[q][1] = lo.y; v[q][2] = lo.z; v[q][3] = lo.w;
            v[q][4] = hi.x; v[q][5] = hi.y; v[q][6] = hi.z; v[q][7] = hi.w;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = a.w2[(tap * 8 + c) * 8 + j];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q][j] = fmaf(v[q][c], w, acc[q][j]);
            }
    }
    // stage the 32x32x8 result through LDS (the mid tile is dead now) so that every wave store
    // instruction writes one whole 1 KiB pixel row segment instead of 64 scattered 16-byte pieces
    __syncthreads();
    float* stage = reinterpret_cast<float*>(&mid[0][0][0]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<float4*>(stage + ((oy + 8 * q) * B1_TW + ox) * 8) = make_float4(relu(acc[q][0]), relu(acc[q][1]), relu(acc[q][2]), relu(acc[q][3]));
        *reinterpret_cast<float4*>(stage + ((oy + 8 * q) * B1_TW + ox) * 8 + 4) = make_float4(relu(acc[q][4]), relu(acc[q][5]), relu(acc[q][6]), relu(acc[q][7]));
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < B1_TH * B1_TW * 8 / 4 / 256; ++k) {
        const int i = tid + k * 256;              // float4 index inside the tile: row = i / 64, 64 float4 per row
        const int ry = i >> 6, rx4 = i & 63;
        const int gy = ty0 + ry, gx = tx0 + (rx4 >> 1);
        if (gy < a.H && gx < a.W)
            *reinterpret_cast<float4*>(a.x1 + ((size_t)b * P + (size_t)gy * a.W + tx0) * 8 + rx4 * 4) =
                *reinterpret_cast<const float4*>(stage + i * 4);
    }
    // the 2x2 max-pool block 2 starts with, from the tile still in LDS: 16x16 pooled pixels x 2 float4
    {
        const int H2 = a.H / 2, W2 = a.W / 2;
#pragma unroll
        for (int k = 0; k < (B1_TH / 2) * (B1_TW / 2) * 2 / 256; ++k) {
            const int i = tid + k * 256;
            const int q = i & 1, px = (i >> 1) & (B1_TW / 2 - 1), py = i >> 5;
            const float* s0 = stage + ((2 * py) * B1_TW + 2 * px) * 8 + 4 * q;
            const float4 v00 = *reinterpret_cast<const float4*>(s0), v01 = *reinterpret_cast<const float4*>(s0 + 8);
            const float4 v10 = *reinterpret_cast<const float4*>(s0 + B1_TW * 8), v11 = *reinterpret_cast<const float4*>(s0 + B1_TW * 8 + 8);
            const int gy = ty0 / 2 + py, gx = tx0 / 2 + px;
            if (gy < H2 && gx < W2)
                *reinterpret_cast<float4*>(a.p1 + (((size_t)b * H2 + gy) * W2 + gx) * 8 + 4 * q) =
                    make_float4(fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x)), fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y)),
                                fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z)), fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w)));
        }
    }
}

// ------------------------------------------------------------------------------------------------ block1 on the matrix cores
// Same fusion as alike_block1 (3 -> 8 -> 8 through LDS), with BOTH convolutions moved from the fp32 vector ALUs to
// v_mfma_f32_16x16x32_f16 on split operands (x = hi + lo halves, three MFMAs per product: alike_block2 has the numerics).
// N = 8 output channels would waste half of a 16-wide MFMA, so one accumulator row stands for a PAIR of horizontally
// adjacent pixels: N = (pixel of the pair s, output channel), and K runs over the 3 x 4 input window the pair shares; the
// weight of window cell (ky, kx) for pixel s is w[ky][kx - s], zero outside the 3 x 3 kernel.
//   conv2 (8 -> 8): 12 pieces of 8 channels = exactly three 32-deep k-blocks (kb = window row, lane group g = window column).
//   conv1 (3 -> 8): the image tile is staged as [position][c0 c1 c2 0] halves (8 bytes), so a piece = two adjacent window
//     cells; 6 pieces = two k-blocks (the second half empty); its accumulators go back to LDS split, two channels per
//     32-bit write after one lane swap, in planes [hi | lo][column parity][row][column / 2] of 16-byte slots -- the layout
//     in which the 16 lanes of conv2's read groups hit 16 consecutive slots.
// Rows go to global memory straight from the accumulators: with pi(4 g + r) = g + 4 r one store instruction covers 8 adjacent
// pixels x 8 channels = 256 B; the 2 x 2 max-pool block 2 starts with (ALike.py:139) is one lane swap (the pair) and one
// register kept per row pair.  16 x 32 tiles: 31 KB of LDS, five workgroups per CU.
struct Block1HArgs {
    Block1Args b;
    const uint4* w1pk;   // [2 kb][hi / lo][64 lanes] fragments of conv1 (pack_b1c1_pairs), scaled by 1 / inv_ws1
    const uint4* w2pk;   // [3 kb][hi / lo][64 lanes] fragments of conv2 (pack_b1c2_pairs), scaled by 1 / inv_ws2
    float inv_ws1, inv_ws2;     // reciprocals of the two power-of-two weight scales
    float l1_c1, bmax_c1;       // max over output channels of sum |w| of conv1, max |bias|: |conv1 output| <= amax(input) l1 + bmax
    float* wmax_x1;             // [B][workgroups][4 waves]: every wave's largest x1 value (x1 >= 0); amax_reduce folds them per image
    int xcd_map;                // kpb_xcd_tile: neighbouring tiles on one XCD
};

constexpr int B1H_TH = 16;

__global__ __launch_bounds__(256) void alike_block1_h(Block1HArgs ha)
{
    // Both products are taken TRANSPOSED, D^T = W^T A^T (the weight fragment as the MFMA's A operand, the input pieces as B:
    // the register layouts of the two operands are the same, so this is only the argument order): the accumulator then has
    // the pixel pair on the LANE (column) and (pixel of the pair, channel) in the four registers of the four lane groups --
    // lane (pair, g) holds channels 4 (g & 1) .. + 3 of pixel g >> 1.  A lane therefore owns one position: one bounds test,
    // one packed split, 8-byte LDS writes for conv1, and for conv2 one 16-byte global store per lane (16 pairs x 64 B = 1 KB
    // contiguous per instruction), with no cross-lane traffic except the pixel-pair maximum of the pooling.
    const Block1Args& a = ha.b;
    constexpr int TH = B1H_TH, IH = TH + 4, IW = B1_TW + 4, MH = TH + 2, MW = B1_TW + 2, HW = MW / 2;     // HW: slots per row and parity
    constexpr int PLANE = MH * HW, REGION = 2 * PLANE, NIN = IH * IW;
    __shared__ __attribute__((aligned(16))) uint2 inh[2 * NIN + 2];        // [hi | lo][position] = (c0 c1 c2 0) halves; + a zero piece
    __shared__ __attribute__((aligned(16))) uint4 mid[2 * REGION];          // [hi | lo][parity][row][column / 2]
    const kpb_tile3 tile = kpb_xcd_tile(ha.xcd_map);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = tile.z;
    const int ty0 = tile.y * TH, tx0 = tile.x * B1_TW;
    const size_t P = (size_t)a.H * a.W;
    const float* img = a.img + (size_t)b * 3 * P;
    h8v c1hi[2], c1lo[2], bhi[3], blo[3];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        c1hi[kb] = __builtin_bit_cast(h8v, ha.w1pk[(kb * 2 + 0) * 64 + lane]);
        c1lo[kb] = __builtin_bit_cast(h8v, ha.w1pk[(kb * 2 + 1) * 64 + lane]);
    }
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
        bhi[kb] = __builtin_bit_cast(h8v, ha.w2pk[(kb * 2 + 0) * 64 + lane]);
        blo[kb] = __builtin_bit_cast(h8v, ha.w2pk[(kb * 2 + 1) * 64 + lane]);
    }
    // the lane's biases are requested with the weights: read where they are used, each was an exposed round trip behind a barrier
    const float4 bias1 = *reinterpret_cast<const float4*>(a.b1 + 4 * ((lane >> 4) & 1));
    const float4 bias2 = *reinterpret_cast<const float4*>(a.b2 + 4 * ((lane >> 4) & 1));
    __shared__ __attribute__((aligned(16))) float s_amax[4];
    int e_img;      // exponent of this tile's largest |pixel|: the image is split at the scale that fits THIS tile (cm_scale_of)
    {   // stage the (TH+4) x 36 image tile split, three channels per position; all loads of a thread in flight first
        constexpr int PER = (NIN + 255) / 256;
        float buf[PER][3];
        float am = 0.0f;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * 256;
            const int y = i / IW, x = i - y * IW;
            const int gy = ty0 - 2 + y, gx = tx0 - 2 + x;
            const bool ok = i < NIN && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
            for (int c = 0; c < 3; ++c) buf[k][c] = ok ? img[c * P + (size_t)gy * a.W + gx] : 0.0f;
            am = kpb_pmax(kpb_pmax(am, fabsf(buf[k][0])), kpb_pmax(fabsf(buf[k][1]), fabsf(buf[k][2])));
        }
        am = cm_wave_max(am);
        if (lane == 0) s_amax[wv] = am;
        __syncthreads();
        float4 q = *reinterpret_cast<const float4*>(s_amax);
        if (!(q.x < INFINITY && q.y < INFINITY && q.z < INFINITY && q.w < INFINITY)) {
            // A NaN / Inf pixel in the tile (kpb_pmax orders bit patterns: a non-finite value wins every integer maximum).  Its
            // exponent would clamp the scale and flush the tile's FINITE pixels to zero; the reference keeps such a fault inside the
            // pixel's receptive field.  Rare path, workgroup-uniform: the scale comes from the finite pixels only.
            float fm = 0.0f;
#pragma unroll
            for (int k = 0; k < PER; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) { const float v = fabsf(buf[k][c]); fm = fmaxf(fm, v < INFINITY ? v : 0.0f); }
            fm = cm_wave_max(fm);
            __syncthreads();
            if (lane == 0) s_amax[wv] = fm;
            __syncthreads();
            q = *reinterpret_cast<const float4*>(s_amax);
        }
        e_img = cm_exp_of(fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
        const float sc = cm_scale_of(e_img);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * 256;
            uint2 hi, lo;
            split4(make_float4(buf[k][0] * sc, buf[k][1] * sc, buf[k][2] * sc, 0.0f), hi, lo);
            if (i < NIN) { inh[i] = hi; inh[NIN + i] = lo; }
        }
        if (tid < 2) inh[2 * NIN + tid] = make_uint2(0u, 0u);
    }
    __syncthreads();
    // conv1's output is bounded by amax(tile) l1 + bmax: the intermediate map is split at the scale of that bound
    const int e_mid = cm_exp_of(fmaf(__uint_as_float((unsigned)(e_img - 126 + 127) << 23), ha.l1_c1, ha.bmax_c1));
    const float un1 = ha.inv_ws1 * cm_unscale_of(e_img), sc_mid = cm_scale_of(e_mid), un2 = ha.inv_ws2 * cm_unscale_of(e_mid);
    const int pr = lane & 15, g = lane >> 4;                         // this lane's pair slot (operand column and accumulator column)
    const int sN = g >> 1, c0 = 4 * (g & 1);                          // ... and what its accumulator holds: pixel sN of the pair, channels c0 .. c0 + 3
    {   // conv1 + ReLU on the MH x MW halo'd positions, 16 pairs per MFMA group: group gi < MH = pairs 0..15 of row gi; the 17th
        // pair of every row goes to two extra groups (rows 0..15 and 16..17)
        const uint4* inq = reinterpret_cast<const uint4*>(inh);       // a piece = two adjacent positions = 16 bytes (even column)
        uint2* midh = reinterpret_cast<uint2*>(mid);
        static_assert((MH + 2) % 4 == 0 && MH % 4 == 2, "alike_block1_h: waves 2 and 3 take the two extra groups");
        // r05: the MH regular groups walk down the tile with everything that depends on the lane alone taken out of the loop -- the two piece
        // addresses (+ IW / 2 slots per row; the empty pieces 6, 7 of lane groups 2, 3 stay on the zero piece), the slot written, the column test,
        // and the intermediate scale folded into the unscale factor and the bias (a power of two: the same bits).  54 -> 30 vector
        // instructions per group of six MFMAs; the general form below is kept for the two extra groups.
        {
            const int gxl = tx0 - 1 + 2 * pr + sN;
            const bool xin = gxl >= 0 && gxl < a.W;
            const float unm = xin ? un1 * sc_mid : 0.0f;               // a column outside the image: relu(0 acc + 0) = 0, conv2's zero padding
            const float4 bm = xin ? make_float4(bias1.x * sc_mid, bias1.y * sc_mid, bias1.z * sc_mid, bias1.w * sc_mid) : make_float4(0.f, 0.f, 0.f, 0.f);
            const bool has1 = g < 2;                                   // second k-block: pieces 4, 5 = (ky 2, half g); 6, 7 are empty
            int at0 = (wv + (g >> 1)) * (IW / 2) + pr + (g & 1);
            int at1 = has1 ? (wv + 2) * (IW / 2) + pr + g : NIN;
            const int st1 = has1 ? 4 * (IW / 2) : 0, lo1 = has1 ? NIN / 2 : 0;
            int h8 = ((sN * PLANE + wv * HW + pr) << 1) + (g & 1);
#pragma unroll 1
            for (int y = wv; y < MH; y += 4, at0 += 4 * (IW / 2), at1 += st1, h8 += 8 * HW) {
                f32x4v acc = {0.f, 0.f, 0.f, 0.f};
                const h8v i0h = __builtin_bit_cast(h8v, inq[at0]), i0l = __builtin_bit_cast(h8v, inq[at0 + NIN / 2]);
                const h8v i1h = __builtin_bit_cast(h8v, inq[at1]), i1l = __builtin_bit_cast(h8v, inq[at1 + lo1]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1hi[0], i0l, acc, 0, 0, 0);      // small terms first
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1lo[0], i0h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1hi[0], i0h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1hi[1], i1l, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1lo[1], i1h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1hi[1], i1h, acc, 0, 0, 0);
                const bool yin = (unsigned)(ty0 - 1 + y) < (unsigned)a.H;       // wave-uniform
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (yin) v = make_float4(relu(fmaf(acc[0], unm, bm.x)), relu(fmaf(acc[1], unm, bm.y)), relu(fmaf(acc[2], unm, bm.z)), relu(fmaf(acc[3], unm, bm.w)));
                uint2 hi, lo;
                split4(v, hi, lo);
                midh[h8] = hi;
                midh[h8 + 2 * REGION] = lo;
            }
        }
        if (wv >= 2) {
            const int gi = MH + wv - 2;
            const bool extra = true;
            const int y = extra ? 16 * (gi - MH) + pr : gi, pc = extra ? 16 : pr;
            const int my = min(y, MH - 1);
            f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int piece = 4 * kb + g;                          // (ky, half) = (piece >> 1, piece & 1); pieces 6, 7 are empty
                const int at = piece < 6 ? ((my + (piece >> 1)) * IW + 2 * pc + 2 * (piece & 1)) >> 1 : NIN;
                const h8v ihi = __builtin_bit_cast(h8v, inq[at]);
                const h8v ilo = __builtin_bit_cast(h8v, inq[piece < 6 ? at + NIN / 2 : NIN]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1hi[kb], ilo, acc, 0, 0, 0);      // small terms first
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1lo[kb], ihi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1hi[kb], ihi, acc, 0, 0, 0);
            }
            const int gy = ty0 - 1 + y, gx = tx0 - 1 + 2 * pc + sN;
            const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;           // conv2 pads its INPUT (the ReLU'd map) with zeros
            float4 v = make_float4(relu(fmaf(acc[0], un1, bias1.x)) * sc_mid, relu(fmaf(acc[1], un1, bias1.y)) * sc_mid,
                                   relu(fmaf(acc[2], un1, bias1.z)) * sc_mid, relu(fmaf(acc[3], un1, bias1.w)) * sc_mid);
            if (!inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
            uint2 hi, lo;
            split4(v, hi, lo);
            if (y < MH) {
                const int h8 = ((sN * PLANE + y * HW + pc) << 1) + (g & 1);            // 8-byte half of the position's slot
                midh[h8] = hi;
                midh[h8 + 2 * REGION] = lo;
            }
        }
    }
    __syncthreads();
    // conv2 + ReLU: a wave owns TH/4 rows of the tile, one 32-pixel row = 16 pairs per MFMA group
    // piece (ky = kb, kx = g) of pair pr: tile column 2 pr + g of row + kb -> parity g & 1, slot pr + (g >> 1)
    const int abase = (g & 1) * PLANE + pr + (g >> 1);
    const int H2 = a.H / 2, W2 = a.W / 2;
    float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
    float xmax = 0.0f;                  // this lane's largest x1 value (x1 >= 0)
    const int gx = tx0 + 2 * pr + sN;
#pragma unroll 1          // (r05: unrolled by two -- both rows' MFMAs ahead of both epilogues -- 2.32 -> 2.375 ms, three interleaved runs: not kept)
    for (int rr = 0; rr < TH / 4; ++rr) {
        const int row = (TH / 4) * wv + rr;
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
            const int at = abase + (row + kb) * HW;
            const h8v ihi = __builtin_bit_cast(h8v, mid[at]);
            const h8v ilo = __builtin_bit_cast(h8v, mid[REGION + at]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhi[kb], ilo, acc, 0, 0, 0);      // small terms first
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(blo[kb], ihi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bhi[kb], ihi, acc, 0, 0, 0);
        }
        const int gy = ty0 + row;
        const float4 v = make_float4(relu(fmaf(acc[0], un2, bias2.x)), relu(fmaf(acc[1], un2, bias2.y)), relu(fmaf(acc[2], un2, bias2.z)), relu(fmaf(acc[3], un2, bias2.w)));
        if (gy < a.H && gx < a.W) {
            {   // non-temporal: x1 is read again only by the head, four kernels later (r05: 2.21 -> 2.16 ms; the same hint on block 2's a2 stores changes nothing)
                typedef float f4v __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store((f4v){v.x, v.y, v.z, v.w}, reinterpret_cast<f4v*>(a.x1 + ((size_t)b * P + (size_t)gy * a.W + gx) * 8 + c0));
            }
            xmax = kpb_pmax(kpb_pmax(xmax, kpb_pmax(v.x, v.y)), kpb_pmax(v.z, v.w));      // (x1 >= 0: integer maxima)
        }
        if ((rr & 1) == 0) {
            keep = v;
        } else {        // max_pool2d(x1, 2, 2): the row pair in registers, the other pixel of the pair two lane groups away
            float4 m = make_float4(kpb_pmax(keep.x, v.x), kpb_pmax(keep.y, v.y), kpb_pmax(keep.z, v.z), kpb_pmax(keep.w, v.w));
            m.x = kpb_pmax32(m.x); m.y = kpb_pmax32(m.y);
            m.z = kpb_pmax32(m.z); m.w = kpb_pmax32(m.w);
            const int py = gy >> 1, pxl = (tx0 >> 1) + pr;
            if (sN == 0 && py < H2 && pxl < W2) *reinterpret_cast<float4*>(a.p1 + (((size_t)b * H2 + py) * W2 + pxl) * 8 + c0) = m;
        }
    }
    amax_commit(xmax, ha.wmax_x1, ((b * gridDim.y + tile.y) * gridDim.x + tile.x) * 4 + wv, lane);
}

// conv1 of block 1, OIHW [8][3][3][3] -> alike_block1_h fragments [2 kb][hi / lo][64 lanes][8 halves]: lane (n = (s, cout), g),
// piece 4 kb + g = (ky, half) for pieces 0..5; element j = (kx = 2 half + (j >> 2), cin = j & 3): w[cout][cin][ky][kx - s]
std::vector<float> pack_b1c1_pairs(const float* w, float scale)
{
    std::vector<uint16_t> hl((size_t)2 * 2 * 64 * 8, 0);
    for (int kb = 0; kb < 2; ++kb)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int n = l & 15, g = l >> 4, s2 = n >> 3, co = n & 7, piece = 4 * kb + g;
                const int ky = piece >> 1, kx = 2 * (piece & 1) + (j >> 2) - s2, c = j & 3;
                const float v = (piece < 6 && c < 3 && kx >= 0 && kx <= 2) ? w[((size_t)co * 3 + c) * 9 + ky * 3 + kx] * scale : 0.0f;
                _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                memcpy(&hl[(((size_t)kb * 2 + 0) * 64 + l) * 8 + j], &hi, 2);
                memcpy(&hl[(((size_t)kb * 2 + 1) * 64 + l) * 8 + j], &lo, 2);
            }
    std::vector<float> out(hl.size() / 2);
    memcpy(out.data(), hl.data(), hl.size() * 2);
    return out;
}

// conv2 of block 1, OIHW [8][8][3][3] -> alike_block1_h fragments [3 kb = ky][hi / lo][64 lanes][8 halves = cin]:
// lane (n = (s, cout), g = kx of the 3 x 4 window) holds w[cout][cin][ky][kx - s] (zero outside the kernel)
std::vector<float> pack_b1c2_pairs(const float* w, float scale)
{
    std::vector<uint16_t> hl((size_t)3 * 2 * 64 * 8, 0);
    for (int kb = 0; kb < 3; ++kb)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int n = l & 15, g = l >> 4, s2 = n >> 3, co = n & 7, kx = g - s2;
                const float v = (kx >= 0 && kx <= 2) ? w[((size_t)co * 8 + j) * 9 + kb * 3 + kx] * scale : 0.0f;
                _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                memcpy(&hl[(((size_t)kb * 2 + 0) * 64 + l) * 8 + j], &hi, 2);
                memcpy(&hl[(((size_t)kb * 2 + 1) * 64 + l) * 8 + j], &lo, 2);
            }
    std::vector<float> out(hl.size() / 2);
    memcpy(out.data(), hl.data(), hl.size() * 2);
    return out;
}

// ------------------------------------------------------------------------------------------------ conv3x3
struct ConvArgs {
    const float* in;     // [B][H*POOL][W*POOL][CIN]
    float* out;          // [B][H][W][COUT]
    const float* w;      // [9][CIN][COUT]
    const float* bias;   // [COUT]
    const float* res_in; // RES: [B][H*RPOOL][W*RPOOL][CDS]
    const float* ds_w;   // [CDS][COUT]
    const float* ds_b;   // [COUT]
    float* res_out;      // DSOUT: [B][H][W][COUT] identity branch ds(pooled input) + ds_b, written for the block's conv2
    int H, W;            // output size
};

// Each thread: one pixel x 16 output channels.  Waves split into COUT/16 channel groups and 4/(COUT/16) pixel groups;
// weights are wave-uniform -> scalar loads.
//
// LDS layout (r02): the input tile is kept as CIN/4 PLANES of float4 (plane c4 = channels 4 c4 .. 4 c4 + 3) and a wave
// covers 64/TW rows of TW consecutive pixels, so the lanes of a ds_read_b128 sit on consecutive 16-byte slots.  With
// TW = 32 a whole 32-lane half-wave reads one tile row: every 16-lane group of the instruction hits 16 distinct 4-bank
// sets at any pitch; with TW = 16 the row pitch is 32 slots, which keeps the two rows of a group ({0-3, 12-15} of one
// row, {4-11} of the next) apart.  The r01 layout ([position][CIN], lanes CIN floats apart) measured 70-86 % conflict
// cycles.  Planes are padded to 4 (mod 8) slots so that the staging writes of neighbouring planes do not collide.
template <int CIN, int COUT, int POOL, bool RES, int CDS, int RPOOL, bool RELU, bool DSOUT = false, int TW = 32>
__global__ __launch_bounds__(256) void conv3x3_k(ConvArgs a)
{
    constexpr int G = COUT / 16, PG = 4 / G, RW = 64 / TW, TH = RW * PG, C4 = CIN / 4;
    constexpr int RP = TW == 16 ? 32 : TW + 2;                       // row pitch in float4 slots
    constexpr int NPOS = (TH + 2) * RP, NPOSP = NPOS + ((4 - NPOS % 8) + 8) % 8;
    static_assert(COUT % 16 == 0 && (G == 1 || G == 2 || G == 4) && CIN % 4 == 0 && (TW == 16 || TW == 32), "channel config");
    __shared__ __attribute__((aligned(16))) float4 tile[C4 * NPOSP];
    const int tid = threadIdx.x, b = blockIdx.z;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int cg = wv % G, pg = wv / G;
    const int ty0 = blockIdx.y * TH, tx0 = blockIdx.x * TW;
    const int Hi = a.H * POOL, Wi = a.W * POOL;
    const float* in = a.in + (size_t)b * Hi * Wi * CIN;

    for (int i = tid; i < (TH + 2) * (TW + 2) * C4; i += 256) {
        const int pos = i / C4, c4 = i - pos * C4;
        const int y = pos / (TW + 2), x = pos - y * (TW + 2);
        const int gy = ty0 - 1 + y, gx = tx0 - 1 + x;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
            const float* p = in + ((size_t)(gy * POOL) * Wi + (size_t)gx * POOL) * CIN + c4 * 4;
            v = *reinterpret_cast<const float4*>(p);
#pragma unroll
            for (int dy = 0; dy < POOL; ++dy)
#pragma unroll
                for (int dx = 0; dx < POOL; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    const float4 q = *reinterpret_cast<const float4*>(p + ((size_t)dy * Wi + dx) * CIN);
                    v.x = fmaxf(v.x, q.x); v.y = fmaxf(v.y, q.y); v.z = fmaxf(v.z, q.z); v.w = fmaxf(v.w, q.w);
                }
        }
        tile[c4 * NPOSP + y * RP + x] = v;
    }
    __syncthreads();

    const int row = pg * RW + lane / TW, col = lane % TW;
    const int gy = ty0 + row, gx = tx0 + col;
    float acc[16];
    const float* bias = a.bias + cg * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias[j];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float4* t = &tile[(row + ky) * RP + col + kx];
            const float* w = a.w + (size_t)((ky * 3 + kx) * CIN) * COUT + cg * 16;
#pragma unroll 2
            for (int c4 = 0; c4 < C4; ++c4) {
                const float4 v4 = t[c4 * NPOSP];
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[j] = fmaf(v[c], w[(c4 * 4 + c) * COUT + j], acc[j]);
            }
        }
    if (gy >= a.H || gx >= a.W) return;
    if (DSOUT) {   // the block's identity branch (ALike.py:76-77) from the pooled centre pixel already staged in LDS
        float ds[16];
        const float* db = a.ds_b + cg * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) ds[j] = db[j];
        const float4* t = &tile[(row + 1) * RP + col + 1];
        const float* dw = a.ds_w + cg * 16;
#pragma unroll 2
        for (int c4 = 0; c4 < C4; ++c4) {
            const float4 v4 = t[c4 * NPOSP];
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 16; ++j) ds[j] = fmaf(v[c], dw[(c4 * 4 + c) * COUT + j], ds[j]);
        }
        float* ro = a.res_out + ((size_t)b * a.H * a.W + (size_t)gy * a.W + gx) * COUT + cg * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(ro + 4 * q) = make_float4(ds[4 * q], ds[4 * q + 1], ds[4 * q + 2], ds[4 * q + 3]);
    }
    if (RES) {   // identity = downsample(x): 1x1 conv (with bias) on the block input (ALike.py:76-77)
        constexpr int D4 = CDS / 4;
        const int Hr = a.H * RPOOL, Wr = a.W * RPOOL;
        const float* p = a.res_in + ((size_t)b * Hr * Wr + (size_t)(gy * RPOOL) * Wr + (size_t)gx * RPOOL) * CDS;
        const float* dw = a.ds_w + cg * 16;
        const float* db = a.ds_b + cg * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] += db[j];
        for (int c4 = 0; c4 < D4; ++c4) {
            float4 v4 = *reinterpret_cast<const float4*>(p + c4 * 4);
#pragma unroll
            for (int dy = 0; dy < RPOOL; ++dy)
#pragma unroll
                for (int dx = 0; dx < RPOOL; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    const float4 q = *reinterpret_cast<const float4*>(p + ((size_t)dy * Wr + dx) * CDS + c4 * 4);
                    v4.x = fmaxf(v4.x, q.x); v4.y = fmaxf(v4.y, q.y); v4.z = fmaxf(v4.z, q.z); v4.w = fmaxf(v4.w, q.w);
                }
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = fmaf(v[c], dw[(c4 * 4 + c) * COUT + j], acc[j]);
        }
    }
    float* o = a.out + ((size_t)b * a.H * a.W + (size_t)gy * a.W + gx) * COUT + cg * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        if (RELU) { v.x = relu(v.x); v.y = relu(v.y); v.z = relu(v.z); v.w = relu(v.w); }
        *reinterpret_cast<float4*>(o + 4 * q) = v;
    }
}

// ------------------------------------------------------------------------------------------------ block 2 + agg2, fused
// ResBlock 8 -> 16 at H/2 (ALike.py:65-81, 139-140) and the aggregation 1x1 + ReLU of its output (147-148) in ONE kernel:
//   conv1 (8 -> 16) + ReLU on the (TH+2) x (TW+2) halo'd positions, from the pooled block-1 tile in LDS, back into LDS split;
//   conv2 (16 -> 16) + the block's 1x1 identity branch (one more K piece, read from the input tile already in LDS) + ReLU -> x2;
//   agg2 (16 -> 16, no bias) + ReLU on x2 -> a2, and a2 . w_score -> S2 (the group's share of the score logit).
// All three are implicit GEMMs on v_mfma_f32_16x16x32_f16 with every fp32 operand split into two half-precision terms (x = hi +
// lo, hi = f16(x), lo = f16(x - hi), to nearest; three MFMAs per product, the lo.lo term dropped: relative 2^-22 per product, fp32
// accumulation -- see alike_head_f16p):
//   M = 16 consecutive pixels of an image row, N = the 16 output channels, K = 32 per MFMA = four 16-byte pieces, piece
//   kidx = (tap, channel octet): lane (row i, group g) of k-block kb supplies piece 4 kb + g.  The input tile sits in LDS
//   already split, as planes of 16-byte slots [hi | lo][octet][position]: a lane's operand is ONE ds_read_b128 per plane, 16
//   consecutive slots per 16 lanes (conflict-free).  The weights of a lane (<= 5 k-blocks x hi / lo x 16 bytes) stay in registers
//   for the whole kernel.  The block's identity branch -- the 1x1 convolution of the pooled block input, ALike.py:76-79 -- is
//   one more 16-byte piece of conv2's K (free: K = 152 <= 160).
// As three kernels (r02: 0.77 + 1.80 + 1.00 ms per 512 images) the 16-channel intermediate t2 made a round trip through HBM and
// x2 was read back for the aggregation: 14.3 MB of avoidable traffic per image; fused, the block reads p1 once and writes
// x2, a2, S2 once.  agg2 needs x2 with the pixel on the lane (A operand) where the accumulator has the channel on the lane:
// each wave passes its 16 x 16 tile through 1 KB of LDS of its own (a wave's LDS operations execute in order: no barrier).
struct Block2Args {
    const float* p1;     // [B][H][W][8]
    float* x2; float* a2; float* S2;
    float* p2;           // [B][H/4][W/4][16] = max_pool2d(x2, 4, 4) (ALike.py:141), what block 3 reads; when given, x2 itself is not written
    const uint4* w1pk;   // pack_h16(b2c1.w, 8, null): 3 k-blocks
    const uint4* w2pk;   // pack_h16(b2c2.w, 16, b2ds.w): 5 k-blocks
    const uint4* wapk;   // pack_1x1_h16(agg2.w): 1 k-block
    const float* b1; const float* bsum; const float* wsg;
    int H, W;
    // dynamic operand range (see conv_mfma.h, cm_scale_of): reciprocal weight scales of the three packs, and the constants of
    // the two bounds |conv1 out| <= amax(p1) l1_c1 + bmax_c1, |x2| <= bound(conv1) l1_c2 + amax(p1) l1_ds + bmax_sum
    float inv_ws1, inv_ws2, inv_wsa, l1_c1, bmax_c1, l1_c2, l1_ds, bmax_sum;
    const unsigned* amax_x1;    // [B] float bits: largest value of the image's x1 (= of p1, its 2 x 2 max-pool)
    float* wmax_a2;             // [B][workgroups][4 waves]: every wave's largest a2 value, folded per image by amax_reduce for the head
    int xcd_map;                // kpb_xcd_tile: neighbouring tiles on one XCD
};


__global__ __launch_bounds__(256, 4) void alike_block2(Block2Args a)      // 39.7 KB of LDS = four workgroups per CU: the allocator stays at 128 registers
{
    // All three products are taken transposed (weights as the MFMA's A operand, the input pieces as B: see alike_block1_h): the
    // accumulator has the pixel on the LANE and channels 4 g .. 4 g + 3 in the registers of lane group g, so a lane owns one
    // position -- one bounds test, one packed split, 8-byte LDS writes -- and x2 / a2 leave as one 16-byte store per lane
    // (16 pixels x 64 B = 1 KB contiguous per instruction).
    constexpr int TH = 8, TW = 32, PH = TH + 4, PW = TW + 4, NP = PH * PW, MH = TH + 2, MW = TW + 2, NM = MH * MW;
    __shared__ __attribute__((aligned(16))) uint4 pin[2 * NP];          // [hi | lo][position]: the pooled block-1 tile, 8 channels per slot
    __shared__ __attribute__((aligned(16))) uint4 mid[4 * NM];          // [hi | lo][octet][position]: conv1's output
    __shared__ __attribute__((aligned(16))) uint4 xs[4][2][2][16];      // per wave: [hi | lo][octet][pixel] of the x2 group in flight
    __shared__ __attribute__((aligned(16))) uint4 zslot;
    const kpb_tile3 tile = kpb_xcd_tile(a.xcd_map);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = tile.z;
    const int ty0 = tile.y * TH, tx0 = tile.x * TW;
    const size_t P = (size_t)a.H * a.W;
    // r06: conv2's and agg2's fragments (48 registers) are requested AFTER conv1 -- while conv1 runs they would only keep the allocator from holding conv1's six
    // operand pieces in flight together (below); their round trip hides behind the barrier between the two products.
    h8v w1h[3], w1l[3], w2h[5], w2l[5];
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) { w1h[kb] = __builtin_bit_cast(h8v, a.w1pk[(kb * 2) * 64 + lane]); w1l[kb] = __builtin_bit_cast(h8v, a.w1pk[(kb * 2 + 1) * 64 + lane]); }
    // the lane's biases and score weights are requested with the fragments (read where they are used, each was an exposed round trip)
    const float4 b1v = *reinterpret_cast<const float4*>(a.b1 + 4 * (lane >> 4));
    const float4 bsum = *reinterpret_cast<const float4*>(a.bsum + 4 * (lane >> 4)), wsg = *reinterpret_cast<const float4*>(a.wsg + 4 * (lane >> 4));
    // One activation scale for the block input AND conv1's output (conv2 accumulates pieces of both: the identity branch reads
    // the input tile): that of the larger of amax(p1) and conv1's bound.  x2 gets its own (agg2 is a separate product).
    const float am_p = __uint_as_float(a.amax_x1[b]);
    const float bound1 = fmaf(am_p, a.l1_c1, a.bmax_c1);
    const int e1 = cm_exp_of(fmaxf(am_p, bound1));
    const int ex = cm_exp_of(fmaf(bound1, a.l1_c2, fmaf(am_p, a.l1_ds, a.bmax_sum)));
    const float sc1 = cm_scale_of(e1), un_c1s = a.inv_ws1, un_c2 = a.inv_ws2 * cm_unscale_of(e1);      // conv1: unscale(e1) x sc1 = 1
    const bool edge = ty0 == 0 || tx0 == 0 || ty0 + TH >= a.H || tx0 + TW >= a.W;
    const float scx = cm_scale_of(ex), un_a = a.inv_wsa * cm_unscale_of(ex);
    {   // stage the (TH+4) x (TW+4) tile of p1, split, zero outside the image
        const float* in = a.p1 + (size_t)b * P * 8;
        // all of a thread's loads go out before the first is used, to a clamped address without a branch (r03 stamps: as a plain
        // loop -- load, split, store, next -- the four round trips of this tile were 6.6 k of the workgroup's 23 k cycles)
        constexpr int PER = (NP * 2 + 255) / 256;
        float4 ld[PER];
        unsigned okm = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = min(tid + 256 * k, NP * 2 - 1);
            const int pos = i >> 1, q = i & 1;
            const int y = pos / PW, x = pos - y * PW;
            const int gy = ty0 - 2 + y, gx = tx0 - 2 + x;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) okm |= 1u << k;
            ld[k] = *reinterpret_cast<const float4*>(in + ((size_t)min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1)) * 8 + 4 * q);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + 256 * k;
            const int pos = i >> 1, q = i & 1;
            float4 v = ((okm >> k) & 1u) ? ld[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            v.x *= sc1; v.y *= sc1; v.z *= sc1; v.w *= sc1;
            uint2 hi, lo;
            split4(v, hi, lo);
            if (i < NP * 2) {
                uint2* d = reinterpret_cast<uint2*>(&pin[pos]) + q;
                d[0] = hi;
                d[2 * NP] = lo;
            }
        }
        if (tid == 0) zslot = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const int px = lane & 15, g = lane >> 4;                  // this lane's position slot (operand column and accumulator column); it holds channels 4 g .. 4 g + 3
    const uint4* zp = &zslot;
    {   // conv1 + ReLU -> mid (split).  Groups gi < 2 MH = (row gi / 2, columns 16 (gi & 1) .. + 15); the two rightmost columns of
        // all rows go to two extra groups: slot k of extra group e is position ((16 e + k) / 2, 32 + (k & 1))
        float4 bias1 = b1v;
        bias1.x *= sc1; bias1.y *= sc1; bias1.z *= sc1; bias1.w *= sc1;       // conv1's output is split at the scale of its input: the scale rides on the bias
        uint2* midh = reinterpret_cast<uint2*>(mid);
        static_assert((2 * MH) % 4 == 0, "alike_block2: the regular conv1 groups split evenly over the four waves");
        // r05 (as alike_block1_h): the 2 MH regular groups walk down the tile with everything that depends on the lane alone taken out of
        // the loop -- the three piece addresses (+ 2 PW slots per step; the empty taps 9..11 of lane groups 1..3 stay on the zero slot), the
        // slot written, the column test -- 53 -> 30 vector instructions per group of nine MFMAs; the general form is kept for the two
        // extra groups (waves 0 and 1).
        {
            const int xl = 16 * (wv & 1) + px, y0 = wv >> 1;
            const int gxl = tx0 - 1 + xl;
            const bool xin = gxl >= 0 && gxl < a.W;
            const float unm = xin ? un_c1s : 0.0f;                    // a column outside the image: relu(0 acc + 0) = 0, conv2's zero padding
            const float4 bm = xin ? bias1 : make_float4(0.f, 0.f, 0.f, 0.f);
            const bool has2 = 8 + g < 9;
            const uint4* q0 = &pin[(y0 + g / 3) * PW + xl + g % 3];
            const uint4* q1 = &pin[(y0 + (4 + g) / 3) * PW + xl + (4 + g) % 3];
            const uint4* q2 = has2 ? &pin[(y0 + 2) * PW + xl + 2] : zp;
            const int st2 = has2 ? 2 * PW : 0, lo2 = has2 ? NP : 0;
            int h8 = (((g >> 1) * NM + y0 * MW + xl) << 1) + (g & 1);
            // r06: a group's six operand pieces are requested while the PREVIOUS group's epilogue (ReLU, split, two LDS writes) runs, all six in flight
            // together.  Left alone the scheduler re-used ONE register quad for them -- ds_read_b128, s_waitcnt lgkmcnt(0), MFMA, six times per group: six
            // exposed LDS round trips (profiles/r06_block2_operand_prefetch_ab.txt).
            constexpr int NIT = 2 * MH / 4;
            h8v i0h = __builtin_bit_cast(h8v, q0[0]), i0l = __builtin_bit_cast(h8v, q0[NP]);
            h8v i1h = __builtin_bit_cast(h8v, q1[0]), i1l = __builtin_bit_cast(h8v, q1[NP]);
            h8v i2h = __builtin_bit_cast(h8v, q2[0]), i2l = __builtin_bit_cast(h8v, q2[lo2]);
#pragma unroll 1      // (unrolled by two the allocator needs 142 registers, or 128 and 48 bytes of scratch: 2.02 -> 2.60 ms)
            for (int it = 0; it < NIT; ++it, h8 += 4 * MW) {
                f32x4v acc = {0.f, 0.f, 0.f, 0.f};
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[0], i0l, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l[0], i0h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[0], i0h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[1], i1l, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l[1], i1h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[1], i1h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[2], i2l, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l[2], i2h, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[2], i2h, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (it + 1 < NIT) {         // the next group's pieces (the tile ends two rows below the last group's)
                    q0 += 2 * PW; q1 += 2 * PW; q2 += st2;
                    i0h = __builtin_bit_cast(h8v, q0[0]); i0l = __builtin_bit_cast(h8v, q0[NP]);
                    i1h = __builtin_bit_cast(h8v, q1[0]); i1l = __builtin_bit_cast(h8v, q1[NP]);
                    i2h = __builtin_bit_cast(h8v, q2[0]); i2l = __builtin_bit_cast(h8v, q2[lo2]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const bool yin = (unsigned)(ty0 - 1 + y0 + 2 * it) < (unsigned)a.H;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (yin) v = make_float4(relu(fmaf(acc[0], unm, bm.x)), relu(fmaf(acc[1], unm, bm.y)), relu(fmaf(acc[2], unm, bm.z)), relu(fmaf(acc[3], unm, bm.w)));
                uint2 hi, lo;
                split4(v, hi, lo);
                midh[h8] = hi;
                midh[h8 + 4 * NM] = lo;                                                // lo half: 2 NM slots = 4 NM 8-byte units further
            }
        }
        if (wv < 2) {
            const int gi = 2 * MH + wv;
            const int k = 16 * (gi - 2 * MH) + px;
            const int y = k >> 1, x = 32 + (k & 1);
            const int my = min(y, MH - 1);
            f32x4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                const int tap = 4 * kb + g;
                const bool has = tap < 9;
                const uint4* ph = has ? &pin[(my + tap / 3) * PW + x + tap % 3] : zp;
                const h8v ihi = __builtin_bit_cast(h8v, ph[0]);
                const h8v ilo = __builtin_bit_cast(h8v, has ? ph[NP] : zp[0]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[kb], ilo, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l[kb], ihi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[kb], ihi, acc, 0, 0, 0);
            }
            const int gy = ty0 - 1 + y, gx = tx0 - 1 + x;
            const bool inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;           // conv2 pads its INPUT with zeros
            float4 v = make_float4(relu(fmaf(acc[0], un_c1s, bias1.x)), relu(fmaf(acc[1], un_c1s, bias1.y)), relu(fmaf(acc[2], un_c1s, bias1.z)), relu(fmaf(acc[3], un_c1s, bias1.w)));
            if (edge && !inside) v = make_float4(0.f, 0.f, 0.f, 0.f);
            uint2 hi, lo;
            split4(v, hi, lo);
            if (y < MH) {
                const int h8 = (((g >> 1) * NM + y * MW + x) << 1) + (g & 1);         // 8-byte half (channels 4 g ..) of (octet g / 2, position)
                midh[h8] = hi;
                midh[h8 + 4 * NM] = lo;
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);       // (not before conv1's last group: see the fragments' comment above)
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) { w2h[kb] = __builtin_bit_cast(h8v, a.w2pk[(kb * 2) * 64 + lane]); w2l[kb] = __builtin_bit_cast(h8v, a.w2pk[(kb * 2 + 1) * 64 + lane]); }
    const h8v wah = __builtin_bit_cast(h8v, a.wapk[lane]), wal = __builtin_bit_cast(h8v, a.wapk[64 + lane]);
    __syncthreads();
    // conv2 + identity branch + ReLU -> x2; agg2 + ReLU -> a2, S2.  16 groups of 16 pixels per tile, four per wave: a wave owns
    // four rows of one 16-column half, so the 4 x 4 max-pool block 3 starts with is a running maximum over its four groups and
    // two lane exchanges, and only the pooled map (a sixteenth of x2) goes to HBM.
    float* x2 = a.x2 + (size_t)b * P * 16;
    float* a2 = a.a2 + (size_t)b * P * 16;
    float* S2 = a.S2 + (size_t)b * P;
    uint2* xh = reinterpret_cast<uint2*>(&xs[wv][0][0][0]);
    float4 pm = make_float4(0.f, 0.f, 0.f, 0.f);        // x2 >= 0 (ReLU): zero is the neutral element of the pool
    float amx = 0.0f;                                    // this lane's largest a2 value
#pragma unroll 1      // (r06: unrolled by two the operand pieces below cannot be held in flight without scratch: 19 registers spilled, 3.7 ms)
    for (int k4 = 0; k4 < 4; ++k4) {
        const int row = 4 * (wv >> 1) + k4, col0 = (wv & 1) * 16;
        const int mbase = row * MW + col0 + px;
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        h8v ihi[5], ilo[5];
        auto fetch = [&](int kb) {
            const int kidx = 4 * kb + g;
            const uint4* ph;
            int lo_off;
            if (kidx < 18) { const int tap = kidx >> 1; ph = &mid[(kidx & 1) * NM + mbase + (tap / 3) * MW + tap % 3]; lo_off = 2 * NM; }
            else if (kidx == 18) { ph = &pin[(row + 2) * PW + col0 + px + 2]; lo_off = NP; }
            else { ph = zp; lo_off = 0; }
            ihi[kb] = __builtin_bit_cast(h8v, ph[0]);
            ilo[kb] = __builtin_bit_cast(h8v, ph[lo_off]);
        };
        auto mm = [&](int kb) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[kb], ilo[kb], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2l[kb], ihi[kb], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[kb], ihi[kb], acc, 0, 0, 0);
        };
        // r06: three k-blocks' pieces in flight together, the other two requested behind the first six MFMAs: 24 operand registers at most (all ten at once
        // measure the same; the next group's first pieces requested under this group's epilogue cost two spilled registers and the gain)
        fetch(0); fetch(1); fetch(2);
        __builtin_amdgcn_sched_barrier(0);
        mm(0); mm(1);
        __builtin_amdgcn_sched_barrier(0);
        fetch(3); fetch(4);
        __builtin_amdgcn_sched_barrier(0);
        mm(2);
        __builtin_amdgcn_sched_barrier(0);
        mm(3); mm(4);
        const int gy = ty0 + row, gx = tx0 + col0 + px;
        const bool ok = gy < a.H && gx < a.W;
        const float4 v = make_float4(relu(fmaf(acc[0], un_c2, bsum.x)), relu(fmaf(acc[1], un_c2, bsum.y)), relu(fmaf(acc[2], un_c2, bsum.z)), relu(fmaf(acc[3], un_c2, bsum.w)));
        if (ok && !a.p2) *reinterpret_cast<float4*>(x2 + ((size_t)gy * a.W + gx) * 16 + 4 * g) = v;
        pm = make_float4(kpb_pmax(pm.x, v.x), kpb_pmax(pm.y, v.y), kpb_pmax(pm.z, v.z), kpb_pmax(pm.w, v.w));
        {   // the x2 group, split, to this wave's LDS strip with the channel octets as slots: the B operand of agg2
            uint2 hi, lo;
            split4(make_float4(v.x * scx, v.y * scx, v.z * scx, v.w * scx), hi, lo);
            const int h8 = (((g >> 1) * 16 + px) << 1) + (g & 1);
            xh[h8] = hi;
            xh[h8 + 64] = lo;                      // lo half: 2 x 16 slots = 64 8-byte units further
        }
        f32x4v ag = {0.f, 0.f, 0.f, 0.f};
        {
            const uint4* ph = g < 2 ? &xs[wv][0][g][px] : zp;
            const h8v ihi = __builtin_bit_cast(h8v, ph[0]);
            const h8v ilo = __builtin_bit_cast(h8v, g < 2 ? ph[32] : zp[0]);
            ag = __builtin_amdgcn_mfma_f32_16x16x32_f16(wah, ilo, ag, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_16x16x32_f16(wal, ihi, ag, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_16x16x32_f16(wah, ihi, ag, 0, 0, 0);
        }
        const float4 av = make_float4(relu(ag[0] * un_a), relu(ag[1] * un_a), relu(ag[2] * un_a), relu(ag[3] * un_a));
        if (ok) {
            *reinterpret_cast<float4*>(a2 + ((size_t)gy * a.W + gx) * 16 + 4 * g) = av;
            amx = kpb_pmax(kpb_pmax(amx, kpb_pmax(av.x, av.y)), kpb_pmax(av.z, av.w));
        }
        float sg = fmaf(av.w, wsg.w, fmaf(av.z, wsg.z, fmaf(av.y, wsg.y, av.x * wsg.x)));      // this group's share of the score logit
        sg = kpb_sum16(sg);
        sg = kpb_sum32(sg);
        if (g == 0 && ok) S2[(size_t)gy * a.W + gx] = sg;
    }
    if (a.p2) {     // columns: lanes px, px ^ 1, px ^ 2, px ^ 3 hold the four pixels of a pooled cell
        pm.x = kpb_pmax(pm.x, kpb_xor1(pm.x)); pm.y = kpb_pmax(pm.y, kpb_xor1(pm.y));
        pm.z = kpb_pmax(pm.z, kpb_xor1(pm.z)); pm.w = kpb_pmax(pm.w, kpb_xor1(pm.w));
        pm.x = kpb_pmax(pm.x, kpb_xor2(pm.x)); pm.y = kpb_pmax(pm.y, kpb_xor2(pm.y));
        pm.z = kpb_pmax(pm.z, kpb_xor2(pm.z)); pm.w = kpb_pmax(pm.w, kpb_xor2(pm.w));
        const int gy = ty0 + 4 * (wv >> 1), gx = tx0 + (wv & 1) * 16 + px;
        if ((px & 3) == 0 && gy < a.H && gx < a.W)
            *reinterpret_cast<float4*>(a.p2 + (size_t)b * (P / 16) * 16 + ((size_t)(gy >> 2) * (a.W >> 2) + (gx >> 2)) * 16 + 4 * g) = pm;
    }
    amax_commit(amx, a.wmax_a2, ((b * gridDim.y + tile.y) * gridDim.x + tile.x) * 4 + wv, lane);
}

// agg [16][16] (cout, cin) -> one k-block of fragments: piece g < 2 = channel octet g
std::vector<float> pack_1x1_h16(const float* w, float scale)
{
    std::vector<uint16_t> hl((size_t)2 * 64 * 8, 0);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const int n = l & 15, g = l >> 4;
            const float v = g < 2 ? w[n * 16 + 8 * g + j] * scale : 0.0f;
            _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            memcpy(&hl[((size_t)0 * 64 + l) * 8 + j], &hi, 2);
            memcpy(&hl[((size_t)1 * 64 + l) * 8 + j], &lo, 2);
        }
    std::vector<float> out(hl.size() / 2);
    memcpy(out.data(), hl.data(), hl.size() * 2);
    return out;
}

// OIHW [16][CIN][3][3] (+ identity-branch [16][8]) -> alike_block2 fragments [KB][hi / lo][64 lanes][8 halves], as floats (bit patterns)
std::vector<float> pack_h16(const float* w, int CIN, const float* ds_w, float scale)
{
    const int OCT = CIN / 8, NK = 9 * OCT + (ds_w ? 1 : 0), KB = (NK + 3) / 4;
    std::vector<uint16_t> hl((size_t)KB * 2 * 64 * 8, 0);
    for (int kb = 0; kb < KB; ++kb)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int n = l & 15, g = l >> 4, kidx = 4 * kb + g;
                float v = 0.0f;
                if (kidx < 9 * OCT) { const int tap = kidx / OCT, o = kidx - tap * OCT; v = w[((size_t)n * CIN + 8 * o + j) * 9 + tap]; }
                else if (ds_w && kidx == 9 * OCT) v = ds_w[n * 8 + j];
                v *= scale;
                _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                memcpy(&hl[(((size_t)kb * 2 + 0) * 64 + l) * 8 + j], &hi, 2);
                memcpy(&hl[(((size_t)kb * 2 + 1) * 64 + l) * 8 + j], &lo, 2);
            }
    std::vector<float> out(hl.size() / 2);
    memcpy(out.data(), hl.data(), hl.size() * 2);
    return out;
}

// ------------------------------------------------------------------------------------------------ 1x1 + ReLU
constexpr int ESTRIDE = 68;             // channels of a projected map: 64 descriptor rows + score row + pad (float4 aligned)

// out = relu(w . in) (ALike.py:147-150); smap = this group's share of the score logit; with E != null also the group's
// share of every head row, E[p][o] = sum_c wproj[c][o] out[p][c] (see alike_head: the head commutes with upsampling)
template <int CIN>
__device__ __forceinline__ void conv1x1_relu_body(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ w /*[CIN][16]*/,
                                                  const float* __restrict__ wsg /*[16]*/, float* __restrict__ smap, size_t npix,
                                                  const float* __restrict__ wproj /*[16][64]*/, float* __restrict__ E /*[npix][ESTRIDE]*/,
                                                  const unsigned bx /* workgroup index within this layer */)
// (__restrict__ matters: the projection weights are read AFTER the stores of out / smap; without it the compiler could not prove them
//  unclobbered and fetched all 1 024 of them with per-lane vector loads, 256 serialised round trips per wave -- 0.46 ms for agg3)
{
    const size_t p = (size_t)bx * 256 + threadIdx.x;
    const bool live = p < npix;        // (dead lanes of the last workgroup still help to copy the projected rows out)
    // One LDS region per wave, used twice: first to turn the wave's 64 pixels x CIN floats -- contiguous in memory, read with
    // consecutive lanes on consecutive 16-byte pieces -- into one pixel per lane (a lane reading its own pixel straight from global
    // memory touches 64 different lines per load instruction: r03 measured 0.46 ms for agg3 against 0.14 ms of traffic), at the end
    // for the projected rows (below).  A wave's LDS operations execute in order: no barrier.
    constexpr bool STAGE_IN = CIN <= 32;                   // (CIN = 64: 70 KB, and agg4 runs on 1/1024 of the pixels)
    constexpr int XP = CIN + 4;                            // floats per staged pixel: lanes 36 floats apart read 16 bytes without bank conflicts
    constexpr int WREG = STAGE_IN && 64 * XP > 16 * ESTRIDE ? 64 * XP : 16 * ESTRIDE;
    __shared__ __attribute__((aligned(16))) float wlds[4][WREG];
    float* wreg = wlds[threadIdx.x >> 6];
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    const float* x = in + (live ? p : 0) * CIN;
    if (STAGE_IN) {
        const int ln = threadIdx.x & 63;
        const size_t pw = (size_t)bx * 256 + (threadIdx.x >> 6) * 64;     // first pixel of this wave
        float4 ld[CIN / 4];
#pragma unroll
        for (int i = 0; i < CIN / 4; ++i) {
            const int j = ln + 64 * i, px = j / (CIN / 4);
            ld[i] = pw + px < npix ? *reinterpret_cast<const float4*>(in + pw * CIN + 4 * (size_t)j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < CIN / 4; ++i) {
            const int j = ln + 64 * i, px = j / (CIN / 4), q = j - px * (CIN / 4);
            *reinterpret_cast<float4*>(wreg + px * XP + 4 * q) = ld[i];
        }
        x = wreg + ln * XP;
    }
#pragma unroll 2
    for (int c4 = 0; c4 < CIN / 4; ++c4) {
        const float4 v4 = *reinterpret_cast<const float4*>(x + c4 * 4);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fmaf(v[c], w[(c4 * 4 + c) * 16 + j], acc[j]);
    }
    float sg = 0.0f;     // this group's share of the score logit: the 1x1 head commutes with the bilinear upsampling
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc[j] = relu(acc[j]); sg = fmaf(acc[j], wsg[j], sg); }
    if (live) {
        float* o = out + p * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(o + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        smap[p] = sg;
    }
    if (E) {
        // 68 floats per pixel, 272 bytes apart: written per thread they would be 17 scattered 16-byte pieces per lane.
        // Instead each wave passes its pixels through LDS 16 at a time and writes the 16 x 272 contiguous bytes with
        // consecutive lanes on consecutive float4s (a wave's LDS operations execute in order: no barrier)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        float* st = wreg;
        const size_t p0 = (size_t)bx * 256 + wv * 64;          // first pixel of this wave
        float4 r[17];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            r[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float* wr = wproj + c * 64 + 4 * q;
                r[q].x = fmaf(acc[c], wr[0], r[q].x); r[q].y = fmaf(acc[c], wr[1], r[q].y);
                r[q].z = fmaf(acc[c], wr[2], r[q].z); r[q].w = fmaf(acc[c], wr[3], r[q].w);
            }
        }
        r[16] = make_float4(sg, 0.f, 0.f, 0.f);
        for (int g = 0; g < 4; ++g) {
            if ((lane >> 4) == g) {
                float* e = st + (lane & 15) * ESTRIDE;
#pragma unroll
                for (int q = 0; q < 17; ++q) *reinterpret_cast<float4*>(e + 4 * q) = r[q];
            }
            const size_t base = p0 + 16 * g;                            // 16 pixels = 16 * 17 float4, contiguous in E
            for (int j = lane; j < 16 * (ESTRIDE / 4); j += 64)
                if (base + j / (ESTRIDE / 4) < npix)
                    *reinterpret_cast<float4*>(E + base * ESTRIDE + 4 * j) = *reinterpret_cast<const float4*>(st + 4 * j);
        }
    }
}

template <int CIN>
__global__ __launch_bounds__(256) void conv1x1_relu(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ w,
                                                    const float* __restrict__ wsg, float* __restrict__ smap, size_t npix,
                                                    const float* __restrict__ wproj, float* __restrict__ E)
{
    conv1x1_relu_body<CIN>(in, out, w, wsg, smap, npix, wproj, E, blockIdx.x);
}

// agg3 and agg4 as ONE launch: workgroups [0, nb3) are agg3's, the rest agg4's.  For a handful of images only (every launch of the
// drop-in path's single image is 5-6 us of dispatch for microseconds of work); a batch keeps the two kernels, whose LDS adds up here.
struct AggArgs { const float* in; float* out; const float* w; const float* wsg; float* smap; size_t npix; const float* wproj; float* E; };
__global__ __launch_bounds__(256) void conv1x1_relu_34(AggArgs a3, AggArgs a4, unsigned nb3)
{
    if (blockIdx.x < nb3) conv1x1_relu_body<32>(a3.in, a3.out, a3.w, a3.wsg, a3.smap, a3.npix, a3.wproj, a3.E, blockIdx.x);
    else conv1x1_relu_body<64>(a4.in, a4.out, a4.w, a4.wsg, a4.smap, a4.npix, a4.wproj, a4.E, blockIdx.x - nb3);
}

// ------------------------------------------------------------------------------------------------ head
struct HeadArgs {
    const float* x1;   // [B][H][W][8]
    const float* a2;   // [B][H/2][W/2][16]
    const float* a3;   // [B][H/8][W/8][16]
    const float* a4;   // [B][H/32][W/32][16]
    const float* agg1; // [8][16]
    const float* whT;  // [64][64]  whT[c][o] = head.w[o][c], o < 64
    const float* wsc;  // [64]      head.w[64][c]
    float* score;      // [B][H][W]
    float* desc;       // [B][H][W][64] or null
    int H, W;
};


// 8 channels [c0, c0+8) of an align_corners=True bilinear upsample (nn.Upsample, ALike.py:126-129) at (y, x)
__device__ __forceinline__ void up8ch(const float* m, int Hs, int Ws, float sy, float sx, int y, int x, int c0, float* f)
{
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const float* p00 = m + ((size_t)y0 * Ws + x0) * 16 + c0;
    const float* p01 = m + ((size_t)y0 * Ws + x1) * 16 + c0;
    const float* p10 = m + ((size_t)y1 * Ws + x0) * 16 + c0;
    const float* p11 = m + ((size_t)y1 * Ws + x1) * 16 + c0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(p00 + 4 * q), b = *reinterpret_cast<const float4*>(p01 + 4 * q);
        const float4 c = *reinterpret_cast<const float4*>(p10 + 4 * q), d = *reinterpret_cast<const float4*>(p11 + 4 * q);
        f[4 * q + 0] = hy * (hx * a.x + lx * b.x) + ly * (hx * c.x + lx * d.x);
        f[4 * q + 1] = hy * (hx * a.y + lx * b.y) + ly * (hx * c.y + lx * d.y);
        f[4 * q + 2] = hy * (hx * a.z + lx * b.z) + ly * (hx * c.z + lx * d.z);
        f[4 * q + 3] = hy * (hx * a.w + lx * b.w) + ly * (hx * c.w + lx * d.w);
    }
}

// the same for taps already in registers (the head reads them from the a2 band it keeps in LDS)
struct Up8Taps { float4 t[8]; float ly, lx; };
// The bilinear interpolation of up8ch from taps already in registers, as four weighted taps: the four weights once per pixel, then a
// multiply and three FMAs per channel (36 vector instructions per pixel; r03's hy (hx a + lx b) + ly (hx c + lx d), unfused, took 56 and
// differs in the last bit).  `sc` (a power of two) scales the result exactly: it rides on the two y weights.
// (History: this form went in at 775c39f and out again a day later, because from that commit on ~0.5 % of the dense maps differed RUN TO
// RUN in 16-pixel groups.  r05 found why -- not this arithmetic but ONE encoding hipcc chose for it: it splats the two lower weights with
// `v_pk_mul_f32 w, ly, {hx, lx} op_sel:[0,1] op_sel_hi:[0,1]`, and gfx950 drops src1's high dword from the LOW lane of a packed-fp32
// instruction with op_sel[0] = 0, op_sel[1] = 1 in lanes 48..63 now and then while an f16 MFMA is executing on the SIMD (DESIGN.md
// section 3; scripts/ubench/pk_opsel.hip reproduces it in isolation).  keypoint_bench_amd/isa_fixup.py now rewrites that encoding in
// every translation unit before it is assembled, scripts/isa_lint.py checks the library for it, and the form is back.)
__device__ __forceinline__ void up8ch_lerp4(const Up8Taps& u, float* f, float sc)
{
    const float lx = u.lx, hx = 1.0f - lx, hy = (1.0f - u.ly) * sc, ly = u.ly * sc;
    const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float4 a = u.t[4 * q], b = u.t[4 * q + 1], c = u.t[4 * q + 2], d = u.t[4 * q + 3];
        f[4 * q + 0] = fmaf(w11, d.x, fmaf(w10, c.x, fmaf(w01, b.x, w00 * a.x)));
        f[4 * q + 1] = fmaf(w11, d.y, fmaf(w10, c.y, fmaf(w01, b.y, w00 * a.y)));
        f[4 * q + 2] = fmaf(w11, d.z, fmaf(w10, c.z, fmaf(w01, b.z, w00 * a.z)));
        f[4 * q + 3] = fmaf(w11, d.w, fmaf(w10, c.w, fmaf(w01, b.w, w00 * a.w)));
    }
}

// ------------------------------------------------------------------------------------------------ head, hybrid form
// convhead2 has no bias and bilinear upsampling is linear, so  W.up(a_g) = up(W.a_g)  (ALike.py:151-159).  For the two
// coarse groups that is a large saving: E3 = W3.a3 and E4 = W4.a4 are projected once at 1/64 and 1/1024 of the pixels
// (inside the aggregation kernels; channel 64 of E carries the score row), and at full resolution their bilinear interpolation becomes
// a handful of extra K steps of the same MFMA accumulation: along a 32-pixel tile of one image row an up8 map touches
// at most 6 source columns and an up32 map at most 3, so
//     desc[pixel][:] = sum_c f[pixel][c] W[c][:]  (c < 32: agg1 and up2(a2), as before)
//                    + sum_t hat3_t(pixel) V3[t][:] + sum_t hat4_t(pixel) V4[t][:]
// where V3/V4 are the rows of E3/E4 already interpolated in y (one strip per 128-pixel row segment, in LDS) and hat_t
// are the x interpolation weights, computed per lane and fed as the A operand.  K drops from 64 to 32 + 6 + 4 and the
// per-pixel bilinear feature work of the two coarse groups disappears.  (Doing the same to up2(a2) was measured and
// loses: 18 source columns per tile and a 64-channel strip cost more than its K = 16.)
struct HybArgs {
    const float* x1;   // [B][H][W][8]
    const float* a2;   // [B][H/2][W/2][16]
    const float* E3;   // [B][H/8][W/8][ESTRIDE]    W3.a3, channel 64 = score share
    const float* E4;   // [B][H/32][W/32][ESTRIDE]  W4.a4
    const float* agg1; // [8][16]
    const float* whT;  // [64][64]  whT[c][o]; rows 0..31 used here
    const float* wsc;  // [64]
    float* score; float* desc;
    int H, W;
    // split-f16 form only: dynamic operand range (conv_mfma.h, cm_scale_of).  The fine features are bounded per image by
    // max(amax(x1) l1_agg1, amax(a2)) -- relu(agg1 x1) and a convex combination of a2 values
    const unsigned* amax_x1; const unsigned* amax_a2;     // [B] float bits, written by blocks 1 and 2
    float l1_agg1;     // max over agg1's output channels of sum |w|
    float ws_h, inv_ws_h;      // power-of-two scale the head's weight fragments were packed with, and its reciprocal
    const uint4* a1h16;        // alike_head_f16p: agg1's weights as MFMA A-operand fragments [hi / lo][64 lanes], scaled by 1 / inv_wa
    float inv_wa;
};

constexpr int SEG_TILES = 4;            // a wave owns one 128-pixel row segment
constexpr int NT3 = 18, NT4 = 7;        // strip rows a segment can touch: floor(127/8)+... see the bounds below

// rows [tb, tb + NT) of map E (clamped to the last column), interpolated in y, into a wave's LDS strip
// `c` (a power of two; exact) scales the 64 descriptor rows of the strip, NOT the score share in channel 64.
// Two phases, so that a caller can put every load of its prologue in flight before the first of them is waited for: as one
// loop (load, interpolate, store, next) the strip cost seven dependent L2 round trips per workgroup -- with the a2 band 5.8 us
// of a workgroup's ~20 us, and a store stream running at 0.7 of the fill rate because of it (r03, stores-only build).
template <int NT>
struct StripRegs { float4 u[(NT * (ESTRIDE / 4) + 63) / 64], d[(NT * (ESTRIDE / 4) + 63) / 64]; };

template <int NT>
__device__ __forceinline__ void strip_load(StripRegs<NT>& r, const float* E, int Hs, int Ws, float sy, int y, int tb, int lane)
{
    const float fy = sy * (float)y;
    const int y0 = (int)fy, y1 = y0 + (y0 < Hs - 1 ? 1 : 0);
#pragma unroll
    for (int k = 0; k < (NT * (ESTRIDE / 4) + 63) / 64; ++k) {
        const int i = min(lane + 64 * k, NT * (ESTRIDE / 4) - 1);
        const int t = i / (ESTRIDE / 4), q = i - t * (ESTRIDE / 4);
        const int xt = min(tb + t, Ws - 1);
        r.u[k] = *reinterpret_cast<const float4*>(E + ((size_t)y0 * Ws + xt) * ESTRIDE + 4 * q);
        r.d[k] = *reinterpret_cast<const float4*>(E + ((size_t)y1 * Ws + xt) * ESTRIDE + 4 * q);
    }
}

template <int NT>
__device__ __forceinline__ void strip_store(const StripRegs<NT>& r, float* V, int Hs, float sy, int y, int lane, float c)
{
    const float fy = sy * (float)y;
    const int y0 = (int)fy;
    const float ly = fy - (float)y0, hy = 1.0f - ly;
#pragma unroll
    for (int k = 0; k < (NT * (ESTRIDE / 4) + 63) / 64; ++k) {
        const int i = lane + 64 * k;
        const int t = i / (ESTRIDE / 4), q = i - t * (ESTRIDE / 4);
        const float cq = q < 16 ? c : 1.0f;
        const float4 u = r.u[k], d = r.d[k];
        float4 o;
        o.x = (hy * u.x + ly * d.x) * cq; o.y = (hy * u.y + ly * d.y) * cq; o.z = (hy * u.z + ly * d.z) * cq; o.w = (hy * u.w + ly * d.w) * cq;
        if (i < NT * (ESTRIDE / 4)) *reinterpret_cast<float4*>(V + t * ESTRIDE + 4 * q) = o;
    }
}

// rows [tb, tb + NT) of map E (clamped to the last column), interpolated in y, into a wave's LDS strip
template <int NT>
__device__ __forceinline__ void build_strip(float* V, const float* E, int Hs, int Ws, float sy, int y, int tb, int lane, float c = 1.0f)
{
    StripRegs<NT> r;
    strip_load<NT>(r, E, Hs, Ws, sy, y, tb, lane);
    strip_store<NT>(r, V, Hs, sy, y, lane, c);
}

// Lane (p, h) of a wave owns pixel p of the tile and, of each of the two fine 16-channel groups, channels 8h..8h+7:
// every lane runs the same instruction stream (no divergence between the wave halves), the A operand comes straight
// from the registers that built the features, and step s of the K loop consumes channel 16*(s/8) + 8h + (s%8) from A
// and B alike (the MFMA does not care which k a lane calls its own, as long as A and B agree).  In the tap steps the
// lane supplies the weight of pixel p for source column 2s + h.
// A workgroup's four waves own the SAME 128-pixel column band of four consecutive rows: the <= 3 a2 rows and the 2 E3 / E4
// rows those rows tap are fetched once per workgroup (they meet in the CU's L1 / the XCD's L2).  (r01's raster-order
// segments fetched every a2 row once per output row that taps it, 2.6x read amplification; an XCD-contiguous walk down a
// column band cut the fetches further without moving the time: the kernel is not fetch-bound.)
// (Measured and rejected, r02: swapping the MFMA operands so that a lane holds four consecutive channels of its pixel and
// stores 16-byte pieces -- 8 store instructions per tile instead of 32, but each touching 32 lines partially: 32.7 ms.)
__global__ __launch_bounds__(256) void alike_head_hyb(HybArgs a)
{
    __shared__ __attribute__((aligned(16))) float Bl[16 * 2 * 64];   // [s][h][out]: head weight of chan(s,h), s < 16
    __shared__ __attribute__((aligned(16))) float A1[2 * 8 * 8];      // [h][cin][j]:  agg1 weight of output 8h+j
    __shared__ __attribute__((aligned(16))) float Ws[2 * 16];         // [h][s]:       score weight of chan(s,h)
    __shared__ __attribute__((aligned(16))) float V3[4][NT3 * ESTRIDE];
    __shared__ __attribute__((aligned(16))) float V4[4][NT4 * ESTRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    for (int i = tid; i < 16 * 2 * 64; i += 256) {
        const int o = i & 63, hh = (i >> 6) & 1, s = i >> 7;
        Bl[i] = a.whT[(16 * (s >> 3) + 8 * hh + (s & 7)) * 64 + o];
    }
    if (tid < 128) { const int j = tid & 7, c = (tid >> 3) & 7, hh = tid >> 6; A1[tid] = a.agg1[c * 16 + 8 * hh + j]; }
    if (tid < 32) { const int s = tid & 15, hh = tid >> 4; Ws[tid] = a.wsc[16 * (s >> 3) + 8 * hh + (s & 7)]; }

    const int H2 = a.H / 2, W2 = a.W / 2, H8 = a.H / 8, W8 = a.W / 8, H32 = a.H / 32, W32 = a.W / 32;
    const float* a2 = a.a2 + (size_t)b * H2 * W2 * 16;
    const float* E3 = a.E3 + (size_t)b * H8 * W8 * ESTRIDE;
    const float* E4 = a.E4 + (size_t)b * H32 * W32 * ESTRIDE;
    const float sy2 = (float)(H2 - 1) / (float)(a.H - 1), sx2 = (float)(W2 - 1) / (float)(a.W - 1);
    const float sy8 = (float)(H8 - 1) / (float)(a.H - 1), sx8 = (float)(W8 - 1) / (float)(a.W - 1);
    const float sy32 = (float)(H32 - 1) / (float)(a.H - 1), sx32 = (float)(W32 - 1) / (float)(a.W - 1);
    const int tiles_per_row = a.W / 32, segs_per_row = (tiles_per_row + SEG_TILES - 1) / SEG_TILES;
    int y, xs;
    bool live;
    {
        const int w = blockIdx.x, band = w % segs_per_row, grp = w / segs_per_row;
        y = 4 * grp + wv; xs = band * (32 * SEG_TILES);
        live = y < a.H;
        if (!live) { y = 0; xs = 0; }
    }
    const int ntile = live ? min(SEG_TILES, tiles_per_row - xs / 32) : 0;
    // a 128-pixel segment spans < 16 source columns of the up8 map and < 4 of the up32 map (scale < 1/8, 1/32), a tile
    // starts at most 12 (3) columns into the strip and uses 6 (4) rows from there: NT3 = 12 + 6, NT4 = 3 + 4
    const int tb3 = (int)(sx8 * (float)xs), tb4 = (int)(sx32 * (float)xs);
    if (live) {
        build_strip<NT3>(V3[wv], E3, H8, W8, sy8, y, tb3, lane);
        build_strip<NT4>(V4[wv], E4, H32, W32, sy32, y, tb4, lane);
    }
    __syncthreads();

    for (int t = 0; t < ntile; ++t) {
        const int x0 = xs + 32 * t, x = x0 + p;
        const size_t pix = (size_t)b * a.H * a.W + (size_t)y * a.W + x;
        int z = 0;                              // opaque zero: keeps the tile-invariant LDS reads inside the loop
        asm volatile("" : "+v"(z));             // (hoisted, they would pin ~50 registers across it)
        const float* Blz = Bl + z; const float* A1z = A1 + z; const float* Wsz = Ws + z;
        float f[16];
        {   // group 0: relu(agg1 . x1), outputs 8h..8h+7 (ALike.py:147)
            const float4 lo = *reinterpret_cast<const float4*>(a.x1 + pix * 8), hi = *reinterpret_cast<const float4*>(a.x1 + pix * 8 + 4);
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 w0 = *reinterpret_cast<const float4*>(&A1z[(h * 8 + c) * 8]), w1 = *reinterpret_cast<const float4*>(&A1z[(h * 8 + c) * 8 + 4]);
                f[0] = fmaf(v[c], w0.x, f[0]); f[1] = fmaf(v[c], w0.y, f[1]); f[2] = fmaf(v[c], w0.z, f[2]); f[3] = fmaf(v[c], w0.w, f[3]);
                f[4] = fmaf(v[c], w1.x, f[4]); f[5] = fmaf(v[c], w1.y, f[5]); f[6] = fmaf(v[c], w1.z, f[6]); f[7] = fmaf(v[c], w1.w, f[7]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = relu(f[j]);
        }
        up8ch(a2, H2, W2, sy2, sx2, y, x, 8 * h, f + 8);      // ALike.py:151

        // x interpolation of the two coarse groups: source column and weight of this lane's pixel, relative to the tile
        const float fx3 = sx8 * (float)x, fx4 = sx32 * (float)x;
        const int rb3 = (int)(sx8 * (float)x0) - tb3, rb4 = (int)(sx32 * (float)x0) - tb4;     // tile's first strip row
        const int t3 = (int)fx3 - tb3 - rb3, t4 = (int)fx4 - tb4 - rb4;
        const float lx3 = fx3 - (float)(int)fx3, lx4 = fx4 - (float)(int)fx4;
        const float* v3 = V3[wv] + rb3 * ESTRIDE;
        const float* v4 = V4[wv] + rb4 * ESTRIDE;

        float sc = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = *reinterpret_cast<const float4*>(&Wsz[h * 16 + 4 * q]);
            sc = fmaf(f[4 * q], w.x, sc); sc = fmaf(f[4 * q + 1], w.y, sc); sc = fmaf(f[4 * q + 2], w.z, sc); sc = fmaf(f[4 * q + 3], w.w, sc);
        }
        sc = kpb_sum32(sc);
        sc += (1.0f - lx3) * v3[t3 * ESTRIDE + 64] + lx3 * v3[(t3 + 1) * ESTRIDE + 64];
        sc += (1.0f - lx4) * v4[t4 * ESTRIDE + 64] + lx4 * v4[(t4 + 1) * ESTRIDE + 64];
        if (h == 0) a.score[pix] = __fdiv_rn(1.0f, 1.0f + expf(-sc));   // torch.sigmoid (ALike.py:162)

        f32x16 acc0 = {0}, acc1 = {0};
#define HEAD_MFMA(fv, w0, w1)                                                       \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fv, w0, acc0, 0, 0, 0);            \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fv, w1, acc1, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float b0 = Blz[(s * 2 + h) * 64 + p], b1 = Blz[(s * 2 + h) * 64 + 32 + p];
            HEAD_MFMA(f[s], b0, b1)
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = 2 * s + h;
            const float w = k == t3 ? 1.0f - lx3 : (k == t3 + 1 ? lx3 : 0.0f);
            const float b0 = v3[k * ESTRIDE + p], b1 = v3[k * ESTRIDE + 32 + p];
            HEAD_MFMA(w, b0, b1)
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = 2 * s + h;
            const float w = k == t4 ? 1.0f - lx4 : (k == t4 + 1 ? lx4 : 0.0f);
            const float b0 = v4[k * ESTRIDE + p], b1 = v4[k * ESTRIDE + 32 + p];
            HEAD_MFMA(w, b0, b1)
        }
#undef HEAD_MFMA
        // 40 GB per launch that nothing re-reads before they have left every cache: streaming stores keep the L2 for a2
        // and the strips
        float* d = a.desc + ((size_t)b * a.H * a.W + (size_t)y * a.W + x0) * 64;
        // D[row = pixel][col = out channel]: lane holds channel p (+32), rows (r&3) + 8*(r>>2) + 4h
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rowp = (r & 3) + 8 * (r >> 2) + 4 * h;
            __builtin_nontemporal_store(acc0[r], d + (size_t)rowp * 64 + p);
            __builtin_nontemporal_store(acc1[r], d + (size_t)rowp * 64 + 32 + p);
        }
    }
}

// ------------------------------------------------------------------------------------------------ head, split-f16 form
// fp32 MFMA on gfx950 runs at the fp32 VECTOR rate and (measured, r02: the kernel takes 10.9 ms with its stores removed)
// shares the datapath with the VALU work that builds its operands, so the 32-deep fine-group product was the largest
// single cost of the head.  Here it runs on v_mfma_f32_32x32x16_f16 (16x the fp32 rate per K) with every fp32 operand
// split into two halves-precision terms, x = hi + lo with hi = f16(x) and lo = f16(x - hi) (both to nearest):
//     a b  =  a_hi b_hi + a_hi b_lo + a_lo b_hi  (+ a_lo b_lo, dropped: <= 2^-22 |a b|)
// Three MFMAs per 16-deep block accumulate in fp32; the dropped term and the rounding of the lo parts are at the level of
// fp32's own rounding (relative 2^-22 per product against 2^-24), far inside the 1e-4 descriptor contract, and the result
// does not depend on a reduced-precision mode: x - hi is exact in fp32.  The coarse groups' ten tap steps stay on the
// fp32 MFMA (their B operand is the y-interpolated strip, rebuilt per row; K = 10).
// Per 32-pixel tile: 12 f16 MFMAs (32 cycles) + 10 fp32 MFMAs (64 cycles) = 1024 matrix cycles instead of 2688.
__device__ __forceinline__ void split8(const float* f, h8v& hi, h8v& lo)
{
    uint4 a, b;
    cm_split2(f[0], f[1], a.x, b.x); cm_split2(f[2], f[3], a.y, b.y);
    cm_split2(f[4], f[5], a.z, b.z); cm_split2(f[6], f[7], a.w, b.w);
    hi = __builtin_bit_cast(h8v, a); lo = __builtin_bit_cast(h8v, b);
}

// How a tile's stores leave (alike_head_f16p below).  r02: the row stores are issued BETWEEN matrix instructions, two per MFMA, and
// the rest under the NEXT tile's loads and feature arithmetic, instead of 32 in one burst at the end: a wave that meets a full store
// queue then stalls for one queue slot while its MFMAs run, not for the drain of a whole 8 KB burst with the SIMD's other waves doing
// the same (stores in one burst, four waves per SIMD, no early line touch, plain instead of streaming line touch: all measured
// slower and gone).  r03: the fine features are scaled by the power of two that fits the image's bound (HybArgs), for free -- the
// scale rides on the agg1 weights staged in LDS and on the two y weights of the a2 interpolation; the projected coarse rows carry the
// same units, and the accumulator is scaled back before the store; the a2 pixels' four 16-byte slots are XOR-swizzled by
// (column / 4) % 4 in LDS so that the 8 columns a ds_read_b128 lane group touches fall on distinct banks.  r04: one store instruction
// = one whole pixel = 256 contiguous bytes -- v_permlane32_swap trades the upper half of acc0[r] (pixel R + 4, channels 0..31) for
// the lower half of acc1[r] (pixel R, channels 32..63); r03 wrote two 128-byte half pixels per instruction and the other halves a
// tile later.  Alone that is worth 5 % to the memory system beside loads (scripts/hbm_store_patterns.hip,
// profiles/r04_hbm_store_patterns.txt: 5.50 -> 5.77 TB/s) and 0.4 % to r03's one-group kernel (profiles/r04_ab_knobs.txt).
// ------------------------------------------------------------------------------------------------ head, persistent form (r04)
// r03's stamps (profiles/r03_head_stamps.txt): a workgroup of r03's one-group kernel (alike_head_f16: four rows x 128 pixels, deleted) lived 45.5 k cycles, 18.6 k of them in a prologue that
// stores nothing -- 27 loads per wave queued behind the other workgroups' stores, strips built, a barrier -- for four tiles per wave.
// Here a workgroup is PERSISTENT down a 64-pixel column band: it walks `groups_per_wg` row groups (4 rows, one per wave, two 32-pixel
// tiles each) and everything the NEXT groups need arrives while the current one computes:
//   * the a2 rows live in an LDS ring of 8 rows (absolute row & 7); consecutive groups share half of their rows, so a group
//     brings two new rows of 34 pixels (4.4 KB) instead of a 16.9 KB band;
//   * the projected coarse rows E3 / E4 are kept RAW in rings of 4 rows (row & 3) and interpolated in y where they are used
//     (hy, ly carry the accumulator's power-of-two scale: the values are bit for bit those of r03's per-wave strips), so nothing
//     is rebuilt per row and a new E3 row (2.7 KB) is needed every other group;
//   * the rows of group g + 2 are requested by LDS-DMA (global_load_lds_dwordx4: no registers) right after the barrier of group g;
//     a wave's vector-memory operations complete in order, so `s_waitcnt vmcnt(63)` two groups (>= 68 operations) later proves
//     its requests have landed without draining its stores.  The DMA is issued from inline assembly: the compiler's own LDS-DMA
//     tracking would put a full wait in front of the first ring read after every request.
// Ring capacities: a group g computes while rows up to group g + HP_LA arrive; output rows 4 g .. 4 (g + HP_LA) + 3 tap source rows
// floor(4 g s) .. floor((4 (g + HP_LA) + 3) s) + 1, at most floor(11 s) + 3 of them for a vertical scale s = (Hs - 1) / (H - 1) < 1/2,
// 1/8, 1/32: 8, 4, 3 -- the a2 and E3 rings (8 and 4 rows) are exactly full, with no slack (ADVICE r04 corrected the count: the r04
// comment said 7, 3, 2).  It is safe because the barrier at the top of a group precedes the next request; hp_ring_rows_ok() below
// re-derives the bound on the host for the launch's H and refuses a shape / look-ahead that would alias a live row.
// LDS 50.8 KB with the x1 slots (r03: 52.9), three workgroups per CU as before; whole-pixel stores (see above).
constexpr int HP_TILES = 2, HP_BW = 32 * HP_TILES;
constexpr int HP_A2C = 34, HP_A2R = 8, HP_A2S = HP_A2C * 4;                   // a2 ring: rows x 16-byte slots (136 per row)
constexpr int HP_NT3 = 10, HP_NT4 = 5, HP_ER = 4;                             // raw E3 / E4 rings: rows x band columns x ESTRIDE
constexpr int HP_E3S = HP_NT3 * (ESTRIDE / 4), HP_E4S = HP_NT4 * (ESTRIDE / 4);     // 16-byte slots per ring row: 170, 85
constexpr int HP_LA = 2;                                                      // row groups of look-ahead

// rows of a source map (height Hs) live at once while the head walks an image of height H: lo(g) .. hi(g + HP_LA) over all groups g
inline bool hp_ring_rows_ok(int H, int Hs, int ring_rows)
{
    const float s = (float)(Hs - 1) / (float)(H - 1);
    for (int g = 0; g < H / 4; ++g) {
        const int lo = (int)(s * (float)(4 * g));
        const int last = std::min(H / 4 - 1, g + HP_LA);
        const int r = (int)(s * (float)(4 * last + 3));
        const int hi = r + (r < Hs - 1 ? 1 : 0);
        if (hi - lo + 1 > ring_rows) return false;
    }
    return true;
}

// 16 bytes per active lane from global memory to LDS at lds_wave_base + 16 * lane (wave-uniform base), no VGPR in between.
template <bool NT = false>
__device__ __forceinline__ void hp_dma16(const void* gsrc, const void* lds_wave_base)
{
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(lds_wave_base));   // low half of a shared-aperture address = LDS offset
    unsigned keep;      // M0 is the compiler's (reserved register): saved and put back around the instruction that reads it
    if (NT)     // once-read stream: non-temporal (MI355X_MICROARCH.md, nt-weights: issued -> landed -18 %)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(m0v), "v"(gsrc) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(m0v), "v"(gsrc) : "memory");
}

// The same with a wave-uniform base pointer (scalar registers) and a 32-bit per-lane BYTE offset: no 64-bit vector address arithmetic.
__device__ __forceinline__ void hp_dma16s(const void* sbase, unsigned voff_bytes, const void* lds_wave_base)
{
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(lds_wave_base));
    const unsigned long long sb = (unsigned long long)reinterpret_cast<uintptr_t>(sbase);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
    const unsigned long long sbu = ((unsigned long long)hi << 32) | lo;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(m0v), "v"(voff_bytes), "s"(sbu) : "memory");
}

#ifdef KPB_STAMPS
// One-off instrumentation (scripts/head_stamps.sh builds with -DKPB_STAMPS; never in the product build): lane 0 of sampled waves leaves
// s_memtime at phase boundaries of ONE row group in the middle of its walk.  [wave slot][32] shader-clock stamps.
__device__ unsigned long long kpb_stamp_buf[256 * 32];
__device__ unsigned kpb_stamp_slots;
#define KPB_STAMP(i) do { if (stamp_on) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                          if (lane == 0) kpb_stamp_buf[stamp_slot * 32 + (i)] = t_; } } while (0)
extern "C" __attribute__((visibility("default"))) int kpb_debug_head_stamps(unsigned long long* out_host, unsigned* slots_host)
{
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(kpb_stamp_buf), sizeof(unsigned long long) * 256 * 32) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(slots_host, HIP_SYMBOL(kpb_stamp_slots), sizeof(unsigned)) != hipSuccess) return -1;
    const unsigned zero = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(kpb_stamp_slots), &zero, sizeof(unsigned)) == hipSuccess ? 0 : -1;
}
#else
#define KPB_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(256, 3) void alike_head_f16p(HybArgs a, const uint4* __restrict__ wh16, int groups_per_wg)
{
    __shared__ __attribute__((aligned(16))) uint4 Bh[2][256];
    __shared__ __attribute__((aligned(16))) uint4 A1h[2][64];         // agg1 as MFMA A fragments [hi / lo][lane] (HybArgs::a1h16)
    __shared__ __attribute__((aligned(16))) float Ws[2 * 16];
    __shared__ __attribute__((aligned(16))) float4 A2r[HP_A2R * HP_A2S];
    __shared__ __attribute__((aligned(16))) float E3r[HP_ER][HP_NT3 * ESTRIDE];
    __shared__ __attribute__((aligned(16))) float E4r[HP_ER][HP_NT4 * ESTRIDE];
    __shared__ __attribute__((aligned(16))) float4 X1r[4][2][64];     // per wave: the x1 pixels of two tiles in flight ([half][pixel] x 16 bytes)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int ef = cm_exp_of(fmaxf(__uint_as_float(a.amax_x1[b]) * a.l1_agg1, __uint_as_float(a.amax_a2[b])));
    // wave-uniform values the whole walk needs are pinned in scalar registers (uf / ui): computed on the vector ALU they would each
    // hold a VGPR for the life of the workgroup (the first build of this kernel spilled 70)
    auto uf = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    auto ui = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const float scf = uf(cm_scale_of(ef)), unf = cm_unscale_of(ef);
    const float cacc = uf(scf * a.ws_h), unacc = uf(unf * a.inv_ws_h);
    Bh[0][tid] = wh16[tid];
    Bh[1][tid] = wh16[256 + tid];
    if (tid < 128) A1h[tid >> 6][tid & 63] = a.a1h16[tid];
    // score weights in the lane's k order: group 0 = the agg1 accumulator's channels {0..3, 8..11} + 4 h, group 1 = 16 + 8 h + (s & 7)
    if (tid < 32) { const int s = tid & 15, hh = tid >> 4; Ws[tid] = a.wsc[s < 8 ? (s < 4 ? s : s + 4) + 4 * hh : 16 + 8 * hh + (s & 7)] * unf; }
    // x1 >= 0 is split at the image's own scale (amax_x1), the product is brought to the features' units on the accumulator
    const int ex1 = cm_exp_of(__uint_as_float(a.amax_x1[b]));
    const float sx1 = uf(cm_scale_of(ex1)), una1 = uf((cm_unscale_of(ex1) * scf) * a.inv_wa);      // (the two data-dependent powers of two first: their product is moderate)

    const int H2 = a.H / 2, W2 = a.W / 2, H8 = a.H / 8, W8 = a.W / 8, H32 = a.H / 32, W32 = a.W / 32;
    const float* a2 = a.a2 + (size_t)b * H2 * W2 * 16;
    const float* E3 = a.E3 + (size_t)b * H8 * W8 * ESTRIDE;
    const float* E4 = a.E4 + (size_t)b * H32 * W32 * ESTRIDE;
    const float sy2 = uf((float)(H2 - 1) / (float)(a.H - 1)), sx2 = uf((float)(W2 - 1) / (float)(a.W - 1));
    const float sy8 = uf((float)(H8 - 1) / (float)(a.H - 1)), sx8 = uf((float)(W8 - 1) / (float)(a.W - 1));
    const float sy32 = uf((float)(H32 - 1) / (float)(a.H - 1)), sx32 = uf((float)(W32 - 1) / (float)(a.W - 1));
    const int tiles_per_row = a.W / 32, bands = (tiles_per_row + HP_TILES - 1) / HP_TILES, ngroups = a.H / 4;
    const int band = (int)blockIdx.x % bands, seg = (int)blockIdx.x / bands;
    const int g0 = seg * groups_per_wg, g1 = min(g0 + groups_per_wg, ngroups);
    const int xs = band * HP_BW;
    const int ntile = min(HP_TILES, tiles_per_row - xs / 32);
    const int c_lo = ui((int)(sx2 * (float)xs)), tb3 = ui((int)(sx8 * (float)xs)), tb4 = ui((int)(sx32 * (float)xs));

    // rows [lo(G), hi(G)] of a source map that row group G reads (align_corners taps of its rows 4 G .. 4 G + 3)
    auto hi_row = [&](float s, int G, int Hs) { const int r = ui((int)(s * (float)(4 * G + 3))); return r + (r < Hs - 1 ? 1 : 0); };
    int a2next = ui((int)(sy2 * (float)(4 * g0))), e3next = ui((int)(sy8 * (float)(4 * g0))), e4next = ui((int)(sy32 * (float)(4 * g0)));
    // per-lane byte offsets of the 1 KB request units inside a source row, once per walk (8 registers): LDS position q = 64 j + lane of
    // an a2 ring row holds slot (q & 3) ^ swizzle of pixel q >> 2 (bank spread, see above); of an E ring row, 16-byte piece q % 17 of
    // band column q / 17 (clamped to the map's last column)
    unsigned offA[(HP_A2S + 63) / 64], off3[(HP_E3S + 63) / 64], off4[(HP_E4S + 63) / 64];
#pragma unroll
    for (int j = 0; j < (HP_A2S + 63) / 64; ++j) {
        const int q = min(64 * j + lane, HP_A2S - 1), col = q >> 2, sl = (q & 3) ^ ((col >> 2) & 3), gc = min(c_lo + col, W2 - 1);
        offA[j] = (unsigned)(gc * 16 + 4 * sl) * 4u;
    }
#pragma unroll
    for (int j = 0; j < (HP_E3S + 63) / 64; ++j) {
        const int q = min(64 * j + lane, HP_E3S - 1), t = q / (ESTRIDE / 4), quad = q - t * (ESTRIDE / 4);
        off3[j] = (unsigned)(min(tb3 + t, W8 - 1) * ESTRIDE + 4 * quad) * 4u;
    }
#pragma unroll
    for (int j = 0; j < (HP_E4S + 63) / 64; ++j) {
        const int q = min(64 * j + lane, HP_E4S - 1), t = q / (ESTRIDE / 4), quad = q - t * (ESTRIDE / 4);
        off4[j] = (unsigned)(min(tb4 + t, W32 - 1) * ESTRIDE + 4 * quad) * 4u;
    }
    auto request = [&](const int G) {      // every row group G reads that has not been requested yet; 1 KB units dealt over the waves
        int u = 0;                          // (r04 stamps: with per-unit 64-bit vector addresses this took 2.6 k of a group's 27.8 k cycles)
        const int h2 = hi_row(sy2, G, H2), h3 = hi_row(sy8, G, H8), h4 = hi_row(sy32, G, H32);
        for (int r = a2next; r <= h2; ++r) {
            const float* row = a2 + (size_t)r * W2 * 16;
#pragma unroll
            for (int j = 0; j < (HP_A2S + 63) / 64; ++j, ++u)
                if ((u & 3) == wv && 64 * j + lane < HP_A2S) hp_dma16s(row, offA[j], &A2r[(r & (HP_A2R - 1)) * HP_A2S + 64 * j]);
        }
        a2next = max(a2next, h2 + 1);
        for (int r = e3next; r <= h3; ++r) {
            const float* row = E3 + (size_t)r * W8 * ESTRIDE;
#pragma unroll
            for (int j = 0; j < (HP_E3S + 63) / 64; ++j, ++u)
                if ((u & 3) == wv && 64 * j + lane < HP_E3S) hp_dma16s(row, off3[j], &E3r[r & (HP_ER - 1)][64 * j * 4]);
        }
        e3next = max(e3next, h3 + 1);
        for (int r = e4next; r <= h4; ++r) {
            const float* row = E4 + (size_t)r * W32 * ESTRIDE;
#pragma unroll
            for (int j = 0; j < (HP_E4S + 63) / 64; ++j, ++u)
                if ((u & 3) == wv && 64 * j + lane < HP_E4S) hp_dma16s(row, off4[j], &E4r[r & (HP_ER - 1)][64 * j * 4]);
        }
        e4next = max(e4next, h4 + 1);
    };

    f32x16 pend = {0}, pendB = {0};     // the previous tile, whole pixels: pend = pixels R(r), pendB = pixels R(r) + 4
    float* pend_d = nullptr;
    // x1 arrives by LDS-DMA too, TWO tiles ahead, one 1 KB instruction per tile and wave: lane (p, h) brings half h of pixel p to
    // slot 32 h + p.  With no load left that returns to registers, the only waits of the walk are `s_waitcnt vmcnt(63)` at tile
    // starts: a wave keeps up to 63 vector-memory operations (two tiles of stores) in flight instead of the one tile the in-order
    // return of a register load one tile ahead allowed.
    auto fetch = [&](int y, int t, int slot) {
        const size_t pix = (size_t)b * a.H * a.W + (size_t)y * a.W + (xs + 32 * t + p);
        hp_dma16<true>(a.x1 + pix * 8 + 4 * h, &X1r[wv][slot][0]);        // read once: non-temporal (-0.4 %, profiles/r04_ab_knobs.txt)
    };
    // tile k of this wave's walk = (group g0 + k / ntile, tile k % ntile)
    if (g0 < g1) {
        fetch(4 * g0 + wv, 0, 0);
        if (ntile == 2) fetch(4 * g0 + wv, 1, 1);
        else if (g0 + 1 < g1) fetch(4 * (g0 + 1) + wv, 0, 1);
    }
    for (int G = g0; G < min(g0 + HP_LA, g1); ++G) request(G);
    int ktile = 0;
#ifdef KPB_STAMPS
    bool stamp_on = false;
    unsigned stamp_slot = 0;
    const bool stamp_wg = (blockIdx.x % 61) == 7 && (blockIdx.y % 16) == 3 && ntile == 2;
#endif

    // per group and wave: the y taps of the raw coarse rows (strip_store's arithmetic, applied where the value is read)
    int U3 = 0, D3 = 0, U4 = 0, D4 = 0;        // float offsets of the two tapped ring rows inside E3r / E4r (offsets, not pointers: the reads stay ds_read)
    float hy3 = 0.f, ly3 = 0.f, hy4 = 0.f, ly4 = 0.f, hy3c = 0.f, ly3c = 0.f, hy4c = 0.f, ly4c = 0.f;

    // The x taps of the coarse maps depend on the tile position only, not on the row group: the hat weights and their fractions are
    // taken once per walk for both tiles (14 registers; the a2 tap offsets and the rest stay per tile -- hoisting everything spilled)
    float w3t[HP_TILES][3], w4t[HP_TILES][2], lx3t[HP_TILES], lx4t[HP_TILES];
#pragma unroll
    for (int t = 0; t < HP_TILES; ++t) {
        const int x0 = xs + 32 * t, x = x0 + p;
        const float fx3 = sx8 * (float)x, fx4 = sx32 * (float)x;
        const int rb3 = (int)(sx8 * (float)x0) - tb3, rb4 = (int)(sx32 * (float)x0) - tb4;
        const int t3 = (int)fx3 - tb3 - rb3, t4 = (int)fx4 - tb4 - rb4;
        lx3t[t] = fx3 - (float)(int)fx3; lx4t[t] = fx4 - (float)(int)fx4;
#pragma unroll
        for (int s = 0; s < 3; ++s) { const int kk = 2 * s + h; w3t[t][s] = kk == t3 ? 1.0f - lx3t[t] : (kk == t3 + 1 ? lx3t[t] : 0.0f); }
#pragma unroll
        for (int s = 0; s < 2; ++s) { const int kk = 2 * s + h; w4t[t][s] = kk == t4 ? 1.0f - lx4t[t] : (kk == t4 + 1 ? lx4t[t] : 0.0f); }
    }
    int xso = xs;       // the band's first column, made opaque once per group: everything derived from x is invariant along the walk, and
                        // hoisted out of it the tap offsets and hat weights of both tiles held ~40 registers (spills)
    auto body = [&](auto first_tag, const int y, const int t, const int ny, const int nt) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const int x0 = xso + 32 * t, x = x0 + p;
        const size_t pix = (size_t)b * a.H * a.W + (size_t)y * a.W + x;
        // this tile's x1 (requested two tiles = >= 65 operations ago) and every ring row of its group have landed once all but the
        // youngest 63 operations of the wave have completed; the first tiles of a walk have fewer behind them and wait for all
        // (vmcnt(40) and vmcnt(24) here cost nothing either -- profiles/r04_ab_knobs.txt: the stores of the tile before are acknowledged
        // within a tile's time, the walk is not waiting for the memory system)
        if (ktile < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
        KPB_STAMP(4 + 8 * t);
        const int slot = ktile & 1;
        const float4 x1lo = X1r[wv][slot][p], x1hi = X1r[wv][slot][32 + p];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot has been read: the request for two tiles ahead may overwrite it
        if (ny >= 0) fetch(ny, nt, slot);
        ++ktile;
        KPB_STAMP(5 + 8 * t);
        Up8Taps taps;
        {
            const float fy = sy2 * (float)y, fx = sx2 * (float)x;
            const int gy0 = (int)fy, gx0 = (int)fx;
            const int gy1 = gy0 + (gy0 < H2 - 1 ? 1 : 0);
            const int rx0 = gx0 - c_lo, rx1 = rx0 + (gx0 < W2 - 1 ? 1 : 0);
            taps.ly = fy - (float)gy0; taps.lx = fx - (float)gx0;
            const float4* r0 = A2r + (gy0 & (HP_A2R - 1)) * HP_A2S;
            const float4* r1 = A2r + (gy1 & (HP_A2R - 1)) * HP_A2S;
            const int s0 = (rx0 >> 2) & 3, s1 = (rx1 >> 2) & 3;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                taps.t[4 * q + 0] = r0[rx0 * 4 + ((2 * h + q) ^ s0)]; taps.t[4 * q + 1] = r0[rx1 * 4 + ((2 * h + q) ^ s1)];
                taps.t[4 * q + 2] = r1[rx0 * 4 + ((2 * h + q) ^ s0)]; taps.t[4 * q + 3] = r1[rx1 * 4 + ((2 * h + q) ^ s1)];
            }
        }
        if constexpr (!FIRST) {
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(pend[r], pend_d + ((r & 3) + 8 * (r >> 2)) * 64);
        }
        KPB_STAMP(6 + 8 * t);                   // taps read, first 16 stores of the previous tile issued
        int z = 0;                              // opaque zero: keeps the tile-invariant LDS reads inside the loop
        asm volatile("" : "+v"(z));
        const uint4* Bhz = &Bh[0][0] + z; const uint4* A1z = &A1h[0][0] + z; const float* Wsz = Ws + z;
        float f[16];
        {   // group 0: relu(agg1 . x1) (ALike.py:147) on the matrix pipe (r04): D[o][pixel] = A1[o][c] x1[pixel][c] as three split-f16 MFMAs
            // (K = 16, the upper eight are zero weights); lane (p, h) finds channels {0..3, 8..11} + 4 h of ITS pixel in registers 0..7.
            // r03 / early r04: 64 FMAs per lane against 16 weight reads from LDS -- 3.3 k of a tile's 11 k cycles were this phase.
            const float xs[8] = {x1lo.x * sx1, x1lo.y * sx1, x1lo.z * sx1, x1lo.w * sx1, x1hi.x * sx1, x1hi.y * sx1, x1hi.z * sx1, x1hi.w * sx1};
            h8v xh, xl;
            split8(xs, xh, xl);
            const h8v wh = __builtin_bit_cast(h8v, A1z[lane]), wl = __builtin_bit_cast(h8v, A1z[64 + lane]);
            f32x16 aa = {0};
            aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, aa, 0, 0, 0);
            aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, aa, 0, 0, 0);
            aa = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, aa, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = relu(aa[j] * una1);
        }
        up8ch_lerp4(taps, f + 8, scf);

        const float fx3 = sx8 * (float)x, fx4 = sx32 * (float)x;
        const int rb3 = (int)(sx8 * (float)x0) - tb3, rb4 = (int)(sx32 * (float)x0) - tb4;
        const int t3 = (int)fx3 - tb3 - rb3, t4 = (int)fx4 - tb4 - rb4;
        const float lx3 = t == 0 ? lx3t[0] : lx3t[1], lx4 = t == 0 ? lx4t[0] : lx4t[1];
        const float *u3 = &E3r[0][0] + (U3 + rb3 * ESTRIDE + z), *d3 = &E3r[0][0] + (D3 + rb3 * ESTRIDE + z);
        const float *u4 = &E4r[0][0] + (U4 + rb4 * ESTRIDE + z), *d4 = &E4r[0][0] + (D4 + rb4 * ESTRIDE + z);
        // the y-interpolated coarse row at (strip row k, channel c), in the accumulator's units (channel 64: the score share, unscaled)
        auto e3 = [&](int k, int c) { return fmaf(ly3c, d3[k * ESTRIDE + c], hy3c * u3[k * ESTRIDE + c]); };     // (one fused step: r04)
        auto e4 = [&](int k, int c) { return fmaf(ly4c, d4[k * ESTRIDE + c], hy4c * u4[k * ESTRIDE + c]); };

        float sc = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = *reinterpret_cast<const float4*>(&Wsz[h * 16 + 4 * q]);
            sc = fmaf(f[4 * q], w.x, sc); sc = fmaf(f[4 * q + 1], w.y, sc); sc = fmaf(f[4 * q + 2], w.z, sc); sc = fmaf(f[4 * q + 3], w.w, sc);
        }
        sc = kpb_sum32(sc);
        sc += (1.0f - lx3) * (hy3 * u3[t3 * ESTRIDE + 64] + ly3 * d3[t3 * ESTRIDE + 64]) + lx3 * (hy3 * u3[(t3 + 1) * ESTRIDE + 64] + ly3 * d3[(t3 + 1) * ESTRIDE + 64]);
        sc += (1.0f - lx4) * (hy4 * u4[t4 * ESTRIDE + 64] + ly4 * d4[t4 * ESTRIDE + 64]) + lx4 * (hy4 * u4[(t4 + 1) * ESTRIDE + 64] + ly4 * d4[(t4 + 1) * ESTRIDE + 64]);
        if (h == 0) a.score[pix] = __fdiv_rn(1.0f, 1.0f + expf(-sc));
        KPB_STAMP(7 + 8 * t);                   // features, score

        float* d = a.desc + ((size_t)b * a.H * a.W + (size_t)y * a.W + x0) * 64;
        f32x16 acc0 = {0}, acc1 = {0};
        h8v ahi[2], alo[2];
        split8(f, ahi[0], alo[0]);
        split8(f + 8, ahi[1], alo[1]);
        float w3[3], w4[2];
#pragma unroll
        for (int s = 0; s < 3; ++s) w3[s] = t == 0 ? w3t[0][s] : w3t[1][s];
#pragma unroll
        for (int s = 0; s < 2; ++s) w4[s] = t == 0 ? w4t[0][s] : w4t[1][s];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const h8v b0h = __builtin_bit_cast(h8v, Bhz[((kb * 2 + 0) * 2 + h) * 32 + p]), b0l = __builtin_bit_cast(h8v, Bhz[256 + ((kb * 2 + 0) * 2 + h) * 32 + p]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[kb], b0h, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[kb], b0l, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[kb], b0h, acc0, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3[s], e3(2 * s + h, p), acc0, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 2; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[s], e4(2 * s + h, p), acc0, 0, 0, 0);
        KPB_STAMP(8 + 8 * t);                   // split, first MFMA chain issued
        h8v b1h[2], b1l[2];
        float t3b[3], t4b[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { b1h[kb] = __builtin_bit_cast(h8v, Bhz[((kb * 2 + 1) * 2 + h) * 32 + p]); b1l[kb] = __builtin_bit_cast(h8v, Bhz[256 + ((kb * 2 + 1) * 2 + h) * 32 + p]); }
#pragma unroll
        for (int s = 0; s < 3; ++s) t3b[s] = e3(2 * s + h, 32 + p);
#pragma unroll
        for (int s = 0; s < 2; ++s) t4b[s] = e4(2 * s + h, 32 + p);
        int r = 0;
        float* pb = pend_d + 4 * 64;
#define HP_ST1() { if constexpr (!FIRST) __builtin_nontemporal_store(pendB[r], pb + ((r & 3) + 8 * (r >> 2)) * 64); ++r; }
#define HP_ST2() { HP_ST1() HP_ST1() }
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[0], b1h[0], acc1, 0, 0, 0); HP_ST2()
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[0], b1l[0], acc1, 0, 0, 0); HP_ST2()
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[0], b1h[0], acc1, 0, 0, 0); HP_ST2()
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[1], b1h[1], acc1, 0, 0, 0); HP_ST2()
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[1], b1l[1], acc1, 0, 0, 0); HP_ST2()
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[1], b1h[1], acc1, 0, 0, 0); HP_ST2()
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3[0], t3b[0], acc1, 0, 0, 0); HP_ST2()
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3[1], t3b[1], acc1, 0, 0, 0); HP_ST2()
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w3[2], t3b[2], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[0], t4b[0], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[1], t4b[1], acc1, 0, 0, 0);
#undef HP_ST2
#undef HP_ST1
        if constexpr (!FIRST) {
            __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x040, 2, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
        KPB_STAMP(9 + 8 * t);                   // second MFMA chain + 16 interleaved stores issued
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[q] * unacc), __float_as_uint(acc1[q] * unacc), false, false);
            pend[q] = __uint_as_float(sw[0]); pendB[q] = __uint_as_float(sw[1]);
        }
        pend_d = d + lane;
        KPB_STAMP(10 + 8 * t);                  // accumulators read, scaled, swapped
    };

    for (int g = g0; g < g1; ++g) {
#ifdef KPB_STAMPS
        stamp_on = false;
        if (stamp_wg && g == g0 + 10) {       // one group in the middle of the walk
            unsigned sl = 0;
            if (lane == 0) sl = atomicAdd(&kpb_stamp_slots, 1u);
            stamp_slot = __builtin_amdgcn_readfirstlane(sl);
            stamp_on = stamp_slot < 256;
        }
#endif
        KPB_STAMP(0);
        // this wave's ring requests for group g (two groups = at least two tiles old) have landed: same rule as a tile's x1
        if (ktile < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
        KPB_STAMP(1);
        __syncthreads();        // every wave has left group g - 1, every request for group g has landed
        KPB_STAMP(2);
        if (g + HP_LA < g1) request(g + HP_LA);
        KPB_STAMP(3);
        const int y = 4 * g + wv;
        {
            const float fy3 = sy8 * (float)y, fy4 = sy32 * (float)y;
            const int y03 = (int)fy3, y04 = (int)fy4;
            const int y13 = y03 + (y03 < H8 - 1 ? 1 : 0), y14 = y04 + (y04 < H32 - 1 ? 1 : 0);
            ly3 = uf(fy3 - (float)y03); hy3 = uf(1.0f - ly3); ly4 = uf(fy4 - (float)y04); hy4 = uf(1.0f - ly4);
            hy3c = uf(hy3 * cacc); ly3c = uf(ly3 * cacc); hy4c = uf(hy4 * cacc); ly4c = uf(ly4 * cacc);
            U3 = ui((y03 & (HP_ER - 1)) * (HP_NT3 * ESTRIDE)); D3 = ui((y13 & (HP_ER - 1)) * (HP_NT3 * ESTRIDE));
            U4 = ui((y04 & (HP_ER - 1)) * (HP_NT4 * ESTRIDE)); D4 = ui((y14 & (HP_ER - 1)) * (HP_NT4 * ESTRIDE));
        }
        asm volatile("" : "+s"(xso));
        if (ntile == 2) {       // the tile two ahead is the same tile of the next group
            const int ny = g + 1 < g1 ? y + 4 : -1;
            if (g == g0) body(std::true_type{}, y, 0, ny, 0);
            else body(std::false_type{}, y, 0, ny, 0);
            body(std::false_type{}, y, 1, ny, 1);
        } else {                // one-tile band: two groups ahead
            const int ny = g + 2 < g1 ? y + 8 : -1;
            if (g == g0) body(std::true_type{}, y, 0, ny, 0);
            else body(std::false_type{}, y, 0, ny, 0);
        }
        KPB_STAMP(20);
    }
    if (g0 < g1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(pend[r], pend_d + ((r & 3) + 8 * (r >> 2)) * 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(pendB[r], pend_d + ((r & 3) + 8 * (r >> 2) + 4) * 64);
    }
}

// ------------------------------------------------------------------------------------------------ score, linear form
// convhead2 is a bias-free 1x1 convolution and nn.Upsample(bilinear) is linear, so the score row of the head commutes
// with the three upsamplings (ALike.py:151-162):
//     w . [f1 | up2(a2) | up8(a3) | up32(a4)]  =  w1 . f1  +  up2(w2 . a2)  +  up8(w3 . a3)  +  up32(w4 . a4)
// The scalar maps S_g = w_g . a_g come out of the aggregation kernels at 1/4, 1/64 and 1/1024 of the pixels; when no
// dense descriptor map is wanted the full-resolution work is one 8 -> 16 product and twelve scalar taps per pixel.
// (The same rewrite of the 64 descriptor rows was measured and lost to alike_head: interpolating three projected
// 64-channel maps costs more vector/LDS work than the 64-deep MFMA it saves.)
struct LinArgs {
    const float* x1;                       // [B][H][W][8]
    const float* S2; const float* S3; const float* S4;   // [B][H/2][W/2], ...
    const float* agg1;                     // [8][16]
    const float* wsc;                      // [64]      score row; entries 0..15 belong to f1
    float* score;
    int H, W;
    float sy2, sx2, sy3, sx3, sy4, sx4;    // (Hs - 1) / (H - 1), (Ws - 1) / (W - 1) of the three coarse score maps: the same float divisions, taken once on the host
};

__device__ __forceinline__ float lerp_scalar(const float* m, int Hs, int Ws, float sy, float sx, int y, int x)
{
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    return hy * (hx * m[(size_t)y0 * Ws + x0] + lx * m[(size_t)y0 * Ws + x1]) + ly * (hx * m[(size_t)y1 * Ws + x0] + lx * m[(size_t)y1 * Ws + x1]);
}

// one thread per pixel
__global__ __launch_bounds__(256) void alike_score_lin(LinArgs a)
{
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int P = a.H * a.W;
    if (pix >= P) return;
    const int y = pix / a.W, x = pix - y * a.W;
    const float* px = a.x1 + ((size_t)b * P + pix) * 8;
    const float4 lo = *reinterpret_cast<const float4*>(px), hi = *reinterpret_cast<const float4*>(px + 4);
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    // r05: the kernel is bound by vector issue (9.5e8 instructions per launch: more than block 1).  agg1's 128 FMAs go two output channels at a
    // time (v_pk_fma_f32 against a pair of scalar weights: the same fused operations in the same order, 64 instructions), the six scale
    // divisions -- wave-uniform, 8 instructions each -- come from the host.
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (f2v){0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f2v w = {a.agg1[c * 16 + 2 * j], a.agg1[c * 16 + 2 * j + 1]};
            f[j] = __builtin_elementwise_fma((f2v){v[c], v[c]}, w, f[j]);
        }
    float sc = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sc = fmaf(relu(f[j >> 1][j & 1]), a.wsc[j], sc);
    const int H2 = a.H / 2, W2 = a.W / 2, H8 = a.H / 8, W8 = a.W / 8, H32 = a.H / 32, W32 = a.W / 32;
    sc += lerp_scalar(a.S2 + (size_t)b * H2 * W2, H2, W2, a.sy2, a.sx2, y, x);
    sc += lerp_scalar(a.S3 + (size_t)b * H8 * W8, H8, W8, a.sy3, a.sx3, y, x);
    sc += lerp_scalar(a.S4 + (size_t)b * H32 * W32, H32, W32, a.sy4, a.sx4, y, x);
    a.score[(size_t)b * P + pix] = __fdiv_rn(1.0f, 1.0f + expf(-sc));   // torch.sigmoid (ALike.py:162)
}

// ------------------------------------------------------------------------------------------------ desc_at
struct DescAtArgs {
    HeadArgs h;
    const float* pts; const int* n; float* out;
    int pts_cols, max_n;
    int kpw;                // keypoints a wave takes in turn (1 for a handful of images: the drop-in path wants the parallelism, not the reuse)
};

// one wave per keypoint; lane = feature channel while sampling, = output channel for the mat-vec.
// Equals kpb_sample() on the dense map because the 1x1 head and the bilinear taps are both linear.
// r05: a workgroup takes a.kpw keypoints per wave and keeps the 64 x 64 head matrix in LDS, transposed ([output][feature], 68-float rows:
// sixteen 16-byte reads per lane, conflict-free) -- one wave per keypoint fetched its 16 KB from L2 for every keypoint (8.4 GB per launch of
// 512 000 keypoints: 0.70 ms, 74 % of the wave-cycles waiting).  The products are the same fused chain over the features in ascending order.
constexpr int DA_PITCH = 68;
__global__ __launch_bounds__(256) void alike_desc_at(DescAtArgs a)
{
    __shared__ __attribute__((aligned(16))) float wh[64 * DA_PITCH];
    __shared__ __attribute__((aligned(16))) float fs[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = a.n ? min(a.n[b], a.max_n) : a.max_n;
    const int i0 = (blockIdx.x * 4 + wv) * a.kpw;
    if (blockIdx.x * 4 * a.kpw >= n) return;              // workgroup-uniform: nothing to do for any wave
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = threadIdx.x + 256 * k;                // whT[c][o], o fastest
        wh[(i & 63) * DA_PITCH + (i >> 6)] = a.h.whT[i];
    }
    __syncthreads();
    const int H = a.h.H, W = a.h.W;
    // this lane's feature channel c = lane: group g = c / 16 decides the formula
    const int g = lane >> 4, j = lane & 15;
    for (int kk = 0; kk < a.kpw; ++kk) {
        const int i = i0 + kk;
        if (i >= n) break;                                   // wave-uniform
        const float* pt = a.pts + ((size_t)b * a.max_n + i) * a.pts_cols;
        const float gx = (pt[0] - 0.5f) * 2.0f, gy = (pt[1] - 0.5f) * 2.0f;   // matcher.py:221-222
        const float x = (gx + 1.0f) * ((float)(W - 1) / 2.0f), y = (gy + 1.0f) * ((float)(H - 1) / 2.0f);
        const float xw = floorf(x), yn = floorf(y);
        const float w = x - xw, e = 1.0f - w, nn = y - yn, s = 1.0f - nn;
        const float cw[4] = {s * e, s * w, nn * e, nn * w};
        const int x0 = (int)xw, y0 = (int)yn;
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int tx = x0 + (t & 1), ty = y0 + (t >> 1);
            if (tx < 0 || tx >= W || ty < 0 || ty >= H) continue;   // grid_sample zero padding
            float v;
            if (g == 0) {
                const float* px = a.h.x1 + ((size_t)b * H * W + (size_t)ty * W + tx) * 8;
                v = 0.0f;
#pragma unroll
                for (int c = 0; c < 8; ++c) v = fmaf(px[c], a.h.agg1[c * 16 + j], v);
                v = relu(v);
            } else {
                const int sh = g == 1 ? 2 : (g == 2 ? 8 : 32);
                const int Hs = H / sh, Ws = W / sh;
                const float* m = (g == 1 ? a.h.a2 : (g == 2 ? a.h.a3 : a.h.a4)) + (size_t)b * Hs * Ws * 16;
                const float fy = ((float)(Hs - 1) / (float)(H - 1)) * (float)ty, fx = ((float)(Ws - 1) / (float)(W - 1)) * (float)tx;
                const int sy0 = (int)fy, sx0 = (int)fx;
                const int sy1 = sy0 + (sy0 < Hs - 1 ? 1 : 0), sx1 = sx0 + (sx0 < Ws - 1 ? 1 : 0);
                const float ly = fy - (float)sy0, lx = fx - (float)sx0, hy = 1.0f - ly, hx = 1.0f - lx;
                v = hy * (hx * m[((size_t)sy0 * Ws + sx0) * 16 + j] + lx * m[((size_t)sy0 * Ws + sx1) * 16 + j]) +
                    ly * (hx * m[((size_t)sy1 * Ws + sx0) * 16 + j] + lx * m[((size_t)sy1 * Ws + sx1) * 16 + j]);
            }
            acc = fmaf(cw[t], v, acc);
        }
        __builtin_amdgcn_wave_barrier();                     // the previous keypoint's reads of fs are done (a wave's LDS operations execute in order)
        fs[wv][lane] = acc;
        __builtin_amdgcn_wave_barrier();
        float o = 0.0f;
        const float4* wr = reinterpret_cast<const float4*>(wh + lane * DA_PITCH);
        const float4* fr = reinterpret_cast<const float4*>(fs[wv]);
#pragma unroll
        for (int c4 = 0; c4 < 16; ++c4) {
            const float4 f4 = fr[c4], w4 = wr[c4];
            o = fmaf(f4.x, w4.x, o); o = fmaf(f4.y, w4.y, o); o = fmaf(f4.z, w4.z, o); o = fmaf(f4.w, w4.w, o);
        }
        a.out[((size_t)b * a.max_n + i) * 64 + lane] = o;
    }
}

}  // namespace

// ================================================================================================ host side
namespace {

struct AlikeNet : kpb_net {
    float *p1 = nullptr, *x1 = nullptr, *t2 = nullptr, *x2 = nullptr, *a2 = nullptr, *t3 = nullptr, *x3 = nullptr, *a3 = nullptr,
          *t4 = nullptr, *x4 = nullptr, *a4 = nullptr, *S2 = nullptr, *S3 = nullptr, *S4 = nullptr, *E3 = nullptr, *E4 = nullptr;
    // split-f16 form: reciprocal power-of-two weight scales of the custom packs, and the L1 norms / bias maxima behind the
    // bounds the fused kernels scale their intermediate maps by (conv_mfma.h, cm_scale_of)
    std::map<std::string, float> k;
    HeadArgs head_args(float* score, float* desc)
    {
        HeadArgs h;
        h.x1 = x1; h.a2 = a2; h.a3 = a3; h.a4 = a4;
        h.agg1 = wp("agg1.w"); h.whT = wp("head.wT"); h.wsc = wp("head.ws");
        h.score = score; h.desc = desc; h.H = H; h.W = W;
        return h;
    }
    int forward(const float* img_dev, int batch, int H_, int W_, float* score_out_dev, float* desc_out_dev) override;
    int desc_at(const float* pts_dev, int pts_cols, int max_n, const int32_t* n_dev, float* out_dev) override;
};

// OIHW [co][ci][3][3] -> [tap][ci][co]
void repack3x3(const float* w, int co, int ci, std::vector<float>& out)
{
    out.resize((size_t)9 * ci * co);
    for (int o = 0; o < co; ++o)
        for (int c = 0; c < ci; ++c)
            for (int k = 0; k < 9; ++k) out[((size_t)k * ci + c) * co + o] = w[((size_t)o * ci + c) * 9 + k];
}
// [co][ci] -> [ci][co]
void transpose(const float* w, int co, int ci, std::vector<float>& out)
{
    out.resize((size_t)ci * co);
    for (int o = 0; o < co; ++o)
        for (int c = 0; c < ci; ++c) out[(size_t)c * co + o] = w[(size_t)o * ci + c];
}
// max over the `co` output channels of the sum of |w| over the channel's `per` weights, and max |b|
float l1_rows(const float* w, int co, int per)
{
    float m = 0.0f;
    for (int o = 0; o < co; ++o) {
        double s = 0.0;
        for (int i = 0; i < per; ++i) s += std::fabs((double)w[(size_t)o * per + i]);
        m = std::max(m, (float)(s * (1.0 + 1e-6)));
    }
    return m;
}
float max_abs(const float* b, int n)
{
    float m = 0.0f;
    for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(b[i]));
    return m;
}

template <int CIN, int COUT, int POOL, bool RES, int CDS, int RPOOL, bool DSOUT = false, int TW = 32>
void launch_conv(kpb_ctx* ctx, const char* name, hipStream_t st, const ConvArgs& a, int B)
{
    constexpr int G = COUT / 16, TH = (64 / TW) * (4 / G);
    KPB_LAUNCH(ctx, name, (conv3x3_k<CIN, COUT, POOL, RES, CDS, RPOOL, true, DSOUT, TW>), dim3(cdiv(a.W, TW), cdiv(a.H, TH), B), dim3(256), 0, st, a);
}

// max_pool2d(x, 4, 4) of an NHWC map (ALike.py:143, block 4's input): one thread per pooled pixel and channel quad.  At batch size block 4's conv1 reads this
// (19.7 MB per 512 images) instead of max-pooling x3 itself while it stages: sixteen loads per staged value, by four workgroups per tile (r06: 0.26 -> ms).
__global__ __launch_bounds__(256) void maxpool4_nhwc(const float* __restrict__ in, float* __restrict__ out, int Ho, int Wo, int C)
{
    const int C4 = C / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= (size_t)Ho * Wo * C4) return;
    const int c = (int)(i % C4) * 4;
    const size_t pix = i / C4;
    const int y = (int)(pix / Wo), x = (int)(pix - (size_t)y * Wo);
    const float* s = in + ((b * 4 * Ho + 4 * y) * (size_t)(4 * Wo) + 4 * x) * C + c;
    float4 m = *reinterpret_cast<const float4*>(s);
#pragma unroll
    for (int py = 0; py < 4; ++py)
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            if (py == 0 && px == 0) continue;
            const float4 v = *reinterpret_cast<const float4*>(s + ((size_t)py * 4 * Wo + px) * C);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    *reinterpret_cast<float4*>(out + ((b * Ho + y) * (size_t)Wo + x) * C + c) = m;
}

int AlikeNet::forward(const float* img_dev, int batch, int H_, int W_, float* score_out_dev, float* desc_out_dev)
{
    if ((H_ % 32) || (W_ % 32))
        return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_forward: ALIKE needs H and W multiples of 32 (got %dx%d)", H_, W_);
    const int H = H_, W = W_;
    const size_t P = (size_t)H * W, B = batch;
    const bool h16 = conv_mfma_use_h16();       // the split-f16 matrix form (default) or the strict fp32 kernels (KPB_FP32_MATRIX=1)
    const size_t n_x1 = B * P * 8, n_2 = B * (P / 4) * 16, n_3 = B * (P / 64) * 32, n_a3 = B * (P / 64) * 16,
                 n_4 = B * (P / 1024) * 64, n_a4 = B * (P / 1024) * 16;
    const size_t n_s = B * (P / 4 + P / 64 + P / 1024) + 64;
    const size_t n_e = desc_out_dev ? B * (P / 64 + P / 1024) * ESTRIDE : 0;
    const size_t n_p1 = B * (P / 4) * 8;
    const int nw1 = cdiv(W, B1_TW) * cdiv(H, B1H_TH) * 4, nw2 = cdiv(W / 2, 32) * cdiv(H / 2, 8) * 4;      // per-wave maxima of blocks 1 / 2
    const size_t n_rng = (2 * B + B * (nw1 + nw2) + 63) / 64 * 64;
    const size_t total = n_x1 + n_p1 + 3 * n_2 + 6 * n_3 + n_a3 + 5 * n_4 + n_4 / 2 + n_a4 + n_s + n_e + n_rng;
    if (int rc = kpb_reserve(ctx, act, total * sizeof(float))) return rc;
    float* p = static_cast<float*>(act.p);
    x1 = p; p += n_x1;
    p1 = p; p += n_p1;
    t2 = p; p += n_2; x2 = p; p += n_2; a2 = p; p += n_2;
    t3 = p; p += n_3; x3 = p; p += n_3; a3 = p; p += n_a3;
    t4 = p; p += n_4; x4 = p; p += n_4; a4 = p; p += n_a4;
    float* r3 = p; p += n_3;
    float* r4 = p; p += n_4;
    float* t3r3 = p; p += 2 * n_3;      // conv_mfma_h form of blocks 3 / 4: conv1's output and the identity branch side by side per pixel
    float* t4r4 = p; p += 2 * n_4;
    float* p2 = p; p += n_3 / 2;        // max_pool2d(x2, 4): [B][H/8][W/8][16]
    float* p3 = p; p += n_4 / 2;        // max_pool2d(x3, 4): [B][H/32][W/32][32] (batches: maxpool4_nhwc)
    S2 = p; p += B * (P / 4); S3 = p; p += B * (P / 64); S4 = p; p += B * (P / 1024) + 64;
    E3 = E4 = nullptr;
    if (desc_out_dev) { E3 = p; p += B * (P / 64) * ESTRIDE; E4 = p; p += B * (P / 1024) * ESTRIDE; }
    unsigned* amax_x1 = reinterpret_cast<unsigned*>(p);      // [B], [B]: per-image largest x1 / a2 value (float bits)
    unsigned* amax_a2 = amax_x1 + B;
    float* wmax_x1 = p + 2 * B;                                // [B][nw1], [B][nw2]: the per-wave maxima they are folded from
    float* wmax_a2 = wmax_x1 + B * nw1;
    p += n_rng;
    this->B = batch; this->H = H; this->W = W;
    hipStream_t st = ctx->stream;

    Block1Args b1{img_dev, x1, p1, wp("b1c1.w"), wp("b1c1.b"), wp("b1c2.w"), wp("b1c2.b"), H, W};
    ConvArgs c;
    if (h16) {
        Block1HArgs hb{b1, reinterpret_cast<const uint4*>(wp("b1c1.pairs")), reinterpret_cast<const uint4*>(wp("b1c2.pairs")),
                       k.at("b1c1.inv_ws"), k.at("b1c2.inv_ws"), k.at("b1c1.l1"), k.at("b1c1.bmax"), wmax_x1, 0};       // XCD-aware map: +4 % on block 1 (not bound by its halo re-reads), off
        KPB_LAUNCH(ctx, "alike_block1", alike_block1_h, dim3(cdiv(W, B1_TW), cdiv(H, B1H_TH), batch), dim3(256), 0, st, hb);
        KPB_LAUNCH(ctx, "amax_reduce", amax_reduce, dim3(batch), dim3(256), 0, st, wmax_x1, nw1, amax_x1);
        // block2 @ H/2 (ALike.py:139-140) + agg2, fused; it hands block 3 the 4 x 4 max-pool of its output (141)
        Block2Args b2{p1, x2, a2, S2, p2, reinterpret_cast<const uint4*>(wp("b2c1.h16")), reinterpret_cast<const uint4*>(wp("b2c2.h16")),
                      reinterpret_cast<const uint4*>(wp("agg2.h16")), wp("b2c1.b"), wp("b2c2.bsum"), wp("head.ws") + 16, H / 2, W / 2,
                      k.at("b2c1.inv_ws"), k.at("b2c2.inv_ws"), k.at("agg2.inv_ws"), k.at("b2c1.l1"), k.at("b2c1.bmax"), k.at("b2c2.l1"), k.at("b2ds.l1"),
                      k.at("b2c2.bsummax"), amax_x1, wmax_a2, 1};     // XCD-aware map: -5 % (profiles/r04_ab_knobs.txt)
        KPB_LAUNCH(ctx, "alike_block2", alike_block2, dim3(cdiv(W / 2, 32), cdiv(H / 2, 8), batch), dim3(256), 0, st, b2);
        KPB_LAUNCH(ctx, "amax_reduce", amax_reduce, dim3(batch), dim3(256), 0, st, wmax_a2, nw2, amax_a2);
        // blocks 3 and 4 @ H/8, H/32 (141-144) on conv_mfma_h: conv1 carries the identity branch ds(pooled input) as 32 / 64 more
        // output channels whose weights sit on the centre tap only (no ReLU on those tiles); conv2 then reads conv1's half of that
        // buffer and adds the other half.  Block 4's conv1 max-pools x3 4 x 4 while it stages it.
        auto block_h = [&](const char* n1, const char* n2, const float* in, float* tr, float* xo, int cin, int cout, int Hi, int Wi, bool prepooled) {
            const std::string k1 = std::string(n1) + ".h", k2 = std::string(n2) + ".wp";
            ConvM m;
            m.in = in; m.out = tr; m.wp = wp(k1.c_str()); m.bias = wp((std::string(n1) + ".hb").c_str()); m.xf = nullptr; m.active = nullptr; m.res = nullptr;
            m.Hi = prepooled ? Hi / 4 : Hi; m.Wi = prepooled ? Wi / 4 : Wi; m.H = Hi / 4; m.W = Wi / 4;
            m.CIN = cin; m.COUT = 2 * cout; m.NCH = 1; m.relu = 2; m.relu_nt = cout / 32;
            m.nblk = cout / 32; m.istride = cin; m.ostride = 2 * cout; m.ooff = 0;
            m.unscale = 1.0f / wscale.at(k1);
            const std::string tag1 = std::string("conv3x3_") + n1, tag2 = std::string("conv3x3_") + n2;
            // small layers are bound by per-workgroup latency: 8-row tiles (r02: b3c1 0.36 -> 0.32 ms, b3c2 0.58 -> 0.49 ms)
            if (prepooled && batch < 16) KPB_LAUNCH(ctx, tag1.c_str(), (conv_mfma_h<3, 1, 16, false, false, false, 2, 1, false, 2, true>), dim3(cdiv(m.W, 16), cdiv(m.H, 8), batch * m.nblk), dim3(256), 0, st, m);
            else if (prepooled && cin == 32) KPB_LAUNCH(ctx, tag1.c_str(), (conv_mfma_h<3, 1, 32, false, false, false, 1, 4, false, 2, false, false, 2>), dim3(cdiv(m.W, 16), cdiv(m.H, 16), batch * m.nblk), dim3(256), 0, st, m);
            else if (prepooled) KPB_LAUNCH(ctx, tag1.c_str(), (conv_mfma_h<3, 1, 16, false, false, false, 1, 2, false, 2, false, false, 2>), dim3(cdiv(m.W, 16), cdiv(m.H, 8), batch * m.nblk), dim3(256), 0, st, m);
            else if (batch >= 16) KPB_LAUNCH(ctx, tag1.c_str(), (conv_mfma_h<3, 1, 32, true, false, false, 1, 4, false, 4, false, false, 2>), dim3(cdiv(m.W, 16), cdiv(m.H, 16), batch * m.nblk), dim3(256), 0, st, m);
            else {      // a handful of images (the drop-in path runs ONE): a 15 x 20 map in 16 x 16 tiles with two n-tiles each is 4 workgroups of
                        // pure latency (41 us); 8-row tiles with one n-tile each are 16 (same weights, same arithmetic per output)
                m.nblk = 2 * cout / 32;
                KPB_LAUNCH(ctx, tag1.c_str(), (conv_mfma_h<3, 1, 32, true, false, false, 1, 1, false, 4, true>), dim3(cdiv(m.W, 16), cdiv(m.H, 8), batch * m.nblk), dim3(256), 0, st, m);
            }
            ConvM c2;
            c2.in = tr; c2.out = xo; c2.wp = wp(k2.c_str()); c2.bias = wp((std::string(n2) + ".bp").c_str()); c2.xf = nullptr; c2.active = nullptr;
            c2.res = tr + cout; c2.rstride = 2 * cout;
            c2.Hi = Hi / 4; c2.Wi = Wi / 4; c2.H = Hi / 4; c2.W = Wi / 4; c2.CIN = cout; c2.COUT = cout; c2.NCH = cout / 32; c2.relu = 0; c2.nblk = 1;
            c2.istride = 2 * cout; c2.ostride = cout; c2.ooff = 0;
            c2.unscale = 1.0f / wscale.at(k2);
            if (cout == 32 && batch < 16) KPB_LAUNCH(ctx, tag2.c_str(), (conv_mfma_h<3, 1, 32, false, false, false, 1, 1, false, 2, true>), dim3(cdiv(c2.W, 16), cdiv(c2.H, 8), batch), dim3(256), 0, st, c2);
            else if (cout == 32) KPB_LAUNCH(ctx, tag2.c_str(), (conv_mfma_h<3, 1, 32, false, false, false, 1, 1>), dim3(cdiv(c2.W, 16), cdiv(c2.H, 8), batch), dim3(256), 0, st, c2);
            else if (batch >= 16) KPB_LAUNCH(ctx, tag2.c_str(), (conv_mfma_h<3, 1, 32, false, false, false, 1, 4, false, 2, false, false, 2>), dim3(cdiv(c2.W, 16), cdiv(c2.H, 16), batch), dim3(256), 0, st, c2);
            else {
                c2.nblk = cout / 32;
                KPB_LAUNCH(ctx, tag2.c_str(), (conv_mfma_h<3, 1, 32, false, false, false, 1, 1, false, 2, true>), dim3(cdiv(c2.W, 16), cdiv(c2.H, 8), batch * c2.nblk), dim3(256), 0, st, c2);
            }
        };
        block_h("b3c1", "b3c2", p2, t3r3, x3, 16, 32, H / 2, W / 2, true);
        if (batch >= 16) {      // block 4's input pooled once, by a kernel of its own (see maxpool4_nhwc); a handful of images keep the fused form (one launch fewer)
            KPB_LAUNCH(ctx, "maxpool4_x3", maxpool4_nhwc, dim3((unsigned)(((size_t)(H / 32) * (W / 32) * 8 + 255) / 256), batch), dim3(256), 0, st, x3, p3, H / 32, W / 32, 32);
            block_h("b4c1", "b4c2", p3, t4r4, x4, 32, 64, H / 8, W / 8, true);
        } else
            block_h("b4c1", "b4c2", x3, t4r4, x4, 32, 64, H / 8, W / 8, false);
    } else {
        // strict fp32: block 1 and the 3x3 convolutions on the fp32 vector ALUs, conv2 of blocks 3 / 4 on the fp32 MFMA
        KPB_LAUNCH(ctx, "alike_block1", alike_block1, dim3(cdiv(W, B1_TW), cdiv(H, B1_TH), batch), dim3(256), 0, st, b1);
        c = ConvArgs{p1, t2, wp("b2c1.w"), wp("b2c1.b"), nullptr, nullptr, nullptr, nullptr, H / 2, W / 2};      // pooled by block1
        launch_conv<8, 16, 1, false, 4, 1>(ctx, "conv3x3_b2c1", st, c, batch);
        c = ConvArgs{t2, x2, wp("b2c2.w"), wp("b2c2.b"), p1, wp("b2ds.w"), wp("b2ds.b"), nullptr, H / 2, W / 2};
        launch_conv<16, 16, 1, true, 8, 1>(ctx, "conv3x3_b2c2", st, c, batch);
        // block3 @ H/8 (141-142): pool4
        c = ConvArgs{x2, t3, wp("b3c1.w"), wp("b3c1.b"), nullptr, wp("b3ds.w"), wp("b3ds.b"), r3, H / 8, W / 8};
        launch_conv<16, 32, 4, false, 4, 1, true, 16>(ctx, "conv3x3_b3c1", st, c, batch);        // 80 columns at 480x640: 16-wide tiles divide them
        {   // conv2 of block3 on the MFMA kernel, identity branch precomputed (ALike.py:72-80)
            ConvM m;
            m.in = t3; m.out = x3; m.wp = wp("b3c2.wp"); m.bias = wp("b3c2.bp"); m.xf = nullptr; m.active = nullptr; m.res = r3;
            m.Hi = H / 8; m.Wi = W / 8; m.H = H / 8; m.W = W / 8; m.CIN = 32; m.COUT = 32; m.NCH = 1; m.relu = 0; m.nblk = 1;
            m.istride = 32; m.ostride = 32; m.ooff = 0;
            KPB_LAUNCH(ctx, "conv3x3_b3c2", (conv_mfma<3, 1, 32, false, false, false, 1>), dim3(cdiv(m.W, 16), cdiv(m.H, 8), batch), dim3(256), 0, st, m);
        }
        // block4 @ H/32 (143-144): pool4
        c = ConvArgs{x3, t4, wp("b4c1.w"), wp("b4c1.b"), nullptr, wp("b4ds.w"), wp("b4ds.b"), r4, H / 32, W / 32};
        launch_conv<32, 64, 4, false, 4, 1, true, 16>(ctx, "conv3x3_b4c1", st, c, batch);
        {
            ConvM m;
            m.in = t4; m.out = x4; m.wp = wp("b4c2.wp"); m.bias = wp("b4c2.bp"); m.xf = nullptr; m.active = nullptr; m.res = r4;
            m.Hi = H / 32; m.Wi = W / 32; m.H = H / 32; m.W = W / 32; m.CIN = 64; m.COUT = 64; m.NCH = 2; m.relu = 0; m.nblk = 1;
            m.istride = 64; m.ostride = 64; m.ooff = 0;
            KPB_LAUNCH(ctx, "conv3x3_b4c2", (conv_mfma<3, 1, 32, false, false, false, 2>), dim3(cdiv(m.W, 16), cdiv(m.H, 8), batch), dim3(256), 0, st, m);
        }
        // aggregation 1x1 + ReLU of block 2 (147-148); agg1 is fused into the head
        KPB_LAUNCH(ctx, "conv1x1_agg2", conv1x1_relu<16>, dim3((unsigned)((B * P / 4 + 255) / 256)), dim3(256), 0, st, x2, a2, wp("agg2.w"), wp("head.ws") + 16, S2, B * P / 4, nullptr, nullptr);
    }
    // aggregation 1x1 + ReLU (149-150), each with its share of the score logit and -- dense mode -- of every head row
    if (batch < 16) {
        const AggArgs g3{x3, a3, wp("agg3.w"), wp("head.ws") + 32, S3, B * P / 64, wp("head.wT") + 32 * 64, E3};
        const AggArgs g4{x4, a4, wp("agg4.w"), wp("head.ws") + 48, S4, B * P / 1024, wp("head.wT") + 48 * 64, E4};
        const unsigned nb3 = (unsigned)((B * P / 64 + 255) / 256), nb4 = (unsigned)((B * P / 1024 + 255) / 256);
        KPB_LAUNCH(ctx, "conv1x1_agg34", conv1x1_relu_34, dim3(nb3 + nb4), dim3(256), 0, st, g3, g4, nb3);
    } else {
        KPB_LAUNCH(ctx, "conv1x1_agg3", conv1x1_relu<32>, dim3((unsigned)((B * P / 64 + 255) / 256)), dim3(256), 0, st, x3, a3, wp("agg3.w"), wp("head.ws") + 32, S3, B * P / 64, wp("head.wT") + 32 * 64, E3);
        KPB_LAUNCH(ctx, "conv1x1_agg4", conv1x1_relu<64>, dim3((unsigned)((B * P / 1024 + 255) / 256)), dim3(256), 0, st, x4, a4, wp("agg4.w"), wp("head.ws") + 48, S4, B * P / 1024, wp("head.wT") + 48 * 64, E4);
    }
    // upsample + concat + head (151-162)
    if (desc_out_dev) {
        HybArgs hy{x1, a2, E3, E4, wp("agg1.w"), wp("head.wT"), wp("head.ws"), score_out_dev, desc_out_dev, H, W, amax_x1, amax_a2, 0.f, 1.f, 1.f, nullptr, 1.f};
        const int work4 = cdiv(H, 4) * cdiv(W / 32, SEG_TILES);     // four consecutive rows of one 128-pixel column band per workgroup
        if (h16) {
            hy.l1_agg1 = k.at("agg1.l1"); hy.inv_ws_h = k.at("head.inv_ws"); hy.ws_h = 1.0f / hy.inv_ws_h;
            hy.a1h16 = reinterpret_cast<const uint4*>(wp("agg1.h16")); hy.inv_wa = k.at("agg1.inv_wa");
            // row groups per persistent workgroup: 30 (15: +1-2 %, 40 / 60: the same, 120: +1 %; profiles/r04_ab_knobs.txt); H is a
            // multiple of 32 (kpb_net_forward checks), so every group has its four rows
            // -- as long as that leaves about two rounds of workgroups for the chip's 768 slots: a single image walks 2 groups per workgroup
            if (!hp_ring_rows_ok(H, H / 2, HP_A2R) || !hp_ring_rows_ok(H, H / 8, HP_ER) || !hp_ring_rows_ok(H, H / 32, HP_ER))
                return kpb_fail(ctx, KPB_E_UNSUPPORTED, "alike head: the LDS rings cannot hold the source rows of %d row groups at height %d", HP_LA + 1, H);
            const int bands_ = cdiv(W / 32, HP_TILES), groups_ = H / 4;
            const int hp = std::max(2, std::min(30, (int)(((long long)bands_ * groups_ * batch) / 1536)));
            KPB_LAUNCH(ctx, "alike_head_dense", alike_head_f16p, dim3(cdiv(W / 32, HP_TILES) * cdiv(H / 4, hp), batch), dim3(256), 0, st, hy,
                       reinterpret_cast<const uint4*>(wp("head.wh16")), hp);
        } else
            KPB_LAUNCH(ctx, "alike_head_dense", alike_head_hyb, dim3(work4, batch), dim3(256), 0, st, hy);
    } else {
        LinArgs la{x1, S2, S3, S4, wp("agg1.w"), wp("head.ws"), score_out_dev, H, W,
                   (float)(H / 2 - 1) / (float)(H - 1), (float)(W / 2 - 1) / (float)(W - 1), (float)(H / 8 - 1) / (float)(H - 1), (float)(W / 8 - 1) / (float)(W - 1),
                   (float)(H / 32 - 1) / (float)(H - 1), (float)(W / 32 - 1) / (float)(W - 1)};
        KPB_LAUNCH(ctx, "alike_head_score", alike_score_lin, dim3(cdiv(H * W, 256), batch), dim3(256), 0, st, la);
    }
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}

int AlikeNet::desc_at(const float* pts_dev, int pts_cols, int max_n, const int32_t* n_dev, float* out_dev)
{
    if (B == 0) return kpb_fail(ctx, KPB_E_INVALID, "kpb_net_desc_at: no forward has run");
    const int kpw = (long long)B * max_n >= 32768 ? 8 : 1;
    DescAtArgs a{head_args(nullptr, nullptr), pts_dev, n_dev, out_dev, pts_cols, max_n, kpw};
    KPB_LAUNCH(ctx, "alike_desc_at", alike_desc_at, dim3(cdiv(max_n, 4 * kpw), B), dim3(256), 0, ctx->stream, a);
    KPB_HIP(ctx, hipGetLastError());
    return KPB_OK;
}

}  // namespace

int alike_create(kpb_ctx* ctx, const KpbwBlob& bl, kpb_net** out)
{
    // this build carries the ALIKE-t channel plan (c1..c4 = 8,16,32,64, dim 64: config/config_MHA.yaml Alike_params)
    const uint32_t c1 = 8, c2 = 16, c3 = 32, c4 = 64, dim = 64;
    struct { const char* n; std::vector<uint32_t> d; } need[] = {
        {"b1c1.w", {c1, 3, 3, 3}}, {"b1c1.b", {c1}}, {"b1c2.w", {c1, c1, 3, 3}}, {"b1c2.b", {c1}},
        {"b2c1.w", {c2, c1, 3, 3}}, {"b2c1.b", {c2}}, {"b2c2.w", {c2, c2, 3, 3}}, {"b2c2.b", {c2}}, {"b2ds.w", {c2, c1}}, {"b2ds.b", {c2}},
        {"b3c1.w", {c3, c2, 3, 3}}, {"b3c1.b", {c3}}, {"b3c2.w", {c3, c3, 3, 3}}, {"b3c2.b", {c3}}, {"b3ds.w", {c3, c2}}, {"b3ds.b", {c3}},
        {"b4c1.w", {c4, c3, 3, 3}}, {"b4c1.b", {c4}}, {"b4c2.w", {c4, c4, 3, 3}}, {"b4c2.b", {c4}}, {"b4ds.w", {c4, c3}}, {"b4ds.b", {c4}},
        {"agg1.w", {dim / 4, c1}}, {"agg2.w", {dim / 4, c2}}, {"agg3.w", {dim / 4, c3}}, {"agg4.w", {dim / 4, c4}},
        {"head.w", {dim + 1, dim}}};
    for (auto& nd : need)
        if (!bl.get(nd.n, nd.d))
            return kpb_fail(ctx, KPB_E_WEIGHTS, "kpb_net_create: tensor %s missing or not ALIKE-t shaped "
                            "(this build supports c1..c4 = 8,16,32,64, dim = 64)", nd.n);
    AlikeNet* net = new AlikeNet();
    net->ctx = ctx; net->arch = KPB_ARCH_ALIKE; net->dim = dim; net->desc_div = 1;
    WeightStage ws;
    std::vector<float> tmp;
    {   // block1 conv1: [co][ci][ky][kx] -> [(ci,ky,kx)][co]
        const float* w = bl.get("b1c1.w", {c1, 3, 3, 3});
        tmp.assign(27 * 8, 0.f);
        for (int o = 0; o < 8; ++o) for (int k = 0; k < 27; ++k) tmp[k * 8 + o] = w[o * 27 + k];
        ws.put("b1c1.w", tmp);
        ws.put_raw("b1c1.b", bl.get("b1c1.b", {c1}), 8);
        repack3x3(bl.get("b1c2.w", {c1, c1, 3, 3}), 8, 8, tmp); ws.put("b1c2.w", tmp);
        {   // split-f16 fragments, each pack scaled to max |w| in [2^12, 2^13), and the constants of conv1's output bound
            const float* w1 = bl.get("b1c1.w", {c1, 3, 3, 3});
            const float* w2 = bl.get("b1c2.w", {c1, c1, 3, 3});
            const float s1 = weight_scale_h(w1, 8 * 27), s2 = weight_scale_h(w2, 8 * 72);
            ws.put("b1c1.pairs", pack_b1c1_pairs(w1, s1));
            ws.put("b1c2.pairs", pack_b1c2_pairs(w2, s2));
            net->k["b1c1.inv_ws"] = 1.0f / s1; net->k["b1c2.inv_ws"] = 1.0f / s2;
            net->k["b1c1.l1"] = l1_rows(w1, 8, 27); net->k["b1c1.bmax"] = max_abs(bl.get("b1c1.b", {c1}), 8);
        }
        ws.put_raw("b1c2.b", bl.get("b1c2.b", {c1}), 8);
    }
    const uint32_t ch[5] = {0, c1, c2, c3, c4};
    for (int i = 2; i <= 4; ++i) {
        char nm[16];
        const uint32_t ci = ch[i - 1], co = ch[i];
        snprintf(nm, 16, "b%dc1.w", i); repack3x3(bl.get(nm, {co, ci, 3, 3}), co, ci, tmp); ws.put(nm, tmp);
        snprintf(nm, 16, "b%dc1.b", i); ws.put_raw(nm, bl.get(nm, {co}), co);
        snprintf(nm, 16, "b%dc2.w", i); repack3x3(bl.get(nm, {co, co, 3, 3}), co, co, tmp); ws.put(nm, tmp);
        snprintf(nm, 16, "b%dc2.b", i); ws.put_raw(nm, bl.get(nm, {co}), co);
        snprintf(nm, 16, "b%dds.w", i); transpose(bl.get(nm, {co, ci}), co, ci, tmp); ws.put(nm, tmp);
        snprintf(nm, 16, "b%dds.b", i); ws.put_raw(nm, bl.get(nm, {co}), co);
    }
    {   // block 2 on the split-f16 MFMA kernel: fragments + combined biases
        const float* w1 = bl.get("b2c1.w", {c2, c1, 3, 3});
        const float* w2 = bl.get("b2c2.w", {c2, c2, 3, 3});
        const float* wd = bl.get("b2ds.w", {c2, c1});
        const float* wa = bl.get("agg2.w", {dim / 4, c2});
        const float s1 = weight_scale_h(w1, 16 * 72);
        const float s2 = std::min(weight_scale_h(w2, 16 * 144), weight_scale_h(wd, 16 * 8));     // conv2 and the identity branch share one accumulation
        const float sa = weight_scale_h(wa, 16 * 16);
        ws.put("b2c1.h16", pack_h16(w1, 8, nullptr, s1));
        ws.put("b2c2.h16", pack_h16(w2, 16, wd, s2));
        const float* b2 = bl.get("b2c2.b", {c2});
        const float* bd = bl.get("b2ds.b", {c2});
        tmp.assign(16, 0.f);
        for (int i = 0; i < 16; ++i) tmp[i] = b2[i] + bd[i];
        ws.put("b2c2.bsum", tmp);
        ws.put("agg2.h16", pack_1x1_h16(wa, sa));
        net->k["b2c1.inv_ws"] = 1.0f / s1; net->k["b2c2.inv_ws"] = 1.0f / s2; net->k["agg2.inv_ws"] = 1.0f / sa;
        net->k["b2c1.l1"] = l1_rows(w1, 16, 72); net->k["b2c1.bmax"] = max_abs(bl.get("b2c1.b", {c2}), 16);
        net->k["b2c2.l1"] = l1_rows(w2, 16, 144); net->k["b2ds.l1"] = l1_rows(wd, 16, 8); net->k["b2c2.bsummax"] = max_abs(tmp.data(), 16);
        net->k["agg1.l1"] = l1_rows(bl.get("agg1.w", {dim / 4, c1}), 16, 8);
    }
    auto put_mfma = [&](const char* name, const float* w, int cout, int cin, int ntb) {
        if (conv_mfma_use_h16()) {
            const float sc = weight_scale_h(w, (size_t)cout * cin * 9);
            ws.put(name, pack_mfma_h(w, cout, cin, 3, 32, ntb, sc));
            ws.wscale[name] = sc;
        } else {
            ws.put(name, pack_mfma(w, cout, cin, 3, 32, ntb));
        }
    };
    auto put_c1ds = [&](const char* name, const float* w1, const float* b1, const float* wds, const float* bds, int cout, int cin) {
        // OIHW [2 cout][cin][3][3]: rows 0 .. cout-1 = conv1, rows cout .. = the 1x1 identity branch on the centre tap
        std::vector<float> w((size_t)2 * cout * cin * 9, 0.0f), bb(2 * cout);
        memcpy(w.data(), w1, (size_t)cout * cin * 9 * sizeof(float));
        for (int o = 0; o < cout; ++o) {
            for (int c = 0; c < cin; ++c) w[((size_t)(cout + o) * cin + c) * 9 + 4] = wds[(size_t)o * cin + c];
            bb[o] = b1[o]; bb[cout + o] = bds[o];
        }
        const float sc = weight_scale_h(w.data(), w.size());
        const std::string k = std::string(name) + ".h";
        ws.put(k, pack_mfma_h(w.data(), 2 * cout, cin, 3, cin == 16 ? 16 : 32, 2, sc));
        ws.wscale[k] = sc;
        ws.put(std::string(name) + ".hb", bb);
    };
    if (conv_mfma_use_h16()) {
        put_c1ds("b3c1", bl.get("b3c1.w", {c3, c2, 3, 3}), bl.get("b3c1.b", {c3}), bl.get("b3ds.w", {c3, c2}), bl.get("b3ds.b", {c3}), 32, 16);
        put_c1ds("b4c1", bl.get("b4c1.w", {c4, c3, 3, 3}), bl.get("b4c1.b", {c4}), bl.get("b4ds.w", {c4, c3}), bl.get("b4ds.b", {c4}), 64, 32);
    }
    put_mfma("b3c2.wp", bl.get("b3c2.w", {c3, c3, 3, 3}), 32, 32, 1);
    ws.put("b3c2.bp", pad_bias(bl.get("b3c2.b", {c3}), 32, 32));
    put_mfma("b4c2.wp", bl.get("b4c2.w", {c4, c4, 3, 3}), 64, 64, 2);
    ws.put("b4c2.bp", pad_bias(bl.get("b4c2.b", {c4}), 64, 64));
    for (int i = 1; i <= 4; ++i) {
        char nm[16];
        snprintf(nm, 16, "agg%d.w", i);
        transpose(bl.get(nm, {dim / 4, ch[i]}), dim / 4, ch[i], tmp); ws.put(nm, tmp);
    }
    {
        const float* hw = bl.get("head.w", {dim + 1, dim});
        transpose(hw, 64, 64, tmp); ws.put("head.wT", tmp);
        ws.put_raw("head.ws", hw + 64 * 64, 64);
        // split-f16 fragments of the fine-group rows (alike_head_f16p): [hi/lo][kb][nh][h][n][j] halves,
        // value = head.w[o = 32 nh + n][c = 16 kb + 8 h + j]; hi = f16(w), lo = f16(w - hi), to nearest
        // (rows 0..63 x channels 0..31: the part this pack carries sets its power-of-two scale)
        float hmax = 0.0f;
        for (int o = 0; o < 64; ++o) for (int cc = 0; cc < 32; ++cc) hmax = std::max(hmax, std::fabs(hw[o * 64 + cc]));
        const float sh = weight_scale_h(&hmax, 1);
        net->k["head.inv_ws"] = 1.0f / sh;
        std::vector<uint16_t> hl(2 * 2 * 2 * 2 * 32 * 8);
        for (int kb = 0; kb < 2; ++kb) for (int nh = 0; nh < 2; ++nh) for (int hh = 0; hh < 2; ++hh) for (int n = 0; n < 32; ++n) for (int j = 0; j < 8; ++j) {
            // k slot (hh, j) of block kb: group 1 (up2 a2) keeps channel 8 hh + j; group 0 (agg1) is produced by an MFMA whose accumulator
            // leaves lane half hh with channels {0..3, 8..11} + 4 hh (rows (r & 3) + 8 (r >> 2) + 4 hh of a 32 x 32 tile), so its k order is that
            const int cch = kb == 0 ? (j < 4 ? j : j + 4) + 4 * hh : 16 + 8 * hh + j;
            const float w = hw[(32 * nh + n) * 64 + cch] * sh;
            const _Float16 hi = (_Float16)w;
            const _Float16 lo = (_Float16)(w - (float)hi);
            const size_t at = ((((size_t)kb * 2 + nh) * 2 + hh) * 32 + n) * 8 + j;
            memcpy(&hl[at], &hi, 2);
            memcpy(&hl[2048 + at], &lo, 2);
        }
        tmp.assign(2048, 0.f);
        memcpy(tmp.data(), hl.data(), 8192);
        ws.put("head.wh16", tmp);
        // agg1 (8 -> 16, 1x1) as the A operand of v_mfma_f32_32x32x16_f16 (r04): rows = output channels (16 of 32 used), k = input
        // channel (8 of 16 used): lane (row o, k half) holds agg1[o][c = 0..7] in its eight halves for o < 16 and k half 0, zeros elsewhere
        {
            const float* a1w = bl.get("agg1.w", {dim / 4, c1});            // [16][8]
            const float sa1 = weight_scale_h(a1w, 16 * 8);
            net->k["agg1.inv_wa"] = 1.0f / sa1;
            std::vector<uint16_t> fr(2 * 64 * 8, 0);
            for (int o = 0; o < 16; ++o) for (int c = 0; c < 8; ++c) {
                const float w = a1w[o * 8 + c] * sa1;
                const _Float16 hi = (_Float16)w;
                const _Float16 lo = (_Float16)(w - (float)hi);
                memcpy(&fr[(size_t)o * 8 + c], &hi, 2);
                memcpy(&fr[(size_t)(64 + o) * 8 + c], &lo, 2);
            }
            tmp.assign(512, 0.f);
            memcpy(tmp.data(), fr.data(), 2048);
            ws.put("agg1.h16", tmp);
        }
    }
    if (int rc = ws.upload(net)) { delete net; return rc; }
    *out = net;
    return KPB_OK;
}
