// Which store pattern does the memory system take fastest from a head-shaped kernel (4-wave workgroups, 8 KB tiles of
// [32 pixels][64 channels] fp32 per wave)?  r04 companion of hbm_mix_rate.hip (the guide measures 6.0-6.2 TB/s
// for plain dword-per-lane 256-B stores, r03's microbenchmark topped out at 5.7 with nontemporal ones).
//   PAT 0  one 256-B pixel per instruction (lanes 0..63 = channels 0..63)                      -- what v_permlane32_swap would give
//   PAT 1  the head's r03 pattern: an instruction writes channels 0..31 of pixels r and r + 4 (two 128-B half pixels); the
//          other halves (channels 32..63) follow ONE TILE LATER
//   PAT 2  16 bytes per lane: 1 KB = four pixels per instruction
//   POL 0 nontemporal, 1 default policy
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/hsp scripts/hbm_store_patterns.hip && /tmp/hsp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
extern __shared__ char dyn[];
typedef float f4 __attribute__((ext_vector_type(4)));
template <int POL, class T> __device__ __forceinline__ void st(T v, T* p) { if (POL == 0) __builtin_nontemporal_store(v, p); else *p = v; }

template <int PAT, int POL, int LOADS>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int tiles)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, p = lane & 31, h = lane >> 5;
    float v = (float)lane;
    if (dyn[threadIdx.x] == 77) v += 1.0f;
    const size_t seg = (size_t)(blockIdx.x * 4 + wv) * tiles;
    float* base = out + seg * 2048;
    const float* ib = in + seg * 256 * (LOADS ? LOADS : 1);
    f4 cur[LOADS ? LOADS : 1], nxt[LOADS ? LOADS : 1];
    for (int q = 0; q < LOADS; ++q) cur[q] = *reinterpret_cast<const f4*>(ib + q * 256 + lane * 4);
    float* pend = nullptr;
    for (int t = 0; t < tiles; ++t) {
        if (t + 1 < tiles) for (int q = 0; q < LOADS; ++q) nxt[q] = *reinterpret_cast<const f4*>(ib + ((t + 1) * LOADS + q) * 256 + lane * 4);
        float* d = base + (size_t)t * 2048;
        float w = v;
        for (int q = 0; q < LOADS; ++q) w += cur[q].x + cur[q].y + cur[q].z + cur[q].w;
        if (PAT == 0) {
#pragma unroll
            for (int r = 0; r < 32; ++r) st<POL>(w + r, d + r * 64 + lane);
        } else if (PAT == 1) {
            if (pend) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st<POL>(w - r, pend + ((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 + p);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) st<POL>(w + r, d + ((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + p);
            pend = d;
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) { f4 x = {w, w + r, w, w}; st<POL>(x, reinterpret_cast<f4*>(d + r * 256 + lane * 4)); }
        }
        for (int q = 0; q < LOADS; ++q) cur[q] = nxt[q];
    }
    if (PAT == 1 && pend) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st<POL>(v - r, pend + ((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 + p);
    }
}

template <int PAT, int POL, int LOADS>
void run(float* out, float* in, size_t bytes, int tiles, int lds, hipEvent_t e0, hipEvent_t e1)
{
    const int wgs = (int)(bytes / (4 * (size_t)tiles * 8192));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<PAT, POL, LOADS>), dim3(wgs), dim3(256), lds, 0, out, in, tiles);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double tot = bytes * (1.0 + LOADS / 8.0);
    printf("pattern %d %-7s tiles/wave %2d %d wg/cu loads %d KB per 8 KB stored: %.3f ms  stores %.0f GB/s  total %.0f GB/s\n", PAT, POL ? "default" : "nt",
           tiles, 160000 / lds, LOADS, best, bytes / best * 1e-6, tot / best * 1e-6);
}

int main()
{
    const size_t bytes = (size_t)16 << 30;
    float* out; CK(hipMalloc(&out, bytes));
    float* in; CK(hipMalloc(&in, bytes / 4)); CK(hipMemset(in, 0, bytes / 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int tiles : {4, 16}) for (int lds : {53000, 78000}) {
        run<0, 0, 0>(out, in, bytes, tiles, lds, e0, e1); run<0, 1, 0>(out, in, bytes, tiles, lds, e0, e1);
        run<1, 0, 0>(out, in, bytes, tiles, lds, e0, e1); run<1, 1, 0>(out, in, bytes, tiles, lds, e0, e1);
        run<2, 0, 0>(out, in, bytes, tiles, lds, e0, e1); run<2, 1, 0>(out, in, bytes, tiles, lds, e0, e1);
        run<0, 0, 1>(out, in, bytes, tiles, lds, e0, e1); run<0, 1, 1>(out, in, bytes, tiles, lds, e0, e1);
        run<1, 0, 1>(out, in, bytes, tiles, lds, e0, e1); run<1, 1, 1>(out, in, bytes, tiles, lds, e0, e1);
        run<2, 0, 1>(out, in, bytes, tiles, lds, e0, e1); run<2, 1, 1>(out, in, bytes, tiles, lds, e0, e1);
    }
    return 0;
}
