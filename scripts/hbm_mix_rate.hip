// Microbenchmark behind DESIGN.md section 5 ("where the head stands"): what the memory system takes from a kernel shaped like the
// head -- 8 KB tiles written with 256-byte streaming row stores by 4-wave workgroups at the head's occupancy -- when 0, 1 or 2 KB of
// streaming loads per tile are mixed in (the head reads 1 KB of x1 per tile plus its a2 band and strips).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/hbm_mix_rate scripts/hbm_mix_rate.hip && /tmp/hbm_mix_rate     (profiles/r03_hbm_mix_rate.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
extern __shared__ char dyn[];
typedef float f4 __attribute__((ext_vector_type(4)));
// LOADS: 16-byte loads per lane and tile (0, 1 = 1 KB per wave-tile, 2 = 2 KB); the value loaded for tile t+1 is stored in tile t+1
template <int LOADS>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int tiles, int wgs)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float v = (float)lane;
    if (dyn[threadIdx.x] == 77) v += 1.0f;
    const int wg = blockIdx.x;
    float* base = out + ((size_t)(wg * 4 + wv) * tiles) * 2048;
    const float* ib = in + ((size_t)(wg * 4 + wv) * tiles) * 256 * (LOADS ? LOADS : 1);
    f4 cur[LOADS ? LOADS : 1], nxt[LOADS ? LOADS : 1];
    for (int q = 0; q < LOADS; ++q) cur[q] = *reinterpret_cast<const f4*>(ib + q * 256 + lane * 4);
    for (int t = 0; t < tiles; ++t) {
        if (t + 1 < tiles) for (int q = 0; q < LOADS; ++q) nxt[q] = *reinterpret_cast<const f4*>(ib + ((t + 1) * LOADS + q) * 256 + lane * 4);
        float* d = base + (size_t)t * 2048;
        float w = v;
        for (int q = 0; q < LOADS; ++q) w += cur[q].x + cur[q].y + cur[q].z + cur[q].w;
#pragma unroll
        for (int r = 0; r < 32; ++r) __builtin_nontemporal_store(w + r, d + r * 64 + lane);
        for (int q = 0; q < LOADS; ++q) cur[q] = nxt[q];
    }
}
int main()
{
    const size_t bytes = (size_t)16 << 30;
    float* out; CK(hipMalloc(&out, bytes));
    float* in; CK(hipMalloc(&in, bytes / 4)); CK(hipMemset(in, 0, bytes / 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int tiles : {4, 16}) for (int lds : {53000, 30000}) for (int loads = 0; loads < 3; ++loads) {
        const int wgs = (int)(bytes / (4 * (size_t)tiles * 8192));
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (loads == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), lds, 0, out, in, tiles, wgs);
            if (loads == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), lds, 0, out, in, tiles, wgs);
            if (loads == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), lds, 0, out, in, tiles, wgs);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double tot = bytes * (1.0 + loads / 8.0);
        printf("tiles/wave %2d lds %5d (%d wg/cu) loads %d KB per 8 KB stored: %.3f ms  stores %.0f GB/s  total %.0f GB/s\n", tiles, lds, 160000 / lds, loads, best, bytes / best * 1e-6, tot / best * 1e-6);
    }
    return 0;
}
