// Sustained rate of v_mfma_f32_32x32x16_f16 on this box (what a split-f16 kernel can be measured against):
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Every wave keeps four independent accumulators busy with operands held in registers (no memory traffic in the loop);
// operands are pseudo-random halves, because the clock the chip holds depends on the data toggling.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void spin(float* out, int iters, unsigned seed)
{
    unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (_Float16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f)); };
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) { a[i][j] = rnd(); b[i][j] = rnd(); }
    f16v acc[4] = {{0}, {0}, {0}, {0}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[i], acc[i], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* out;
    (void)hipMalloc(&out, (size_t)cus * 16 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {      // 1, 2, 3 waves per SIMD
        const int grid = cus * wgs_per_cu, iters = 20000;
        spin<<<grid, 256>>>(out, 1000, 1);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        spin<<<grid, 256>>>(out, iters, 7);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)grid * 4 * iters * 16;           // wave-level instructions
        const double flops = mfma * 2.0 * 32 * 32 * 16;
        printf("%d CUs, %d wave(s)/SIMD: %.2f ms, %.1f TFLOP/s f16, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", cus, wgs_per_cu, ms, flops / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (mfma / (cus * 4.0)));
    }
    return 0;
}
