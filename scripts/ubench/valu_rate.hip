// Microbenchmark: fp32 FMA issue rate with scalar (SGPR) vs vector weight operands, packed vs plain.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* out, int n)
{
    float acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = threadIdx.x * 0.001f + j;
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {   // weights uniform (scalar loads), compiler free to pack
            const float* ww = w + (it & 63) * 16;
#pragma unroll
            for (int j = 0; j < 16; ++j) { acc[j] = fmaf(x0, ww[j], acc[j]); }
#pragma unroll
            for (int j = 0; j < 16; ++j) { acc[j] = fmaf(x1, ww[j], acc[j]); }
#pragma unroll
            for (int j = 0; j < 16; ++j) { acc[j] = fmaf(x2, ww[j], acc[j]); }
#pragma unroll
            for (int j = 0; j < 16; ++j) { acc[j] = fmaf(x3, ww[j], acc[j]); }
        } else {           // weights per lane in VGPRs
            float wv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) wv[j] = x0 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = fmaf(x1 + r, wv[j], acc[j]);
        }
        x0 += 1e-7f;
    }
    float s = 0; for (int j = 0; j < 16; ++j) s += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *w, *o; hipMalloc(&w, 64 * 16 * 4); hipMalloc(&o, 4096 * 256 * 4); hipMemset(w, 0, 64 * 16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, 0, w, o, ITER);
            else hipLaunchKernelGGL(k<1>, dim3(4096), dim3(256), 0, 0, w, o, ITER);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = 4096.0 * 256 * ITER * 64 * 2;
            printf("mode %d rep %d: %.3f ms  %.1f TFLOP/s\n", mode, rep, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
