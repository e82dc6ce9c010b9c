// Does a packed-fp32 multiply whose LOW lane takes the HIGH dword of a source pair (op_sel) compute the right product while other waves of
// the SIMD run MFMAs?  (r05: the run-to-run glitches of the dense ALIKE head, DESIGN.md section 3: ISA-level knock-outs on a build that
// failed in every run put the fault in ten `v_pk_mul_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[0,1]` instructions -- both result lanes are
// the same product, and the LOW one came out wrong in lanes 48..63.)
// Every wave alternates a burst of MFMAs on its own registers with the test: one asm statement on fixed registers, the product taken by
// the packed instruction under test and by two plain v_mul_f32; mismatches are counted per 16-lane row and per result lane (lo / hi), and
// the first samples are kept with all four operand dwords so that the operand the wrong lane really used can be read off.
//   hipcc -O3 --offload-arch=gfx950 -o pk_opsel pk_opsel.hip && ./pk_opsel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define CLOB "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", \
             "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", \
             "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243"

// operands: v[200:201] = (a, ja), v[202:203] = (jb, b); results v[204:205]
#define SETUP "v_mov_b32 v200, %[a]\n\tv_mov_b32 v201, %[ja]\n\tv_mov_b32 v202, %[jb]\n\tv_mov_b32 v203, %[b]\n\tv_mov_b32 v204, 0\n\tv_mov_b32 v205, 0\n\ts_nop 3\n\t"
#define TAIL  "s_nop 7\n\tv_mov_b32 %[lo], v204\n\tv_mov_b32 %[hi], v205\n\t"
#define T_SPLAT  "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel:[0,1] op_sel_hi:[0,1]\n\t"      /* lo = a * b, hi = a * b */
#define T_BCAST  "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel_hi:[0,1]\n\t"                     /* lo = a * jb, hi = a * b */
#define T_PLAIN  "v_pk_mul_f32 v[204:205], v[200:201], v[202:203]\n\t"                                      /* lo = a * jb, hi = ja * b */
#define T_SWAP   "v_pk_mul_f32 v[204:205], v[200:201], v[202:203] op_sel:[1,1] op_sel_hi:[0,0]\n\t"      /* lo = ja * b, hi = a * jb */
#define MF "v_mfma_f32_32x32x16_f16 v[228:243], v[220:223], v[224:227], v[228:243]\n\t"

#define VARIANT(NAME, PRE, T)                                                                                 \
    __device__ __forceinline__ void NAME(float a, float ja, float jb, float b, float& lo, float& hi)         \
    {                                                                                                         \
        asm volatile(SETUP PRE T TAIL : [lo] "=v"(lo), [hi] "=v"(hi) : [a] "v"(a), [ja] "v"(ja), [jb] "v"(jb), [b] "v"(b) : CLOB); \
    }
VARIANT(t_splat, "", T_SPLAT) VARIANT(t_bcast, "", T_BCAST) VARIANT(t_plain, "", T_PLAIN) VARIANT(t_swap, "", T_SWAP)
// the same right behind the wave's OWN MFMA (still executing)
VARIANT(m_splat, MF, T_SPLAT) VARIANT(m_bcast, MF, T_BCAST)

__device__ __forceinline__ void mfma_burst(int n)
{
    for (int i = 0; i < n; ++i) asm volatile(MF MF ::: CLOB);
}

constexpr int NT = 6;
struct Sample { unsigned test, lane, lo, hi, a, ja, jb, b; };
__global__ __launch_bounds__(256) void probe(unsigned* mism /*[NT][2 lanes lo/hi][4 rows]*/, Sample* smp, unsigned* nsmp, int iters, unsigned seed, int burst)
{
    extern __shared__ float dyn[];
    const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4;
    unsigned s = seed ^ (blockIdx.x * 7919u + tid * 104729u);
    unsigned bad[NT][2];
    for (int v = 0; v < NT; ++v) bad[v][0] = bad[v][1] = 0;
    asm volatile("v_mov_b32 v220, 1.0\n\tv_mov_b32 v221, 1.0\n\tv_mov_b32 v222, 1.0\n\tv_mov_b32 v223, 1.0\n\tv_mov_b32 v224, 0\n\tv_mov_b32 v225, 0\n\tv_mov_b32 v226, 0\n\tv_mov_b32 v227, 0" ::: CLOB);
    for (int it = 0; it < iters; ++it) {
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) * (1.0f / 4096.0f) + 0.5f; };
        const float a = rnd(), ja = rnd() + 50.0f, jb = rnd() + 100.0f, b = rnd();
        // waves drift apart: a wave- and iteration-dependent number of MFMAs before the test
        mfma_burst(burst + ((s >> 20) & 3));
        float lo, hi;
#define CHK(I, F, ELO, EHI) { F(a, ja, jb, b, lo, hi); const bool bl = lo != (ELO), bh = hi != (EHI); bad[I][0] += bl; bad[I][1] += bh; \
                              if (bl || bh) { const unsigned k = atomicAdd(nsmp, 1u); if (k < 256) smp[k] = Sample{I, (unsigned)lane, __float_as_uint(lo), __float_as_uint(hi), \
                                              __float_as_uint(a), __float_as_uint(ja), __float_as_uint(jb), __float_as_uint(b)}; } }
        CHK(0, t_splat, a * b, a * b) CHK(1, t_bcast, a * jb, a * b) CHK(2, t_plain, a * jb, ja * b) CHK(3, t_swap, ja * b, a * jb)
        CHK(4, m_splat, a * b, a * b) CHK(5, m_bcast, a * jb, a * b)
    }
    for (int v = 0; v < NT; ++v)
        for (int l = 0; l < 2; ++l)
            if (bad[v][l]) atomicAdd(&mism[(v * 2 + l) * 4 + row], bad[v][l]);
    if (dyn[tid] == 12345.0f) mism[0] = 1;      // keeps the dynamic LDS (occupancy control) alive
}

int main(int argc, char** argv)
{
    const char* name[NT] = {"splat  op_sel:[0,1] op_sel_hi:[0,1]", "bcast  op_sel_hi:[0,1]", "plain", "swap   op_sel:[1,1] op_sel_hi:[0,0]",
                            "splat right behind the wave's own MFMA", "bcast right behind the wave's own MFMA"};
    unsigned *d, *nsmp; Sample* smp;
    hipMalloc(&d, NT * 8 * sizeof(unsigned)); hipMalloc(&nsmp, 4); hipMalloc(&smp, 256 * sizeof(Sample));
    const int iters = argc > 1 ? atoi(argv[1]) : 3000;
    for (int lds_kb : {0, 20, 52}) {                 // 8, 8 (LDS allows 8), 3 workgroups per CU
        for (int burst : {0, 2, 6}) {
            hipMemset(d, 0, NT * 8 * sizeof(unsigned)); hipMemset(nsmp, 0, 4);
            hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
            hipLaunchKernelGGL(probe, dim3(256 * 8), dim3(256), lds_kb * 1024, 0, d, smp, nsmp, iters, 777u + burst, burst);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            std::vector<unsigned> h(NT * 8); unsigned n = 0; std::vector<Sample> hs(256);
            hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(&n, nsmp, 4, hipMemcpyDeviceToHost); hipMemcpy(hs.data(), smp, 256 * sizeof(Sample), hipMemcpyDeviceToHost);
            printf("# dynamic LDS %d KB, MFMA burst %d..%d before each test, %d iterations x %d waves: wrong results per 16-lane row (lo lane | hi lane)\n", lds_kb, 2 * burst, 2 * burst + 6, iters, 256 * 8 * 4);
            for (int v = 0; v < NT; ++v) {
                const unsigned* m = &h[v * 8];
                printf("%-44s lo: %8u %8u %8u %8u | hi: %8u %8u %8u %8u%s\n", name[v], m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7],
                       (m[0] | m[1] | m[2] | m[3] | m[4] | m[5] | m[6] | m[7]) ? "   <-- WRONG" : "");
            }
            for (unsigned k = 0; k < (n < 8 ? n : 8); ++k) {
                const Sample& q = hs[k];
                auto f = [](unsigned u) { union { unsigned i; float x; } c; c.i = u; return c.x; };
                printf("   sample: test %u lane %u: lo %.6g hi %.6g | a %.6g ja %.6g jb %.6g b %.6g | a*b %.6g a*jb %.6g ja*b %.6g ja*jb %.6g\n", q.test, q.lane, f(q.lo), f(q.hi),
                       f(q.a), f(q.ja), f(q.jb), f(q.b), f(q.a) * f(q.b), f(q.a) * f(q.jb), f(q.ja) * f(q.b), f(q.ja) * f(q.jb));
            }
        }
    }
    return 0;
}
