// Which consumers of a packed-fp32 (VOP3P) result need a wait state on gfx950?  (r04: the run-to-run glitches of the dense ALIKE head,
// DESIGN.md section 3.)  Every variant runs, in ONE asm statement on fixed registers, a producer whose destination holds a known OLD
// value, k = 0 / 1 / 2 wait states, and a consumer; the same sequence with 8 wait states is the reference.  A consumer that reads the
// result too early sees the old value in some lanes.  Mismatches are counted per 16-lane row of the wave.
//   hipcc -O3 --offload-arch=gfx950 -o vop3p_hazard vop3p_hazard.hip && ./vop3p_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define CLOB "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", \
             "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", \
             "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243"

#define SETUP \
    "v_mov_b32 v200, %[a0]\n\tv_mov_b32 v201, %[a1]\n\tv_mov_b32 v202, %[b0]\n\tv_mov_b32 v203, %[b1]\n\t" \
    "v_mov_b32 v204, %[c0]\n\tv_mov_b32 v205, %[c1]\n\tv_mov_b32 v206, %[old]\n\tv_mov_b32 v207, %[old]\n\t" \
    "v_mov_b32 v208, 0\n\tv_mov_b32 v209, 1.0\n\ts_nop 7\n\t"
#ifdef BG      // an unrelated 16-pass MFMA is executing while producer and consumer issue (build with -DBG)
#define BGM "v_mfma_f32_32x32x2_f32 v[228:243], v209, v209, v[228:243]\n\t"
#else
#define BGM ""
#endif
#define PROD_PKFMA BGM "v_pk_fma_f32 v[206:207], v[200:201], v[202:203], v[204:205]\n\t"
#define PROD_PKMUL BGM "v_pk_mul_f32 v[206:207], v[200:201], v[202:203]\n\t"
#define PROD_FMAMIX BGM "v_fma_mix_f32 v206, v200, v202, v204 op_sel_hi:[0,0,0]\n\tv_fma_mix_f32 v207, v201, v203, v205 op_sel_hi:[0,0,0]\n\t"
#define PROD_PLAIN BGM "v_fma_f32 v206, v200, v202, v204\n\tv_fma_f32 v207, v201, v203, v205\n\t"
#define W0 ""
#define W1 "s_nop 0\n\t"
#define W2 "s_nop 1\n\t"
#define W8 "s_nop 7\n\t"
#define CONS_ADD "v_add_f32 v208, v206, v207\n\t"
#define CONS_CVT "v_cvt_pk_f16_f32 v208, v206, v207\n\t"
#define CONS_PKADD "v_pk_add_f32 v[210:211], v[206:207], v[206:207]\n\ts_nop 7\n\tv_add_f32 v208, v210, v211\n\t"
#define CONS_FMAMIX "v_fma_mix_f32 v208, v206, v209, v207 op_sel_hi:[0,0,0]\n\t"
#define CONS_DSW "ds_write_b64 %[lds], v[206:207]\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b64 v[210:211], %[lds]\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 v208, v210, v211\n\t"
#define CONS_GST "global_store_dwordx2 %[gp], v[206:207], off\n\ts_waitcnt vmcnt(0)\n\tglobal_load_dwordx2 v[210:211], %[gp], off sc0 sc1\n\ts_waitcnt vmcnt(0)\n\tv_add_f32 v208, v210, v211\n\t"
#define CONS_MFMA "v_mov_b32 v212, 0\n\tv_mov_b32 v213, 0\n\tv_mov_b32 v214, 0\n\tv_mov_b32 v215, 0\n\t" /* placeholder, replaced below */
#define TAIL "s_nop 7\n\tv_mov_b32 %[out], v208\n\t"

#define VARIANT(NAME, PROD, WAIT, CONS)                                                                                      \
    __device__ __forceinline__ unsigned NAME(float a0, float a1, float b0, float b1, float c0, float c1, float old, unsigned lds, float* gp) \
    {                                                                                                                        \
        unsigned out;                                                                                                        \
        asm volatile(SETUP PROD WAIT CONS TAIL : [out] "=v"(out)                                                             \
                     : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [c0] "v"(c0), [c1] "v"(c1), [old] "v"(old),    \
                       [lds] "v"(lds), [gp] "v"(gp)                                                                          \
                     : "memory", CLOB);                                                                                      \
        return out;                                                                                                          \
    }

// the MFMA consumer: the packed result's low register as the A operand of v_mfma_f32_32x32x2_f32 (B = 1.0), accumulator row 0 read back
#undef CONS_MFMA
#define ZACC "v_mov_b32 v212, 0\n\tv_mov_b32 v213, 0\n\tv_mov_b32 v214, 0\n\tv_mov_b32 v215, 0\n\tv_mov_b32 v216, 0\n\tv_mov_b32 v217, 0\n\tv_mov_b32 v218, 0\n\tv_mov_b32 v219, 0\n\t" \
             "v_mov_b32 v220, 0\n\tv_mov_b32 v221, 0\n\tv_mov_b32 v222, 0\n\tv_mov_b32 v223, 0\n\tv_mov_b32 v224, 0\n\tv_mov_b32 v225, 0\n\tv_mov_b32 v226, 0\n\tv_mov_b32 v227, 0\n\ts_nop 7\n\t"
#define CONS_MFMA "v_mfma_f32_32x32x2_f32 v[212:227], v206, v209, v[212:227]\n\ts_nop 15\n\ts_nop 7\n\tv_add_f32 v208, v212, v220\n\t"
#define VARIANT_M(NAME, PROD, WAIT)                                                                                          \
    __device__ __forceinline__ unsigned NAME(float a0, float a1, float b0, float b1, float c0, float c1, float old, unsigned lds, float* gp) \
    {                                                                                                                        \
        unsigned out;                                                                                                        \
        asm volatile(SETUP ZACC PROD WAIT CONS_MFMA TAIL : [out] "=v"(out)                                                   \
                     : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [c0] "v"(c0), [c1] "v"(c1), [old] "v"(old),    \
                       [lds] "v"(lds), [gp] "v"(gp)                                                                          \
                     : "memory", CLOB);                                                                                      \
        return out;                                                                                                          \
    }

#define FAMILY(P, PROD)                                              \
    VARIANT(P##_add_0, PROD, W0, CONS_ADD) VARIANT(P##_add_1, PROD, W1, CONS_ADD) VARIANT(P##_add_8, PROD, W8, CONS_ADD)             \
    VARIANT(P##_cvt_0, PROD, W0, CONS_CVT) VARIANT(P##_cvt_1, PROD, W1, CONS_CVT) VARIANT(P##_cvt_8, PROD, W8, CONS_CVT)             \
    VARIANT(P##_pka_0, PROD, W0, CONS_PKADD) VARIANT(P##_pka_1, PROD, W1, CONS_PKADD) VARIANT(P##_pka_8, PROD, W8, CONS_PKADD)       \
    VARIANT(P##_mix_0, PROD, W0, CONS_FMAMIX) VARIANT(P##_mix_1, PROD, W1, CONS_FMAMIX) VARIANT(P##_mix_8, PROD, W8, CONS_FMAMIX)    \
    VARIANT(P##_dsw_0, PROD, W0, CONS_DSW) VARIANT(P##_dsw_1, PROD, W1, CONS_DSW) VARIANT(P##_dsw_8, PROD, W8, CONS_DSW)             \
    VARIANT(P##_gst_0, PROD, W0, CONS_GST) VARIANT(P##_gst_1, PROD, W1, CONS_GST) VARIANT(P##_gst_8, PROD, W8, CONS_GST)             \
    VARIANT_M(P##_mfma_0, PROD, W0) VARIANT_M(P##_mfma_1, PROD, W1) VARIANT_M(P##_mfma_2, PROD, W2) VARIANT_M(P##_mfma_8, PROD, W8)

FAMILY(pkfma, PROD_PKFMA)
FAMILY(pkmul, PROD_PKMUL)
FAMILY(fmix, PROD_FMAMIX)
FAMILY(plain, PROD_PLAIN)

constexpr int NV = 4 * 18;      // families x (6 consumers x 2 tested distances + mfma x 3)
__global__ __launch_bounds__(256) void probe(unsigned* mism /*[NV][4 rows]*/, float* scratch, int iters, unsigned seed)
{
    __shared__ double ldsbuf[256];
    const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4;
    const unsigned lds = (unsigned)(size_t)(&ldsbuf[tid]);
    float* gp = scratch + ((size_t)blockIdx.x * 256 + tid) * 2;
    unsigned s = seed ^ (blockIdx.x * 7919u + tid * 104729u);
    unsigned bad[NV];
    for (int v = 0; v < NV; ++v) bad[v] = 0;
    for (int it = 0; it < iters; ++it) {
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) * (1.0f / 4096.0f) + 0.5f; };
        const float a0 = rnd(), a1 = rnd(), b0 = rnd(), b1 = rnd(), c0 = rnd(), c1 = rnd(), old = rnd() + 100.0f;
        int v = 0;
#define CMP3(P, C) { const unsigned r = P##_##C##_8(a0, a1, b0, b1, c0, c1, old, lds, gp); \
                     bad[v++] += P##_##C##_0(a0, a1, b0, b1, c0, c1, old, lds, gp) != r; bad[v++] += P##_##C##_1(a0, a1, b0, b1, c0, c1, old, lds, gp) != r; }
#define CMPM(P) { const unsigned r = P##_mfma_8(a0, a1, b0, b1, c0, c1, old, lds, gp); \
                  bad[v++] += P##_mfma_0(a0, a1, b0, b1, c0, c1, old, lds, gp) != r; bad[v++] += P##_mfma_1(a0, a1, b0, b1, c0, c1, old, lds, gp) != r; \
                  bad[v++] += P##_mfma_2(a0, a1, b0, b1, c0, c1, old, lds, gp) != r; bad[v++] += 0; bad[v++] += 0; bad[v++] += 0; }
#define ALLC(P) CMP3(P, add) CMP3(P, cvt) CMP3(P, pka) CMP3(P, mix) CMP3(P, dsw) CMP3(P, gst) CMPM(P)
        ALLC(pkfma) ALLC(pkmul) ALLC(fmix) ALLC(plain)
    }
    for (int v = 0; v < NV; ++v)
        if (bad[v]) atomicAdd(&mism[v * 4 + row], bad[v]);
}

int main()
{
    const char* prod[4] = {"v_pk_fma_f32", "v_pk_mul_f32", "v_fma_mix_f32", "v_fma_f32 (plain)"};
    const char* cons[18] = {"v_add_f32 +0", "v_add_f32 +1", "v_cvt_pk_f16_f32 +0", "v_cvt_pk_f16_f32 +1", "v_pk_add_f32 +0", "v_pk_add_f32 +1",
                            "v_fma_mix_f32 +0", "v_fma_mix_f32 +1", "ds_write_b64 +0", "ds_write_b64 +1", "global_store_dwordx2 +0", "global_store_dwordx2 +1",
                            "v_mfma A operand +0", "v_mfma A operand +1", "v_mfma A operand +2", "-", "-", "-"};
    unsigned* d; float* scratch;
    const int blocks = 512, iters = 400;
    hipMalloc(&d, NV * 4 * sizeof(unsigned)); hipMemset(d, 0, NV * 4 * sizeof(unsigned));
    hipMalloc(&scratch, (size_t)blocks * 256 * 2 * sizeof(float));
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(probe, dim3(rep & 1 ? blocks : 64), dim3(256), 0, 0, d, scratch, iters, 12345u + rep);
    hipDeviceSynchronize();
    std::vector<unsigned> h(NV * 4);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    printf("# producer -> consumer at +k wait states: lanes that read a stale value, per 16-lane row of the wave (of %d samples per row)\n", (blocks + 64) * 2 * 256 / 4 * iters);
    for (int p = 0; p < 4; ++p)
        for (int c = 0; c < 18; ++c) {
            if (cons[c][0] == '-') continue;
            const unsigned* m = &h[(p * 18 + c) * 4];
            printf("%-18s -> %-26s rows 0..3: %8u %8u %8u %8u%s\n", prod[p], cons[c], m[0], m[1], m[2], m[3], (m[0] | m[1] | m[2] | m[3]) ? "   <-- STALE" : "");
        }
    return 0;
}
