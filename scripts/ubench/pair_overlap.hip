// Micro-benchmark: what do the TWO waves that share a SIMD (waves w and w+4 of a 512-thread workgroup) overlap?
// Waves 0-3 run instruction kind XA, waves 4-7 kind XB, 64 instructions per iteration each; prints wall ns per
// iteration for A alone, B alone and both.   hipcc --offload-arch=gfx950 -O3 pair_overlap.hip -o pair_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define N_ITER 4096

enum { NONE = 0, MFMA16 = 1, MFMA32 = 2, EXP = 3, FMAC = 4, FMA3 = 5, CVT = 6, MIXLO = 7, LDS128 = 8, MUL = 9, MIX_EXP_FMAC = 10,
       MFMA16_LDS = 11 };

template <int KIND>
__device__ __forceinline__ void body(floatx16 (&acc)[4], half8& av, half8& bv, float (&v)[8], float c1, float c2, unsigned ldsaddr,
                                     float4 (&ld)[4]) {
#pragma unroll
    for (int m = 0; m < 64; ++m) {
        const int i = m & 7;
        if (KIND == MFMA16) { if ((m & 3) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[(m >> 2) & 3]) : "v"(av), "v"(bv)); }
        if (KIND == MFMA32) { if ((m & 7) == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[(m >> 3) & 3]) : "v"(c1), "v"(c2)); }
        if (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        if (KIND == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
        if (KIND == FMA3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
        if (KIND == CVT) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
        if (KIND == MIXLO) asm volatile("v_fma_mixlo_f16 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(c1), "v"(c2));
        if (KIND == MUL) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "s"(c1));
        if (KIND == LDS128) { if ((m & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(m >> 2) & 3]) : "v"(ldsaddr)); }
        if (KIND == MIX_EXP_FMAC) {   // the edge loop's mix: 1 transcendental per 2.3 plain VOP2
            if ((m % 10) < 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
        }
        if (KIND == MFMA16_LDS) {     // 16 MFMAs + 16 ds_read_b128 per iteration
            if ((m & 3) == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[(m >> 2) & 3]) : "v"(av), "v"(bv));
            if ((m & 3) == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(m >> 2) & 3]) : "v"(ldsaddr));
        }
    }
}

template <int XA, int XB>
__global__ void __launch_bounds__(512) k(float* out, float seed) {
    floatx16 acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = seed * i;
    half8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)(seed + i); bv[i] = (_Float16)(seed - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * 0.01f + threadIdx.x * 0.0001f + i * 0.001f;
    float c1 = 0.9999f + seed * 1e-9f, c2 = 1e-4f * seed;
    float4 ld[4] = {};
    __shared__ float4 lbuf[1024];
    lbuf[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    const unsigned ldsaddr = (threadIdx.x & 63) * 16;
    const int w = threadIdx.x >> 6;
    if (w < 4) {
        for (int o = 0; o < N_ITER; ++o) body<XA>(acc, av, bv, v, c1, c2, ldsaddr, ld);
    } else {
        for (int o = 0; o < N_ITER; ++o) body<XB>(acc, av, bv, v, c1, c2, ldsaddr, ld);
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += ld[i].x + ld[i].w;
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int XA, int XB>
float run1() {
    static float* out = nullptr;
    if (!out) hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<XA, XB>), dim3(256), dim3(512), 0, 0, out, 1.0f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<XA, XB>), dim3(256), dim3(512), 0, 0, out, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / N_ITER;
}

template <int XA, int XB>
void pair(const char* name) {
    const float a = run1<XA, NONE>(), b = run1<NONE, XB>(), ab = run1<XA, XB>();
    printf("%-44s A alone %7.1f  B alone %7.1f  both %7.1f ns/iter   (sum %7.1f, max %7.1f, hidden %4.0f%%)\n", name, a, b, ab,
           a + b, a > b ? a : b, 100.0f * (a + b - ab) / (a < b ? a : b));
}

int main() {
    pair<MFMA16, MFMA16>("16 mfma16 | 16 mfma16");
    pair<MFMA32, MFMA32>("8 mfma32x32x2f32 | same");
    pair<MFMA16, MFMA32>("16 mfma16 | 8 mfma_f32");
    pair<MFMA16, EXP>("16 mfma16 | 64 exp");
    pair<MFMA16, FMAC>("16 mfma16 | 64 fmac");
    pair<MFMA16, FMA3>("16 mfma16 | 64 fma(vop3)");
    pair<MFMA16, CVT>("16 mfma16 | 64 cvt_pkrtz");
    pair<MFMA16, MIXLO>("16 mfma16 | 64 fma_mixlo");
    pair<MFMA16, MUL>("16 mfma16 | 64 mul(sgpr)");
    pair<MFMA16, MIX_EXP_FMAC>("16 mfma16 | 19 exp + 45 fmac");
    pair<MFMA16_LDS, MIX_EXP_FMAC>("16 mfma16+16 lds | 19 exp + 45 fmac");
    pair<MFMA32, EXP>("8 mfma_f32 | 64 exp");
    pair<MFMA32, FMAC>("8 mfma_f32 | 64 fmac");
    pair<EXP, FMAC>("64 exp | 64 fmac");
    pair<EXP, EXP>("64 exp | 64 exp");
    pair<FMAC, FMAC>("64 fmac | 64 fmac");
    pair<FMA3, FMA3>("64 fma3 | 64 fma3");
    pair<MIXLO, MIXLO>("64 mixlo | 64 mixlo");
    pair<CVT, CVT>("64 cvt | 64 cvt");
    pair<EXP, CVT>("64 exp | 64 cvt");
    pair<LDS128, FMAC>("16 lds128 | 64 fmac");
    pair<LDS128, MFMA16>("16 lds128 | 16 mfma16");
    return 0;
}
