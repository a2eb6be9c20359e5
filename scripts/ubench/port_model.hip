// Micro-benchmark (round 4): what ONE matrix instruction costs the VALU issue port of its SIMD, by instruction type and by the
// register file its accumulator lives in, and whether transcendental and plain VALU instructions overlap.
//
// Round 2/3 found for the pair loop's stream (two waves per SIMD, MFMA + K plain VALU repeated):  T = max(32, 21 + 2.35 K)
// cycles per MFMA - an f16 MFMA keeps the SIMD's VALU port closed for ~21 of its 32 cycles.  Questions here:
//   G1  matrix instructions alone: v_mfma_f32_32x32x16_f16, _16x16x32_f16, _32x32x16_fp8_fp8, _32x32x64_f8f6f4 (fp8 and fp4
//       operands) with the accumulator in VGPRs and in AGPRs
//   G2  the same + K independent v_fma_f32 per matrix instruction: intercept (port time of the MFMA) and slope
//   G3  VALU only: plain, transcendental, and mixes in two orders (T T p p p p p p  vs  T p p p T p p p)
//   G4  conversion instructions the fp8 variants would need (v_cvt_pk_fp8_f32, v_cvt_scalef32_pk_fp8_f32, v_perm_b32,
//       v_cvt_pk_f16_f32, v_accvgpr_read_b32) at the plain rate or not
// Wall clock over 256 workgroups (one per CU), 1 and 2 waves per SIMD; s_memtime ticks of block 0 printed beside it.
//   hipcc --offload-arch=gfx950 -O3 port_model.hip -o port_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef int intx6 __attribute__((ext_vector_type(6)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx2 __attribute__((ext_vector_type(2)));

enum { MF_NONE = 0, MF_F16 = 1, MF_F8K64 = 2, MF_F16_16 = 3, MF_FP8_LEGACY = 4, MF_F4K64 = 5 };

template <int MF, int AGPR>
__device__ __forceinline__ void matrix_op(floatx16& acc, floatx4& acc4, const intx8& a, const intx8& b) {
    if constexpr (MF == MF_F16) {
        const intx4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
        if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a4), "v"(b4));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a4), "v"(b4));
    } else if constexpr (MF == MF_F8K64) {
        if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    } else if constexpr (MF == MF_F4K64) {
        const intx4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
        if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+a"(acc) : "v"(a4), "v"(b4));
        else asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(acc) : "v"(a4), "v"(b4));
    } else if constexpr (MF == MF_F16_16) {
        const intx4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
        if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc4) : "v"(a4), "v"(b4));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4) : "v"(a4), "v"(b4));
    } else if constexpr (MF == MF_FP8_LEGACY) {
        const intx2 a2 = {a[0], a[1]}, b2 = {b[0], b[1]};
        if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_fp8_fp8 %0, %1, %2, %0" : "+a"(acc) : "v"(a2), "v"(b2));
        else asm volatile("v_mfma_f32_32x32x16_fp8_fp8 %0, %1, %2, %0" : "+v"(acc) : "v"(a2), "v"(b2));
    }
}

// plain VALU flavours
enum { OP_FMA = 0, OP_CVT_FP8 = 1, OP_CVT_SCALE_FP8 = 2, OP_PERM = 3, OP_CVT_F16 = 4, OP_CVT_PKRTZ = 5, OP_AND = 6 };
template <int OP>
__device__ __forceinline__ void plain_op(float& v, float w0, float w1) {
    if constexpr (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(w0), "v"(w1));
    else if constexpr (OP == OP_CVT_FP8) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(v) : "v"(w0), "v"(w1));
    else if constexpr (OP == OP_CVT_SCALE_FP8) asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %2" : "+v"(v) : "v"(w0), "v"(w1));
    else if constexpr (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v) : "v"(w0), "v"(w1));
    else if constexpr (OP == OP_CVT_F16) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v) : "v"(w0));
    else if constexpr (OP == OP_CVT_PKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v) : "v"(w0));
    else asm volatile("v_and_b32 %0, %0, %1" : "+v"(v) : "v"(w0));
}

// One period = one matrix instruction (two accumulators alternate) + KP plain + KT transcendental instructions.
// PAT 0: the plain block, then the transcendental block (back to back); PAT 1: evenly interleaved (T p p p T p p p ...)
template <int MF, int AGPR, int KP, int KT, int PAT, int OP>
__global__ void __launch_bounds__(512) period(float* out, unsigned long long* ticks, int iters) {
    floatx16 acc0 = {}, acc1 = {};
    floatx4 q0 = {}, q1 = {};
    intx8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + (threadIdx.x & 3) + i; b[i] = 0x34343434 + i; }
    asm volatile("" : "+v"(a), "+v"(b));       // operands live in VGPRs for the whole loop
    float v[8], t[4];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 4; ++i) t[i] = -0.001f * threadIdx.x - i;
    const float w0 = 0.999f, w1 = 0.001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int o = 0; o < iters; ++o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r & 1) matrix_op<MF, AGPR>(acc1, q1, a, b);
            else matrix_op<MF, AGPR>(acc0, q0, a, b);
            if constexpr (PAT == 0) {
#pragma unroll
                for (int k = 0; k < KP; ++k) plain_op<OP>(v[k & 7], w0, w1);
#pragma unroll
                for (int k = 0; k < KT; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(t[k & 3]));
            } else {
                // KT groups: one transcendental, then its share of the plain instructions
                int done = 0;
#pragma unroll
                for (int g = 0; g < (KT > 0 ? KT : 1); ++g) {
                    if (KT > 0) asm volatile("v_exp_f32 %0, %0" : "+v"(t[g & 3]));
                    const int upto = (KP * (g + 1)) / (KT > 0 ? KT : 1);
#pragma unroll
                    for (int k = done; k < upto; ++k) plain_op<OP>(v[k & 7], w0, w1);
                    done = upto;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += t[i] + q0[i] + q1[i];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <typename F>
double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

static float* g_out;
static unsigned long long* g_ticks;

template <int MF, int AGPR, int KP, int KT, int PAT, int OP>
void run(const char* label) {
    const int iters = 4096;
    for (int th : {256, 512}) {
        const double ms = time_ms([&] { hipLaunchKernelGGL((period<MF, AGPR, KP, KT, PAT, OP>), dim3(256), dim3(th), 0, 0, g_out, g_ticks, iters); });
        unsigned long long tk = 0;
        hipMemcpy(&tk, g_ticks, 8, hipMemcpyDeviceToHost);
        const double periods = double(iters) * 4;
        printf("%-44s KP %2d KT %d pat %d  %d w/SIMD: %7.2f ns / period / wave  %7.2f ns / period / SIMD   %7.1f ticks / period / wave\n",
               label, KP, KT, PAT, th / 256, ms * 1e6 / periods, ms * 1e6 / periods / (th / 256), double(tk) / periods);
    }
}

int main() {
    hipMalloc(&g_out, 256 * 1024 * 4);
    hipMalloc(&g_ticks, 64);
    puts("== G1: matrix instructions alone (accumulator in VGPRs / AGPRs)");
    run<MF_F16, 0, 0, 0, 0, 0>("f16 32x32x16  vgpr");
    run<MF_F16, 1, 0, 0, 0, 0>("f16 32x32x16  agpr");
    run<MF_F16_16, 0, 0, 0, 0, 0>("f16 16x16x32  vgpr");
    run<MF_F16_16, 1, 0, 0, 0, 0>("f16 16x16x32  agpr");
    run<MF_FP8_LEGACY, 0, 0, 0, 0, 0>("fp8 32x32x16 (legacy)  vgpr");
    run<MF_F8K64, 0, 0, 0, 0, 0>("fp8 32x32x64 f8f6f4  vgpr");
    run<MF_F8K64, 1, 0, 0, 0, 0>("fp8 32x32x64 f8f6f4  agpr");
    run<MF_F4K64, 0, 0, 0, 0, 0>("fp4 32x32x64 f8f6f4  vgpr");
    puts("== G2: matrix instruction + KP v_fma_f32");
#define G2(MF, AG, L) run<MF, AG, 4, 0, 0, 0>(L); run<MF, AG, 8, 0, 0, 0>(L); run<MF, AG, 12, 0, 0, 0>(L); run<MF, AG, 16, 0, 0, 0>(L); run<MF, AG, 24, 0, 0, 0>(L);
    G2(MF_F16, 0, "f16 32x32x16  vgpr")
    G2(MF_F16, 1, "f16 32x32x16  agpr")
    G2(MF_F16_16, 0, "f16 16x16x32  vgpr")
    G2(MF_F8K64, 0, "fp8 32x32x64  vgpr")
    G2(MF_F8K64, 1, "fp8 32x32x64  agpr")
    run<MF_F8K64, 0, 32, 0, 0, 0>("fp8 32x32x64  vgpr");
    run<MF_F8K64, 0, 40, 0, 0, 0>("fp8 32x32x64  vgpr");
    G2(MF_F4K64, 0, "fp4 32x32x64  vgpr")
    G2(MF_FP8_LEGACY, 0, "fp8 32x32x16 legacy  vgpr")
    puts("== G3: VALU only: plain / transcendental / mixes in two orders");
    run<MF_NONE, 0, 8, 0, 0, 0>("fma only");
    run<MF_NONE, 0, 0, 4, 0, 0>("exp only");
    run<MF_NONE, 0, 6, 2, 0, 0>("6 fma + 2 exp, blocks");
    run<MF_NONE, 0, 6, 2, 1, 0>("6 fma + 2 exp, interleaved");
    run<MF_NONE, 0, 12, 4, 0, 0>("12 fma + 4 exp, blocks");
    run<MF_NONE, 0, 12, 4, 1, 0>("12 fma + 4 exp, interleaved");
    run<MF_NONE, 0, 8, 4, 0, 0>("8 fma + 4 exp, blocks");
    run<MF_NONE, 0, 8, 4, 1, 0>("8 fma + 4 exp, interleaved");
    run<MF_NONE, 0, 4, 4, 1, 0>("4 fma + 4 exp, interleaved");
    run<MF_NONE, 0, 16, 4, 1, 0>("16 fma + 4 exp, interleaved");
    puts("== G3b: f16 MFMA + mixes in two orders");
    run<MF_F16, 0, 6, 2, 0, 0>("f16 vgpr + 6 fma + 2 exp, blocks");
    run<MF_F16, 0, 6, 2, 1, 0>("f16 vgpr + 6 fma + 2 exp, interleaved");
    run<MF_F16, 1, 6, 2, 1, 0>("f16 agpr + 6 fma + 2 exp, interleaved");
    run<MF_F16, 0, 4, 1, 1, 0>("f16 vgpr + 4 fma + 1 exp");
    run<MF_F16, 0, 3, 2, 1, 0>("f16 vgpr + 3 fma + 2 exp (epilogue mix)");
    run<MF_F8K64, 0, 18, 6, 1, 0>("fp8 K64 vgpr + 18 fma + 6 exp, interleaved");
    run<MF_F8K64, 0, 12, 4, 1, 0>("fp8 K64 vgpr + 12 fma + 4 exp, interleaved");
    puts("== G4: conversion / move instructions at the plain rate?");
    run<MF_NONE, 0, 8, 0, 0, OP_CVT_FP8>("v_cvt_pk_fp8_f32");
    run<MF_NONE, 0, 8, 0, 0, OP_CVT_SCALE_FP8>("v_cvt_scalef32_pk_fp8_f32");
    run<MF_NONE, 0, 8, 0, 0, OP_PERM>("v_perm_b32");
    run<MF_NONE, 0, 8, 0, 0, OP_CVT_F16>("v_cvt_pk_f16_f32");
    run<MF_NONE, 0, 8, 0, 0, OP_CVT_PKRTZ>("v_cvt_pkrtz_f16_f32");
    run<MF_NONE, 0, 8, 0, 0, OP_AND>("v_and_b32");
    run<MF_F16, 0, 8, 0, 0, OP_CVT_FP8>("f16 vgpr + 8 v_cvt_pk_fp8_f32");
    run<MF_F16, 0, 8, 0, 0, OP_CVT_SCALE_FP8>("f16 vgpr + 8 v_cvt_scalef32_pk_fp8_f32");
    run<MF_F16, 0, 8, 0, 0, OP_PERM>("f16 vgpr + 8 v_perm_b32");
    return 0;
}
