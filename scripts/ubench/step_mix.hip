// Micro-benchmark: the pair loop's instruction mix per wave and step - 100 v_mfma_f32_32x32x16_f16, 680 plain VALU
// (v_fmac), 256 transcendentals - on the two waves of every SIMD, under different schedules.  Which one lets the matrix
// pipe and the VALU overlap on gfx950?   hipcc --offload-arch=gfx950 -O3 step_mix.hip -o step_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define N_ITER 512
#define MFMA(a) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[(a) & 3]) : "v"(av), "v"(bv))
#define FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[(i) & 7]) : "v"(c1), "v"(c2))
#define EXPF(i) asm volatile("v_exp_f32 %0, -|%0|" : "+v"(v[(i) & 7]))
#define NOP7()  asm volatile("s_nop 7")
#define BAR()   asm volatile("s_barrier" ::: "memory")

// SCHED 0: phases (all VALU, then all MFMA), same order in both waves
//       1: in-wave interleave: per MFMA 7 fmac + 2.5 exp
//       2: phases, waves 4-7 start with the MFMA phase (anti-phase, no barriers)
//       3: ping-pong in 10 segments with s_barrier, waves 4-7 one segment behind; MFMAs back to back
//       4: as 3, s_nop 7 after every MFMA
//       5: as 1 with s_nop 1 after each chunk
//       6: VALU only     7: MFMA only (both waves)     8: as 0 with s_nop 7 after every MFMA     9: as 2 with s_nop 7
template <int SCHED>
__global__ void __launch_bounds__(512) k(float* out, float seed) {
    floatx16 acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = seed * i;
    half8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)(seed + i); bv[i] = (_Float16)(seed - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * 0.01f + threadIdx.x * 0.0001f + i * 0.001f;
    const float c1 = 0.9999f + seed * 1e-9f, c2 = 1e-4f * seed;
    const int w = threadIdx.x >> 6;
#define VALU_PART(N) _Pragma("unroll") for (int m = 0; m < (N); ++m) { if ((m % 11) < 3) EXPF(m); else FMAC(m); }
#define MFMA_PART(N, NOP) _Pragma("unroll") for (int m = 0; m < (N); ++m) { MFMA(m); if (NOP) NOP7(); }
    if (SCHED == 3 || SCHED == 4) { if (w >= 4) BAR(); }
    for (int o = 0; o < N_ITER; ++o) {
        if (SCHED == 0 || SCHED == 8) { VALU_PART(187) VALU_PART(187) VALU_PART(187) VALU_PART(187) VALU_PART(188) MFMA_PART(100, SCHED == 8) }
        if (SCHED == 2 || SCHED == 9) {
            if (w < 4) { VALU_PART(187) VALU_PART(187) VALU_PART(187) VALU_PART(187) VALU_PART(188) MFMA_PART(100, SCHED == 9) } else { MFMA_PART(100, SCHED == 9) VALU_PART(187) VALU_PART(187) VALU_PART(187) VALU_PART(187) VALU_PART(188) }
        }
        if (SCHED == 1 || SCHED == 5) {
#pragma unroll
            for (int m = 0; m < 100; ++m) {
                MFMA(m);
#pragma unroll
                for (int q = 0; q < 9; ++q) { if (q < 2 || (q == 2 && (m & 1))) EXPF(q); else FMAC(q); }
                if (SCHED == 5) asm volatile("s_nop 1");
            }
        }
        if (SCHED == 3 || SCHED == 4) {
#pragma unroll
            for (int s = 0; s < 5; ++s) { VALU_PART(187) BAR(); MFMA_PART(20, SCHED == 4) BAR(); }
        }
        if (SCHED == 6) { VALU_PART(187) VALU_PART(187) VALU_PART(187) VALU_PART(187) VALU_PART(188) }
        if (SCHED == 7) { MFMA_PART(100, false) }
    }
    if (SCHED == 3 || SCHED == 4) { if (w < 4) BAR(); }
    float s_ = 0;
    for (int i = 0; i < 8; ++i) s_ += v[i];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) s_ += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s_;
}
template <int SCHED>
void run(const char* name) {
    static float* out = nullptr;
    if (!out) hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SCHED>), dim3(256), dim3(512), 0, 0, out, 1.0f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<SCHED>), dim3(256), dim3(512), 0, 0, out, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-72s %8.1f ns per step (both waves of a SIMD one step each)\n", name, ms * 1e6f / N_ITER);
}
int main() {
    run<6>("VALU only (680 fmac + 256 exp per wave)");
    run<7>("MFMA only (100 per wave)");
    run<0>("phases: VALU then MFMA, same order in both waves");
    run<8>("phases, s_nop 7 after every MFMA");
    run<2>("phases, waves 4-7 start with their MFMA phase");
    run<9>("phases, waves 4-7 start with MFMA, s_nop 7 after every MFMA");
    run<1>("in-wave interleave: MFMA + 7 fmac + 2.5 exp");
    run<5>("in-wave interleave + s_nop 1");
    run<3>("ping-pong, 10 segments, s_barrier, half a period apart");
    run<4>("ping-pong + s_nop 7 after every MFMA");
    return 0;
}
