// Micro-benchmark: the second-layer MFMA phase of the pair loop in isolation - 48 v_mfma_f32_32x32x16_f16 per iteration
// on NACC accumulators (dependent chains of 48 / NACC), W fragments read from LDS one slab ahead, on 1 or 2 waves per SIMD;
// optionally the partner waves (w >= 4) run a VALU mix instead.  hipcc --offload-arch=gfx950 -O3 mphase.hip -o mphase
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define N_ITER 2048

template <int NACC, int LDSREADS, int PARTNER, int VPRIO = 0>   // PARTNER 0: same code, 1: VALU mix, 2: idle; VPRIO: s_setprio of the VALU waves
__global__ void __launch_bounds__(512) k(float* out, float seed, int nwaves) {
    __shared__ uint4 wbuf[4096];            // 64 KB
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) wbuf[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    floatx16 acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = seed * i;
    half8 bv[8];
    for (int s = 0; s < 8; ++s) for (int i = 0; i < 8; ++i) bv[s][i] = (_Float16)(seed + i + s);
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * 0.01f + threadIdx.x * 0.0001f + i * 0.001f;
    const float c1 = 0.9999f + seed * 1e-9f, c2 = 1e-4f * seed;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (w >= nwaves) return;
    if (PARTNER == 2 && w < 4) return;
    const uint4* Wq = wbuf + lane;
    if (PARTNER >= 1 && w >= 4) {
        if (VPRIO) __builtin_amdgcn_s_setprio(VPRIO);
        for (int o = 0; o < N_ITER; ++o) {
#pragma unroll
            for (int m = 0; m < 480; ++m) {
                const int i = m & 7;
                if ((m % 10) < 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
            }
        }
    } else {
        for (int o = 0; o < N_ITER; ++o) {
            int opq = 0; asm volatile("" : "+s"(opq));
            const uint4* Wp = Wq + opq;
            uint4 af[2][4];
            if (LDSREADS) { af[0][0] = Wp[0]; af[0][1] = Wp[64]; af[0][2] = Wp[2048]; af[0][3] = Wp[2048 + 64]; }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (LDSREADS && s < 7) {
                    af[(s + 1) & 1][0] = Wp[(s + 1) * 256]; af[(s + 1) & 1][1] = Wp[(s + 1) * 256 + 64];
                    af[(s + 1) & 1][2] = Wp[2048 + (s + 1) * 256]; af[(s + 1) & 1][3] = Wp[2048 + (s + 1) * 256 + 64];
                }
                half8 a0, a1, a2, a3;
                if (LDSREADS) {
                    a0 = __builtin_bit_cast(half8, af[s & 1][0]); a1 = __builtin_bit_cast(half8, af[s & 1][1]);
                    a2 = __builtin_bit_cast(half8, af[s & 1][2]); a3 = __builtin_bit_cast(half8, af[s & 1][3]);
                } else { a0 = bv[(s + 1) & 7]; a1 = bv[(s + 2) & 7]; a2 = bv[(s + 3) & 7]; a3 = bv[(s + 4) & 7]; }
                const half8 as[6] = {a2, a3, a0, a1, a0, a1};
#pragma unroll
                for (int m = 0; m < 6; ++m)
                    acc[(6 * s + m) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as[m], bv[s], acc[(6 * s + m) % NACC], 0, 0, 0);
            }
            if (LDSREADS) {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                for (int s = 0; s < 7; ++s) { __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); __builtin_amdgcn_sched_group_barrier(0x008, 6, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            }
        }
    }
    float s_ = 0;
    for (int i = 0; i < 8; ++i) s_ += v[i];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) s_ += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s_;
}

template <int NACC, int LDSREADS, int PARTNER, int VPRIO = 0>
void run(const char* name, int nwaves) {
    static float* out = nullptr;
    if (!out) hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDSREADS, PARTNER, VPRIO>), dim3(256), dim3(512), 0, 0, out, 1.0f, nwaves);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NACC, LDSREADS, PARTNER, VPRIO>), dim3(256), dim3(512), 0, 0, out, 1.0f, nwaves);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s waves %d: %8.1f ns per 48-MFMA phase (ideal 1 wave/SIMD at 2.4 GHz: 640 ns)\n", name, nwaves, ms * 1e6f / N_ITER);
}

int main() {
    run<4, 0, 0>("4 accumulators, register operands", 4);
    run<4, 0, 0>("4 accumulators, register operands", 8);
    run<2, 0, 0>("2 accumulators, register operands", 4);
    run<2, 0, 0>("2 accumulators, register operands", 8);
    run<1, 0, 0>("1 accumulator, register operands", 4);
    run<2, 1, 0>("2 accumulators, W from LDS one slab ahead", 4);
    run<2, 1, 0>("2 accumulators, W from LDS one slab ahead", 8);
    run<4, 1, 0>("4 accumulators, W from LDS one slab ahead", 8);
    run<2, 1, 1>("2 acc + LDS | partner: 144 exp + 336 fmac", 8);
    run<2, 0, 1>("2 acc regs | partner: 144 exp + 336 fmac", 8);
    run<2, 1, 2>("partner alone: 144 exp + 336 fmac (waves 4-7 only)", 8);
    run<4, 0, 1>("4 acc regs | partner: 144 exp + 336 fmac", 8);
    run<4, 1, 1>("4 acc + LDS | partner: 144 exp + 336 fmac", 8);
    run<1, 0, 1>("1 acc regs | partner: 144 exp + 336 fmac", 8);
    run<1, 1, 1>("1 acc + LDS | partner: 144 exp + 336 fmac", 8);
    run<2, 0, 1, 3>("2 acc regs | partner at s_setprio 3", 8);
    run<2, 1, 1, 3>("2 acc + LDS | partner at s_setprio 3", 8);
    run<4, 0, 1, 1>("4 acc regs | partner at s_setprio 1", 8);
    return 0;
}
