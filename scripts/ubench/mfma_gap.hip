// Micro-benchmark: a back-to-back MFMA stream in waves 0-3 blocks the VALU of waves 4-7 (same SIMDs) completely
// (mphase.hip).  Does the partner get through when the MFMA wave WAITS between its MFMAs instead of standing at the issue
// stage?  Waves 0-3: 48 x { v_mfma_f32_32x32x16_f16 ; <gap> }, waves 4-7: 144 v_exp + 336 v_fmac per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define N_ITER 2048
template <int GAP, int PARTNER>     // GAP 0 none, 1 s_nop 7, 2 s_nop 15, 3 2 x s_nop 15, 4 s_sleep 1, 5 4 independent v_mov, 6 s_nop 3
__global__ void __launch_bounds__(512) k(float* out, float seed) {
    floatx16 acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = seed * i;
    half8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)(seed + i); bv[i] = (_Float16)(seed - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * 0.01f + threadIdx.x * 0.0001f + i * 0.001f;
    const float c1 = 0.9999f + seed * 1e-9f, c2 = 1e-4f * seed;
    const int w = threadIdx.x >> 6;
    if (w >= 4) {
        if (PARTNER) {
            for (int o = 0; o < N_ITER; ++o) {
#pragma unroll
                for (int m = 0; m < 480; ++m) {
                    const int i = m & 7;
                    if ((m % 10) < 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                    else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
                }
            }
        }
    } else {
        for (int o = 0; o < N_ITER; ++o) {
#pragma unroll
            for (int m = 0; m < 48; ++m) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(av), "v"(bv));
                if (GAP == 1) asm volatile("s_nop 7");
                if (GAP == 2) asm volatile("s_nop 15");
                if (GAP == 3) asm volatile("s_nop 15\n s_nop 15");
                if (GAP == 4) asm volatile("s_sleep 1");
                if (GAP == 5) asm volatile("v_mov_b32 %0, %0\n v_mov_b32 %1, %1\n v_mov_b32 %2, %2\n v_mov_b32 %3, %3" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
                if (GAP == 6) asm volatile("s_nop 3");
            }
        }
    }
    float s_ = 0;
    for (int i = 0; i < 8; ++i) s_ += v[i];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) s_ += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s_;
}
template <int GAP, int PARTNER>
void run(const char* name) {
    static float* out = nullptr;
    if (!out) hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<GAP, PARTNER>), dim3(256), dim3(512), 0, 0, out, 1.0f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<GAP, PARTNER>), dim3(256), dim3(512), 0, 0, out, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %s: %8.1f ns per iteration\n", name, PARTNER ? "with VALU partner" : "MFMA waves alone ", ms * 1e6f / N_ITER);
}
int main() {
    run<0, 0>("no gap"); run<0, 1>("no gap");
    run<6, 0>("s_nop 3"); run<6, 1>("s_nop 3");
    run<1, 0>("s_nop 7"); run<1, 1>("s_nop 7");
    run<2, 0>("s_nop 15"); run<2, 1>("s_nop 15");
    run<3, 0>("2 x s_nop 15"); run<3, 1>("2 x s_nop 15");
    run<4, 0>("s_sleep 1"); run<4, 1>("s_sleep 1");
    run<5, 0>("4 v_mov"); run<5, 1>("4 v_mov");
    return 0;
}
