// Micro-benchmark (round 4): can two waves of a SIMD share the matrix pipe at ITS rate if neither ever stands at the issue stage
// with an MFMA the pipe cannot take yet?  (A wave stalled there closes the VALU port for its partner: profiles/r02 mphase,
// profiles/r04 wave_timeline.)  Stream per wave: MFMA, KP v_fma_f32 (+ KT v_exp_f32 spread among them), NOPS x `s_nop 7` - the
// padding keeps the wave away from its next MFMA while the partner's MFMA occupies the pipe.
//   hipcc --offload-arch=gfx950 -O3 pace_model.hip -o pace_model
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx4 __attribute__((ext_vector_type(4)));

template <int KP, int KT, int NOPS, int SLEEP>
__global__ void __launch_bounds__(512) paced(float* out, unsigned long long* ticks, int iters) {
    floatx16 acc0 = {}, acc1 = {};
    intx4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = 0x38003800 + (threadIdx.x & 3) + i; b[i] = 0x34003400 + i; }
    asm volatile("" : "+v"(a), "+v"(b));
    float v[8], t[4];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 4; ++i) t[i] = -0.001f * threadIdx.x - i;
    const float w0 = 0.999f, w1 = 0.001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int o = 0; o < iters; ++o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
            int done = 0;
#pragma unroll
            for (int g = 0; g < (KT > 0 ? KT : 1); ++g) {
                if (KT > 0) asm volatile("v_exp_f32 %0, %0" : "+v"(t[g & 3]));
                const int upto = (KP * (g + 1)) / (KT > 0 ? KT : 1);
#pragma unroll
                for (int k = done; k < upto; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(w0), "v"(w1));
                done = upto;
            }
#pragma unroll
            for (int n = 0; n < NOPS; ++n) asm volatile("s_nop 7");
            if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += t[i];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
    if (blockIdx.x == 0 && threadIdx.x == 256) ticks[1] = t1 - t0;
}

static float* g_out;
static unsigned long long* g_ticks;

template <int KP, int KT, int NOPS, int SLEEP>
void run() {
    const int iters = 4096;
    for (int th : {256, 512}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((paced<KP, KT, NOPS, SLEEP>), dim3(256), dim3(th), 0, 0, g_out, g_ticks, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((paced<KP, KT, NOPS, SLEEP>), dim3(256), dim3(th), 0, 0, g_out, g_ticks, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long tk[2] = {0, 0};
        hipMemcpy(tk, g_ticks, 16, hipMemcpyDeviceToHost);
        const double periods = double(iters) * 4;
        printf("MFMA + %2d fma + %d exp + %d x s_nop 7 + s_sleep %d   %d w/SIMD: %7.2f ns / period / SIMD   ticks / period: wave 0 %6.1f  wave 4 %6.1f\n",
               KP, KT, NOPS, SLEEP, th / 256, ms * 1e6 / periods / (th / 256), double(tk[0]) / periods, th == 512 ? double(tk[1]) / periods : 0.0);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}

int main() {
    hipMalloc(&g_out, 256 * 1024 * 4);
    hipMalloc(&g_ticks, 64);
#define SWEEP(KP, KT) run<KP, KT, 0, 0>(); run<KP, KT, 1, 0>(); run<KP, KT, 2, 0>(); run<KP, KT, 3, 0>(); run<KP, KT, 4, 0>(); run<KP, KT, 5, 0>(); run<KP, KT, 6, 0>(); run<KP, KT, 8, 0>();
    SWEEP(4, 1)
    SWEEP(6, 2)
    SWEEP(8, 0)
    SWEEP(12, 0)
    SWEEP(3, 2)
    run<4, 1, 0, 1>();
    run<6, 2, 0, 1>();
    return 0;
}
