// Micro-benchmark: how one SIMD of gfx950 shares its VALU and matrix pipe between 1, 2 and 4 resident waves.
//   A: VALU only (independent v_fma_f32 / v_exp_f32 chains), wall time per wave-instruction vs waves per SIMD
//   B: one v_mfma_f32_32x32x16_f16 followed by K independent VALU instructions, repeated (in-wave interleave), 1 and 2 waves per SIMD
// Times are wall-clock (HIP events) over 256 workgroups (one per CU); cycles quoted at the clock rocm-smi would show are left
// to the reader: the table prints ns.     hipcc --offload-arch=gfx950 -O3 simd_model.hip -o simd_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void __launch_bounds__(1024) valu_only(float* out, int iters) {
    float v0 = threadIdx.x * 0.001f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    const float w0 = 0.999f, w1 = 0.001f;
    for (int o = 0; o < iters; ++o) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v##i) : "v"(w0), "v"(w1));
                REP8(X)
#undef X
            } else {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v##i));
                REP8(X)
#undef X
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

// K plain VALU (and KT transcendentals) after every MFMA; two accumulators alternate
template <int K, int KT>
__global__ void __launch_bounds__(512) mfma_valu(float* out, int iters) {
    floatx16 acc0 = {}, acc1 = {};
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x & 7) + i); b[i] = (_Float16)(0.02f * i); }
    float v0 = threadIdx.x * 0.001f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    const float w0 = 0.999f, w1 = 0.001f;
    for (int o = 0; o < iters; ++o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                switch (k & 7) {
                    case 0: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(w0), "v"(w1)); break;
                    case 1: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(w0), "v"(w1)); break;
                    case 2: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(w0), "v"(w1)); break;
                    case 3: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(w0), "v"(w1)); break;
                    case 4: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v4) : "v"(w0), "v"(w1)); break;
                    case 5: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v5) : "v"(w0), "v"(w1)); break;
                    case 6: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v6) : "v"(w0), "v"(w1)); break;
                    default: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v7) : "v"(w0), "v"(w1)); break;
                }
            }
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                if (k & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v1));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(v0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same with KP packed fp32 instructions (v_pk_fma_f32: two values each) in place of the plain ones
template <int KP, int KT>
__global__ void __launch_bounds__(512) mfma_pk(float* out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    floatx16 acc0 = {}, acc1 = {};
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x & 7) + i); b[i] = (_Float16)(0.02f * i); }
    f2 p0 = {threadIdx.x * 0.001f, 1.0f}, p1 = {2.0f, 3.0f}, p2 = {4.0f, 5.0f}, p3 = {6.0f, 7.0f};
    const f2 q = {0.999f, 0.001f};
    float v0 = 0.5f, v1 = 0.25f;
    for (int o = 0; o < iters; ++o) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                switch (k & 3) {
                    case 0: asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(q)); break;
                    case 1: asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p1) : "v"(q)); break;
                    case 2: asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2) : "v"(q)); break;
                    default: asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p3) : "v"(q)); break;
                }
            }
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                if (k & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v1));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(v0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = v0 + v1 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 4096;
    for (int th : {256, 512, 1024}) {
        const double n = double(iters) * 64;
        double ms = time_ms([&] { hipLaunchKernelGGL(valu_only<0>, dim3(256), dim3(th), 0, 0, out, iters); });
        printf("A  v_fma_f32 only, %d waves/SIMD: %6.2f ns per wave-instruction per wave, %6.2f ns per instruction per SIMD\n", th / 256, ms * 1e6 / n, ms * 1e6 / n / (th / 256));
        ms = time_ms([&] { hipLaunchKernelGGL(valu_only<1>, dim3(256), dim3(th), 0, 0, out, iters); });
        printf("A  v_exp_f32 only, %d waves/SIMD: %6.2f ns per wave-instruction per wave, %6.2f ns per instruction per SIMD\n", th / 256, ms * 1e6 / n, ms * 1e6 / n / (th / 256));
    }
    const int it2 = 8192;
#define RUNB(K, KT) for (int th : {256, 512}) { \
        double ms = time_ms([&] { hipLaunchKernelGGL((mfma_valu<K, KT>), dim3(256), dim3(th), 0, 0, out, it2); }); \
        printf("B  MFMA + %2d fma + %d exp, %d waves/SIMD: %7.2f ns per MFMA per wave (%7.2f ns per MFMA per SIMD)\n", K, KT, th / 256, ms * 1e6 / (double(it2) * 4), ms * 1e6 / (double(it2) * 4) / (th / 256)); }
    RUNB(0, 0) RUNB(2, 0) RUNB(4, 0) RUNB(6, 0) RUNB(8, 0) RUNB(12, 0) RUNB(16, 0) RUNB(24, 0) RUNB(6, 2) RUNB(12, 4) RUNB(14, 5)
#define RUNP(KP, KT) for (int th : {256, 512}) { \
        double ms = time_ms([&] { hipLaunchKernelGGL((mfma_pk<KP, KT>), dim3(256), dim3(th), 0, 0, out, it2); }); \
        printf("C  MFMA + %2d pk_fma + %d exp, %d waves/SIMD: %7.2f ns per MFMA per wave (%7.2f ns per MFMA per SIMD)\n", KP, KT, th / 256, ms * 1e6 / (double(it2) * 4), ms * 1e6 / (double(it2) * 4) / (th / 256)); }
    RUNP(2, 0) RUNP(4, 0) RUNP(6, 0) RUNP(8, 0) RUNP(3, 2) RUNP(3, 3) RUNP(6, 5)
    return 0;
}
