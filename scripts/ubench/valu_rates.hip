// Micro-benchmark: issue cost (shader cycles per wave64 instruction) of the VALU / transcendental / LDS-permute
// instructions the edge pass is made of, with 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_INNER 64
#define N_OUTER 256

template <int OP>
__global__ void k(float* out, long long* cyc, float seed) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * 0.001f + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int o = 0; o < N_OUTER; ++o) {
#pragma unroll
        for (int r = 0; r < N_INNER / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) v[i] = __builtin_amdgcn_exp2f(v[i]);
                if (OP == 1) v[i] = __builtin_amdgcn_rcpf(v[i]);
                if (OP == 2) v[i] = fmaf(v[i], 1.0001f, 0.5f);
                if (OP == 3) v[i] = v[i] * 1.0001f;
                if (OP == 4) { auto h = __builtin_amdgcn_cvt_pkrtz(v[i], v[(i + 1) & 7]); v[i] += float(h[0]); }
                if (OP == 5) v[i] = __shfl_xor(v[i], 16);
                if (OP == 6) v[i] = __builtin_amdgcn_sqrtf(v[i]);
            }
        }
        if (OP == 7) {   // packed fma: 4 x v_pk_fma_f32 = 8 values
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int r = 0; r < N_INNER / 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    f2 a = {v[i], v[i + 1]};
                    a = __builtin_elementwise_fma(a, (f2){1.0001f, 1.0001f}, (f2){0.5f, 0.5f});
                    v[i] = a[0]; v[i + 1] = a[1];
                }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= 256;
    double per = avg / (double(N_OUTER) * N_INNER);
    printf("%-28s threads/WG %4d (waves/SIMD %d): %7.2f cycles per wave-instruction (per wave), %6.2f per SIMD-instr\n", name, threads,
           threads / 256, per, per / (threads / 256.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int th : {256, 512, 1024}) {
        run<0>("v_exp_f32", th); run<1>("v_rcp_f32", th); run<6>("v_sqrt_f32", th); run<2>("v_fma_f32", th); run<3>("v_mul_f32", th);
        run<7>("v_pk_fma_f32 (per 2 values)", th); run<4>("cvt_pkrtz+cvt_f32_f16+add", th); run<5>("ds_bpermute (shfl_xor)", th);
    }
    return 0;
}
