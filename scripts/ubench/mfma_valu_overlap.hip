// Micro-benchmark: do v_mfma_f32_32x32x16_f16 and VALU / transcendental instructions of the SAME wave (interleaved)
// and of co-resident waves on one SIMD overlap?   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip
// Prints s_memtime ticks and wall-clock ns per iteration (12 MFMAs + 12*NV VALU ops) for 1/2/3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define N_ITER 2048

// MF: MFMAs on/off; NV: VALU ops after every MFMA slot; KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_cvt_pkrtz+add
template <int MF, int NV, int KIND>
__global__ void __launch_bounds__(768) k(float* out, long long* cyc, float seed) {
    floatx16 acc[4];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = seed * i;
    half8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)(seed + i); bv[i] = (_Float16)(seed - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + threadIdx.x * 0.001f + i;
    float c1 = 1.0001f + seed * 1e-9f, c2 = 0.5f * seed, tmp = 0.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    f2 pk[4], pc1 = {c1, c1}, pc2 = {c2, c2};
    for (int i = 0; i < 4; ++i) pk[i] = (f2){v[2 * i], v[2 * i + 1]};
    f4 ld[4] = {};
    __shared__ float4 lbuf[1024];
    lbuf[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    const unsigned ldsaddr = (threadIdx.x & 63) * 16;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int o = 0; o < N_ITER; ++o) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            // asm volatile: keeps the program order exactly MFMA, NV VALU, MFMA, NV VALU, ...
            if (MF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(av), "v"(bv));
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int i = (m * NV + q) & 7;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                if (KIND == 2) asm volatile("v_cvt_pkrtz_f16_f32 %1, %0, %0\n v_cvt_f32_f16 %1, %1\n v_sub_f32 %0, %0, %1" : "+v"(v[i]), "=&v"(tmp));
                if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c2));                 // 2 VGPR sources
                if (KIND == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "s"(c1));                 // 1 VGPR + 1 SGPR
                if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(c1), "s"(c1));    // 1 VGPR + SGPR (same)
                if (KIND == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));       // 3 VGPR reads (dst)
                if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[i & 3]) : "v"(pc1), "v"(pc2));
                if (KIND == 8) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[i & 3]) : "v"(ldsaddr));
                if (KIND == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c2));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    s += tmp;
    if (KIND == 8) asm volatile("s_waitcnt lgkmcnt(0)");
    for (int i = 0; i < 4; ++i) s += pk[i][0] + pk[i][1] + ld[i][0] + ld[i][3];
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MF, int NV, int KIND>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads = 256; threads <= 768; threads += 256) {
        hipLaunchKernelGGL((k<MF, NV, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MF, NV, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(256);
        hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto c : h) avg += c; avg /= 256;
        printf("%-26s waves/SIMD %d: %8.1f ticks/iter  %8.1f ns/iter  (ticks/ns %.3f)\n", name, threads / 256, avg / N_ITER,
               ms * 1e6 / N_ITER, avg / (ms * 1e6));
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1, 0, 0>("12 mfma only");
    run<0, 8, 0>("96 fma only");
    run<1, 8, 0>("12 mfma + 96 fma");
    run<0, 16, 0>("192 fma only");
    run<1, 16, 0>("12 mfma + 192 fma");
    run<0, 2, 1>("24 exp only");
    run<1, 2, 1>("12 mfma + 24 exp");
    run<0, 4, 1>("48 exp only");
    run<1, 4, 1>("12 mfma + 48 exp");
    run<0, 2, 2>("24 (cvt_pkrtz,cvt,add) only");
    run<1, 2, 2>("12 mfma + 24 cvt-triples");
    run<0, 8, 3>("96 add(2v) only");
    run<1, 8, 3>("12 mfma + 96 add(2v)");
    run<0, 8, 4>("96 mul(1v,1s) only");
    run<1, 8, 4>("12 mfma + 96 mul(1v,1s)");
    run<0, 8, 5>("96 fma(1v,s,s) only");
    run<1, 8, 5>("12 mfma + 96 fma(1v,s,s)");
    run<0, 8, 6>("96 fmac(3v) only");
    run<1, 8, 6>("12 mfma + 96 fmac(3v)");
    run<0, 4, 7>("48 pk_fma only");
    run<1, 4, 7>("12 mfma + 48 pk_fma");
    run<0, 4, 8>("48 ds_read_b128 only");
    run<1, 4, 8>("12 mfma + 48 ds_read_b128");
    run<0, 8, 9>("96 cndmask only");
    run<1, 8, 9>("12 mfma + 96 cndmask");
    return 0;
}
