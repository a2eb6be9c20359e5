// Micro-benchmark: issue cost of single VALU instructions (8 independent chains per wave, asm-pinned), 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 op_rates.hip -o op_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void k(float* out, long long* cyc, float seed) {
    float v0 = seed + threadIdx.x * 0.001f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    float w0 = 0.5f, w1 = 0.25f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v4, v5}, p3 = {v6, v7}, q = {1.0001f, 0.9999f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int o = 0; o < 256; ++o) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v##i) : "v"(w0), "v"(w1));
                REP8(X)
#undef X
            }
            if (OP == 1) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(v##i) : "v"(w0), "v"(w1));
                REP8(X)
#undef X
            }
            if (OP == 2) {
#define X(i) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(v##i) : "v"(w0), "v"(w1));
                REP8(X)
#undef X
            }
            if (OP == 3) {
#define X(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v##i) : "v"(w0));
                REP8(X)
#undef X
            }
            if (OP == 4) {
#define X(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(v##i) : "v"(w0));
                REP8(X)
#undef X
            }
            if (OP == 5) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(q));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p1) : "v"(q));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2) : "v"(q));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p3) : "v"(q));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(q));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p1) : "v"(q));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2) : "v"(q));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p3) : "v"(q));
            }
            if (OP == 6) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(q));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p1) : "v"(q));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2) : "v"(q));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p3) : "v"(q));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(q));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p1) : "v"(q));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2) : "v"(q));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p3) : "v"(q));
            }
            if (OP == 7) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v##i));
                REP8(X)
#undef X
            }
            if (OP == 8) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v##i));
                REP8(X)
#undef X
            }
            if (OP == 9) {
#define X(i) asm volatile("v_exp_f16 %0, %0" : "+v"(v##i));
                REP8(X)
#undef X
            }
            if (OP == 10) {
#define X(i) asm volatile("v_rcp_f16 %0, %0" : "+v"(v##i));
                REP8(X)
#undef X
            }
            if (OP == 11) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v##i) : "v"(w0));
                REP8(X)
#undef X
            }
            if (OP == 12) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v##i) : "v"(w0));
                REP8(X)
#undef X
            }
            if (OP == 13) {
#define X(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v##i) : "v"(1));
                REP8(X)
#undef X
            }
            if (OP == 14) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##i) : "v"(q));
                X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int threads, float* out, long long* cyc) {
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= 256;
    printf("%-34s waves/SIMD %d: %6.2f ticks per wave-instruction (per wave)\n", name, threads / 256, avg / (256.0 * 64));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    for (int th : {256, 512}) {
        run<0>("v_fma_f32", th, out, cyc); run<12>("v_mul_f32", th, out, cyc); run<1>("v_fma_mixlo_f16 (f32 srcs)", th, out, cyc);
        run<2>("v_fma_mixhi_f16 (f16 src2)", th, out, cyc); run<3>("v_cvt_pkrtz_f16_f32", th, out, cyc); run<4>("v_and_b32", th, out, cyc);
        run<5>("v_pk_fma_f32", th, out, cyc); run<6>("v_pk_mul_f32", th, out, cyc); run<14>("v_pk_add_f32", th, out, cyc);
        run<7>("v_exp_f32", th, out, cyc); run<8>("v_rcp_f32", th, out, cyc);
        run<9>("v_exp_f16", th, out, cyc); run<10>("v_rcp_f16", th, out, cyc); run<11>("v_cndmask_b32", th, out, cyc);
        run<13>("v_ldexp_f32", th, out, cyc);
    }
    return 0;
}
