// Micro-benchmark + numerical check of the fp16 hi/lo split of a product taken with v_fma_mix{lo,hi}_f16 (pack_layout.h:
// split_mix2) against the truncating sequence it replaced (mul, v_and, v_sub, 2 x v_cvt_pkrtz).
//   hipcc --offload-arch=gfx950 -O3 -I ../../include mix_split.hip -o mix_split && ./mix_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../difflinker_amd/csrc/pack_layout.h"

__device__ __forceinline__ void split_trunc2(float a0, float b0, float a1, float b1, unsigned& hp, unsigned& lp) {
    const float u0 = a0 * b0, u1 = a1 * b1;
    const float h0 = __uint_as_float(__float_as_uint(u0) & 0xffffe000u), h1 = __uint_as_float(__float_as_uint(u1) & 0xffffe000u);
    hp = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    lp = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(u0 - h0, u1 - h1));
}

__global__ void check(const float* a, const float* b, unsigned* hp, unsigned* lp, unsigned* hp_t, unsigned* lp_t, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    split_mix2(a[2 * i], b[2 * i], a[2 * i + 1], b[2 * i + 1], hp[i], lp[i]);
    split_trunc2(a[2 * i], b[2 * i], a[2 * i + 1], b[2 * i + 1], hp_t[i], lp_t[i]);
}

template <int OP>
__global__ void rate(float* out, long long* cyc, float seed) {
    float v[8], e[8];
    unsigned acc = 0;
    for (int i = 0; i < 8; ++i) { v[i] = seed + threadIdx.x * 0.001f + i; e[i] = 0.37f + 0.01f * i; }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int o = 0; o < 256; ++o) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                unsigned h, l;
                if (OP == 0) split_mix2(v[i], e[i], v[i + 1], e[i + 1], h, l);
                else split_trunc2(v[i], e[i], v[i + 1], e[i + 1], h, l);
                acc ^= h + l;
                asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(acc);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float h2f(unsigned short h) {
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v = (e == 0) ? ldexpf(float(m), -24) : (e == 31 ? INFINITY : ldexpf(float(m | 1024), e - 25));
    return s ? -v : v;
}

int main() {
    const int n = 1 << 16;
    std::vector<float> a(n), b(n);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return float(s >> 8) * (1.0f / 16777216.0f); };
    for (int i = 0; i < n; ++i) {
        const float mag = ldexpf(1.0f, int(rnd() * 28.0f) - 20);           // 2^-20 .. 2^8: normal, denormal and tiny fp16 results
        a[i] = (rnd() * 2.0f - 1.0f) * mag * 16.0f;
        b[i] = rnd() * 2.0f - 1.0f;
    }
    float *da, *db; unsigned *dh, *dl, *dht, *dlt;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dh, n * 2); hipMalloc(&dl, n * 2); hipMalloc(&dht, n * 2); hipMalloc(&dlt, n * 2);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(n / 2 / 256), dim3(256), 0, 0, da, db, dh, dl, dht, dlt, n);
    std::vector<unsigned> h(n / 2), l(n / 2), ht(n / 2), lt(n / 2);
    hipMemcpy(h.data(), dh, n * 2, hipMemcpyDeviceToHost); hipMemcpy(l.data(), dl, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), dht, n * 2, hipMemcpyDeviceToHost); hipMemcpy(lt.data(), dlt, n * 2, hipMemcpyDeviceToHost);
    double worst_mix = 0, worst_tr = 0, worst_mix_abs = 0, worst_tr_abs = 0;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        const double p = double(a[i]) * double(b[i]);
        const unsigned short hm = (i & 1) ? h[i / 2] >> 16 : h[i / 2] & 0xffff, lm = (i & 1) ? l[i / 2] >> 16 : l[i / 2] & 0xffff;
        const unsigned short hq = (i & 1) ? ht[i / 2] >> 16 : ht[i / 2] & 0xffff, lq = (i & 1) ? lt[i / 2] >> 16 : lt[i / 2] & 0xffff;
        const double em = fabs(double(h2f(hm)) + double(h2f(lm)) - p), et = fabs(double(h2f(hq)) + double(h2f(lq)) - p);
        if (fabs(p) >= 6.2e-5 * 2048) {        // both parts normal: relative error
            worst_mix = fmax(worst_mix, em / fabs(p)); worst_tr = fmax(worst_tr, et / fabs(p));
        } else { worst_mix_abs = fmax(worst_mix_abs, em); worst_tr_abs = fmax(worst_tr_abs, et); }
        if (!(em <= fabs(p) * 2.5e-7 + 6.0e-8)) ++bad;
    }
    printf("split of a product, %d values: worst relative error (normal range) mix %.3e, trunc %.3e; worst absolute error (denormal range) "
           "mix %.3e, trunc %.3e; mix out of bound: %d\n", n, worst_mix, worst_tr, worst_mix_abs, worst_tr_abs, bad);
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    for (int th : {256, 512}) {
        for (int op = 0; op < 2; ++op) {
            for (int rep = 0; rep < 2; ++rep) {
                if (op == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(th), 0, 0, out, cyc, 1.0f);
                else hipLaunchKernelGGL(rate<1>, dim3(256), dim3(th), 0, 0, out, cyc, 1.0f);
            }
            hipDeviceSynchronize();
            std::vector<long long> hc(256);
            hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto c : hc) avg += c; avg /= 256;
            printf("%-28s waves/SIMD %d: %7.2f s_memtime ticks per element pair (per wave)\n", op == 0 ? "v_fma_mix split (4 instr)" : "truncating split (8 instr)",
                   th / 256, avg / (256.0 * 8 * 4));
        }
    }
    return bad != 0;
}
