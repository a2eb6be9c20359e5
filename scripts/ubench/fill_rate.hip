// How fast can ONE compute unit pull an L2-resident (or MALL-resident) buffer - into registers, or into LDS by LDS-DMA?
// round 6: the per-atom phases of egnn_fc.hip stream ~450 KB per pass and compute unit; this measures the ceiling of that stream.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/fill_rate.hip -o /tmp/fill_rate && /tmp/fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %d\n", int(e_), __LINE__); return 1; } } while (0)

// mode 0: global_load_dwordx4 -> registers (every wave); 1: LDS-DMA (waves < nw issue); `bytes` per sweep, `sweeps` sweeps;
// `stride_wg`: 0 = every workgroup reads the SAME buffer (weights), else its own slice (scratch)
template <int MODE>
__global__ void __launch_bounds__(512) fill_kernel(const float4* __restrict__ src, size_t bytes, int sweeps, int nw, size_t stride_wg,
                                                    unsigned long long* ticks, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[32768];          // 128 KB ring
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const float4* base = src + (stride_wg * blockIdx.x) / 16;
    const size_t n4 = bytes / 16;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < sweeps; ++s) {
        if (MODE == 0) {
            if (w < nw)
                for (size_t i = size_t(w) * 64 + lane; i < n4; i += size_t(nw) * 64 * 8) {
                    float4 v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = base[min(i + size_t(k) * nw * 64, n4 - 1)];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
                }
        } else {
            if (w < nw) {
                const size_t pieces = n4 / 64;                            // 1 KB pieces
                for (size_t p = w; p < pieces; p += nw) {
                    const int slot = int(p & 127);                         // 128 x 1 KB ring
                    __builtin_amdgcn_global_load_lds((gptr_t)(base + p * 64 + lane), (lptr_t)(lds + slot * 256), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    if (MODE == 1) acc.x = lds[tid];
    if (acc.x == 123.456f) sink[0] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    const size_t total = size_t(64) << 20;
    float4* buf; unsigned long long* ticks; float* sink;
    CK(hipMalloc(&buf, total)); CK(hipMalloc(&ticks, 1024 * 8)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int mode, nw; size_t bytes; size_t stride; int grid; };
    std::vector<Case> cases;
    for (int grid : {64, 256})
        for (size_t kb : {320, 5760}) {
            cases.push_back({"regs, 8 waves, shared buffer", 0, 8, kb << 10, 0, grid});
            cases.push_back({"regs, 4 waves, shared buffer", 0, 4, kb << 10, 0, grid});
            cases.push_back({"lds-dma, 4 waves, shared buffer", 1, 4, kb << 10, 0, grid});
            cases.push_back({"lds-dma, 8 waves, shared buffer", 1, 8, kb << 10, 0, grid});
        }
    cases.push_back({"regs, 8 waves, OWN 64 KB slice (scratch)", 0, 8, size_t(64) << 10, size_t(64) << 10, 256});
    cases.push_back({"regs, 8 waves, OWN 192 KB slice", 0, 8, size_t(192) << 10, size_t(192) << 10, 256});
    for (const Case& c : cases) {
        const int sweeps = int((size_t(64) << 20) / c.bytes) + 4;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (c.mode == 0) hipLaunchKernelGGL(fill_kernel<0>, dim3(c.grid), dim3(512), 0, 0, buf, c.bytes, sweeps, c.nw, c.stride, ticks, sink);
            else hipLaunchKernelGGL(fill_kernel<1>, dim3(c.grid), dim3(512), 0, 0, buf, c.bytes, sweeps, c.nw, c.stride, ticks, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t[4]; CK(hipMemcpy(t, ticks, sizeof(t), hipMemcpyDeviceToHost));
        const double per_cu = double(c.bytes) * sweeps;
        printf("grid %3d  %-44s %5zu KB/sweep: %7.1f GB/s per CU (%6.2f TB/s chip), %5.1f B/tick (ticks %llu, %.3f ms)\n", c.grid, c.name, c.bytes >> 10,
               per_cu / (ms * 1e-3) / 1e9, per_cu * c.grid / (ms * 1e-3) / 1e12, per_cu / double(t[0]), t[0], ms);
    }
    return 0;
}
