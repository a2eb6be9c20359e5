// size_gnn.hip — the linker-size predictor that runs once before a sampling chain.
//
// Reference: SizeGNN.forward (src/linker_size.py:83-91: embedding_in -> n_layers x GCL[ReLU, one edge attribute,
// normalization_factor 1, 'sum'] -> embedding_out) as SizeClassifier.forward drives it at inference
// (src/linker_size_lightning.py:83-110): fragment atoms only, edge attribute = SQUARED distance (coord2diff's
// `radial`, src/egnn.py:295-301), an edge is kept where `edge_mask.bool() & (radial < 6)`, the logits are the mean of
// embedding_out over ALL padded nodes (masked nodes contribute the output bias).
//
// One 256-thread workgroup per molecule; the compacted fragment (<= 64 atoms) lives in LDS for the whole network:
// H, P, Q, AGG [64][128] fp32.  This is one small forward per chain (a 500-step chain runs 501 denoiser forwards), so
// the arithmetic is plain fp32 FMA in the reference's summation structure - no matrix cores, nothing to tune:
//   P_i = W1[:, :128] h_i + b1,  Q_j = W1[:, 128:256] h_j,  u_ij = relu(P_i + Q_j + r_ij * W1[:, 256])
//   m_ij = relu(W2 u_ij + b2);   agg_i = sum_j keep_ij m_ij  (j ascending: deterministic)
//   h_i <- h_i + W4 relu(W3 [h_i, agg_i] + b3) + b4
// nn.BatchNorm1d (normalization='batch_norm') is an affine map in eval mode; the host folds it into W3/b3, W4/b4.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/difflinker_hip.h"

struct dl_size_model {
    dl_size_config cfg;
    float* dev = nullptr;      // packed, transposed weights
    size_t floats = 0;
};

namespace {

constexpr int H = 128;
constexpr int NF_MAX = 64;          // fragment atoms per molecule held in LDS
constexpr int IN_MAX = 16;
constexpr int OUT_MAX = 64;
constexpr int THREADS = 256;

// packed layout (floats), all matrices transposed to [k][f] so that consecutive threads read consecutive words
//   emb_in:  WT[in][128], b[128]
//   per GCL: W1aT[128][128], W1bT[128][128], wd[128], b1[128], W2T[128][128], b2[128],
//            W3aT[128][128], W3bT[128][128], b3[128], W4T[128][128], b4[128]
//   emb_out: WT[128][out], b[out]
constexpr int G_W1A = 0, G_W1B = G_W1A + H * H, G_WD = G_W1B + H * H, G_B1 = G_WD + H, G_W2 = G_B1 + H,
              G_B2 = G_W2 + H * H, G_W3A = G_B2 + H, G_W3B = G_W3A + H * H, G_B3 = G_W3B + H * H, G_W4 = G_B3 + H,
              G_B4 = G_W4 + H * H, G_SIZE = G_B4 + H;

struct Args {
    const float* w;
    int in_nf, out_nf, n_layers;
    int B, N;
    const float* one_hot;
    const float* positions;
    const float* fragment_mask;
    const float* edge_mask;
    const float* distances;     // optional [B,N,N]: precomputed edge attribute; then edge_mask is final
    float* logits;
    int* flags;
};

// dst[a][f] = act(bias[f] + sum_k a0[a][k] W0T[k][f] (+ sum_k a1[a][k] W1T[k][f])) (+ resid[a][f]),  a < 64, f < 128
// thread: f = tid & 127, atoms a = (tid >> 7) + 2 m.  LDS reads are wave-wide broadcasts.
template <bool RELU>
__device__ void lin128(float* dst, const float* a0, const float* __restrict__ W0T, const float* a1,
                       const float* __restrict__ W1T, const float* __restrict__ bias, const float* resid, int tid) {
    const int f = tid & (H - 1), half = tid >> 7;
    float acc[NF_MAX / 2];
    const float b = bias ? bias[f] : 0.0f;
#pragma unroll
    for (int m = 0; m < NF_MAX / 2; ++m) acc[m] = b;
    for (int k = 0; k < H; ++k) {
        const float w = W0T[k * H + f];
#pragma unroll
        for (int m = 0; m < NF_MAX / 2; ++m) acc[m] = fmaf(a0[(half + 2 * m) * H + k], w, acc[m]);
    }
    if (a1 != nullptr) {
        for (int k = 0; k < H; ++k) {
            const float w = W1T[k * H + f];
#pragma unroll
            for (int m = 0; m < NF_MAX / 2; ++m) acc[m] = fmaf(a1[(half + 2 * m) * H + k], w, acc[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < NF_MAX / 2; ++m) {
        float v = RELU ? fmaxf(acc[m], 0.0f) : acc[m];
        if (resid != nullptr) v += resid[(half + 2 * m) * H + f];
        acc[m] = v;
    }
    __syncthreads();            // every read of a0 / a1 / resid done: dst may alias them
#pragma unroll
    for (int m = 0; m < NF_MAX / 2; ++m) dst[(half + 2 * m) * H + f] = acc[m];
    __syncthreads();
}

__global__ void __launch_bounds__(THREADS) size_gnn_kernel(Args p) {
    __shared__ float sH[NF_MAX * H], sP[NF_MAX * H], sQ[NF_MAX * H], sA[NF_MAX * H];
    __shared__ float sX[NF_MAX * 4];
    __shared__ float sU[4][4][H];
    __shared__ int sIdx[NF_MAX];
    __shared__ int sCount;
    const int tid = threadIdx.x, b = blockIdx.x, N = p.N;
    const int lane = tid & 63, wv = tid >> 6;

    // compact the fragment atoms (fragment_mask != 0), in padded order
    if (tid < 64) {
        int count = 0;
        for (int base = 0; base < N; base += 64) {
            const int a = base + tid;
            const bool real = (a < N) && (p.fragment_mask[size_t(b) * N + a] != 0.0f);
            const unsigned long long bal = __ballot(real);
            const int pos = count + __popcll(bal & ((1ull << tid) - 1ull));
            if (real && pos < NF_MAX) sIdx[pos] = a;
            count += __popcll(bal);
        }
        if (tid == 0) sCount = count;
    }
    for (int e = tid; e < NF_MAX * H; e += THREADS) { sH[e] = 0.0f; sP[e] = 0.0f; sQ[e] = 0.0f; sA[e] = 0.0f; }
    __syncthreads();
    const int total = sCount;
    if (total > NF_MAX) {
        if (tid == 0) p.flags[b] = 4;
        for (int o = tid; o < p.out_nf; o += THREADS) p.logits[size_t(b) * p.out_nf + o] = 0.0f;
        return;
    }
    const int n = total;

    // masked inputs (x * fragment_mask, h * fragment_mask; linker_size_lightning.py:91-92) and embedding_in
    if (tid < 4 * n) {
        const int a = tid >> 2, k = tid & 3;
        const float fm = p.fragment_mask[size_t(b) * N + sIdx[a]];
        sX[tid] = (k < 3) ? p.positions[(size_t(b) * N + sIdx[a]) * 3 + k] * fm : 0.0f;
    }
    {
        const float* WT = p.w;                       // [in][128]
        const float* be = p.w + p.in_nf * H;
        const int f = tid & (H - 1);
        for (int a = tid >> 7; a < n; a += THREADS / H) {
            const float fm = p.fragment_mask[size_t(b) * N + sIdx[a]];
            const float* hin = p.one_hot + (size_t(b) * N + sIdx[a]) * p.in_nf;
            float acc = be[f];
            for (int k = 0; k < p.in_nf; ++k) acc = fmaf(hin[k] * fm, WT[k * H + f], acc);
            sH[a * H + f] = acc;
        }
    }
    __syncthreads();

    const float* g = p.w + p.in_nf * H + H;
    for (int layer = 0; layer < p.n_layers; ++layer, g += G_SIZE) {
        // projections of the first edge layer
        lin128<false>(sP, sH, g + G_W1A, nullptr, nullptr, g + G_B1, nullptr, tid);
        lin128<false>(sQ, sH, g + G_W1B, nullptr, nullptr, nullptr, nullptr, tid);
        // edge pass: wave wv owns the receiving atoms i = wv, wv + 4, ...; 4 senders j at a time
        const float wd0 = g[G_WD + lane], wd1 = g[G_WD + lane + 64];
        const float b20 = g[G_B2 + lane], b21 = g[G_B2 + lane + 64];
        const float* __restrict__ W2T = g + G_W2;
        for (int i = wv; i < n; i += 4) {
            const float p0 = sP[i * H + lane], p1 = sP[i * H + lane + 64];
            const float xi0 = sX[4 * i], xi1 = sX[4 * i + 1], xi2 = sX[4 * i + 2];
            const size_t erow = (size_t(b) * N + sIdx[i]) * N;
            float s0 = 0.0f, s1 = 0.0f;
            for (int j0 = 0; j0 < n; j0 += 4) {
                bool keep[4];
                bool any = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = j0 + q;
                    keep[q] = false;
                    float r = 0.0f;
                    if (j < n) {
                        const size_t e = erow + sIdx[j];
                        if (p.distances != nullptr) {
                            r = p.distances[e];
                            keep[q] = p.edge_mask[e] != 0.0f;
                        } else {
                            const float d0 = xi0 - sX[4 * j], d1 = xi1 - sX[4 * j + 1], d2 = xi2 - sX[4 * j + 2];
                            r = d0 * d0 + d1 * d1 + d2 * d2;                      // coord2diff `radial` (egnn.py:298)
                            keep[q] = (p.edge_mask[e] != 0.0f) && (r < 6.0f);     // linker_size_lightning.py:107
                        }
                        const int jj = j;
                        sU[wv][q][lane] = fmaxf(p0 + sQ[jj * H + lane] + r * wd0, 0.0f);
                        sU[wv][q][lane + 64] = fmaxf(p1 + sQ[jj * H + lane + 64] + r * wd1, 0.0f);
                    }
                    any = any || keep[q];
                }
                if (!any) continue;                                              // wave-uniform
                float acc[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[q][0] = b20; acc[q][1] = b21; }
                for (int k = 0; k < H; ++k) {
                    const float w0 = W2T[k * H + lane], w1 = W2T[k * H + lane + 64];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float u = sU[wv][q][k];
                        acc[q][0] = fmaf(u, w0, acc[q][0]);
                        acc[q][1] = fmaf(u, w1, acc[q][1]);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (keep[q]) { s0 += fmaxf(acc[q][0], 0.0f); s1 += fmaxf(acc[q][1], 0.0f); }
                }
            }
            sA[i * H + lane] = s0;
            sA[i * H + lane + 64] = s1;
        }
        __syncthreads();
        // node MLP + residual (node_mask = fragment_mask = 1 on every compacted atom)
        lin128<true>(sP, sH, g + G_W3A, sA, g + G_W3B, g + G_B3, nullptr, tid);
        lin128<false>(sH, sP, g + G_W4, nullptr, nullptr, g + G_B4, sH, tid);
        // rows >= n of sH must stay zero for the next layer's broadcast GEMMs (they received bias terms)
        for (int e = n * H + tid; e < NF_MAX * H; e += THREADS) sH[e] = 0.0f;
        __syncthreads();
    }

    // embedding_out and the mean over all N padded nodes: masked nodes hold h = 0 -> they contribute the bias
    const float* WoT = g;                            // [128][out]
    const float* bo = g + H * p.out_nf;
    int nan = 0;
    for (int o = tid; o < p.out_nf; o += THREADS) {
        float sum = 0.0f;
        for (int a = 0; a < n; ++a) {
            float acc = bo[o];
            for (int k = 0; k < H; ++k) acc = fmaf(sH[a * H + k], WoT[k * p.out_nf + o], acc);
            sum += acc;
        }
        sum += float(N - n) * bo[o];
        const float v = sum / float(N);
        p.logits[size_t(b) * p.out_nf + o] = v;
        if (v != v) nan = 1;
    }
    if (tid == 0) p.flags[b] = 0;
    __syncthreads();
    if (nan) atomicOr(&p.flags[b], 1);
}

int32_t check(const dl_size_config* c) {
    if (!c) return DL_ERR_BAD_ARG;
    if (c->hidden_nf != H) return DL_ERR_UNSUPPORTED;
    if (c->in_node_nf < 1 || c->in_node_nf > IN_MAX) return DL_ERR_UNSUPPORTED;
    if (c->out_node_nf < 1 || c->out_node_nf > OUT_MAX) return DL_ERR_UNSUPPORTED;
    if (c->n_layers < 1 || c->n_layers > 64) return DL_ERR_UNSUPPORTED;
    return DL_OK;
}

void transpose_into(float* dst, const float* src, int rows, int ld, int col0, int cols) {
    // src [rows][ld] (nn.Linear weight [out][in]) -> dst[k][f] = src[f][col0 + k],  k < cols, f < rows
    for (int k = 0; k < cols; ++k)
        for (int f = 0; f < rows; ++f) dst[size_t(k) * rows + f] = src[size_t(f) * ld + col0 + k];
}

}  // namespace

extern "C" {

int32_t dl_size_model_num_tensors(const dl_size_config* cfg) {
    if (check(cfg) != DL_OK) return check(cfg);
    return 4 + 8 * cfg->n_layers;
}

int32_t dl_size_max_fragment_atoms(void) { return NF_MAX; }

int32_t dl_size_model_create(const dl_size_config* cfg, const void* const* tensors, int32_t n_tensors,
                             dl_size_model** out) {
    const int32_t st = check(cfg);
    if (st != DL_OK) return st;
    if (!tensors || !out || n_tensors != 4 + 8 * cfg->n_layers) return DL_ERR_BAD_ARG;
    for (int i = 0; i < n_tensors; ++i)
        if (!tensors[i]) return DL_ERR_BAD_ARG;
    const int in = cfg->in_node_nf, on = cfg->out_node_nf, L = cfg->n_layers;
    const size_t total = size_t(in) * H + H + size_t(L) * G_SIZE + size_t(H) * on + on;
    std::vector<float> host(total, 0.0f);
    float* w = host.data();
    const float* const* t = reinterpret_cast<const float* const*>(tensors);
    int ti = 0;
    transpose_into(w, t[ti++], H, in, 0, in);                       // embedding_in.weight [128][in]
    std::memcpy(w + size_t(in) * H, t[ti++], H * sizeof(float));
    float* g = w + size_t(in) * H + H;
    for (int l = 0; l < L; ++l, g += G_SIZE) {
        const float* w1 = t[ti++];                                  // edge_mlp.0.weight [128][257]
        transpose_into(g + G_W1A, w1, H, 2 * H + 1, 0, H);
        transpose_into(g + G_W1B, w1, H, 2 * H + 1, H, H);
        for (int f = 0; f < H; ++f) g[G_WD + f] = w1[size_t(f) * (2 * H + 1) + 2 * H];
        std::memcpy(g + G_B1, t[ti++], H * sizeof(float));
        transpose_into(g + G_W2, t[ti++], H, H, 0, H);              // edge_mlp.2
        std::memcpy(g + G_B2, t[ti++], H * sizeof(float));
        const float* w3 = t[ti++];                                  // node_mlp.0.weight [128][256]
        transpose_into(g + G_W3A, w3, H, 2 * H, 0, H);
        transpose_into(g + G_W3B, w3, H, 2 * H, H, H);
        std::memcpy(g + G_B3, t[ti++], H * sizeof(float));
        transpose_into(g + G_W4, t[ti++], H, H, 0, H);              // node_mlp.{2|3}
        std::memcpy(g + G_B4, t[ti++], H * sizeof(float));
    }
    transpose_into(g, t[ti++], on, H, 0, H);                        // embedding_out.weight [out][128] -> [128][out]
    std::memcpy(g + size_t(H) * on, t[ti++], on * sizeof(float));

    dl_size_model* m = new (std::nothrow) dl_size_model();
    if (!m) return DL_ERR_ALLOC;
    m->cfg = *cfg;
    m->floats = total;
    if (hipMalloc(reinterpret_cast<void**>(&m->dev), total * sizeof(float)) != hipSuccess) {
        delete m;
        return DL_ERR_ALLOC;
    }
    if (hipMemcpy(m->dev, host.data(), total * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(m->dev);
        delete m;
        return DL_ERR_HIP;
    }
    *out = m;
    return DL_OK;
}

void dl_size_model_destroy(dl_size_model* m) {
    if (!m) return;
    if (m->dev) (void)hipFree(m->dev);
    delete m;
}

int32_t dl_size_gnn_forward(const dl_size_model* m, int32_t B, int32_t N, const float* one_hot, const float* positions,
                            const float* fragment_mask, const float* edge_mask, const float* distances, float* logits,
                            int32_t* flags, void* stream) {
    if (!m || B < 0 || N < 1 || !one_hot || !fragment_mask || !edge_mask || !logits || !flags) return DL_ERR_BAD_ARG;
    if (!positions && !distances) return DL_ERR_BAD_ARG;
    if (B == 0) return DL_OK;
    Args a;
    a.w = m->dev;
    a.in_nf = m->cfg.in_node_nf; a.out_nf = m->cfg.out_node_nf; a.n_layers = m->cfg.n_layers;
    a.B = B; a.N = N;
    a.one_hot = one_hot; a.positions = positions; a.fragment_mask = fragment_mask; a.edge_mask = edge_mask;
    a.distances = distances; a.logits = logits; a.flags = flags;
    hipLaunchKernelGGL(size_gnn_kernel, dim3(B), dim3(THREADS), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? DL_OK : DL_ERR_HIP;
}

}  // extern "C"
