// egnn_sparse.hip — MI355X (gfx950) kernels for DiffLinker's pocket-conditioned denoiser
// (reference DynamicsWithPockets.forward, src/egnn.py:470-552; radius graph :554-596; EGNN :218-238 with
// edge_mask=None).  Molecules here have N ~ 300 atoms (fragments + pocket + linker) and a SPARSE edge set
// (ligand-ligand fully connected, pocket-pocket <= 4 A, ligand-pocket <= 10 A, no self loops), so the
// LDS-resident one-workgroup-per-molecule design of egnn_fc.hip does not apply.  Structure:
//   * node features h, the first-layer projections P,Q and the coordinates live in a caller-provided HBM
//     workspace (L2/MALL-resident: 19 k atoms x 128 fp32 = 9.6 MB each at the C4 config);
//   * the radius graph is rebuilt on the GPU every forward (count -> scan -> fill), one wave per atom; each
//     atom's neighbour list is padded to whole QUADS of 8 edge slots and the lists follow each other, so an MFMA tile
//     (32 consecutive slots = 4 quads) holds the edges of up to four receiving atoms (padding: 5 % of the slots at
//     the pocket configuration; whole tiles per atom cost 22 %);
//   * the edge pass runs over tiles: first layer generated as MFMA A-fragments from gathered P_i, Q_j rows,
//     second layer [32x128]x[128x128] on v_mfma_f32_32x32x2_f32 with the weights in LDS, SiLU, and the sum
//     over the edges of each quad reduced IN REGISTERS; quads of one atom inside a tile are merged, so a tile writes
//     one partial row per receiving atom it holds (no atomics: the node kernel adds an atom's partial rows in a
//     fixed order, so the result is deterministic);
//   * the node MLP (+ the next pass's projections) is one kernel per pass over 32-atom row tiles.
// Arithmetic: fp32 MFMA or the f16x3 split scheme of egnn_fc.hip (template PREC); on this path the fp16 scales are
// local: every node-kernel workgroup scales its own 32-row tiles, every edge tile is scaled by its own max |u|.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pack_layout.h"

namespace {

constexpr int LDT = 132;                 // LDS row stride of a [32][128] fp32 tile (conflict-free ds_read_b128)
constexpr int NODE_THREADS = 256;
constexpr int EDGE_THREADS = 256;

struct PkDims {
    int B, N, V;                         // V = B*N padded atoms
    int nf, ctx, fin, D, ct;             // ct: condition_time (the time feature joins the node inputs)
    int graph_type;                      // 0: '4A', 1: 'FC-4A', 2: 'FC-10A-4A', 3: fully connected with an int8 edge mask
    const int8_t* emask;                 // graph_type 3: [B,N,N] mask values (0: no edge; the value weights the message)
    float norm_constant;
    // optional hyper-parameters (round 3; src/egnn.py:42-43,52-54 attention, :104-105 tanh, :315-319 mean)
    int attention, tanh, mean;
    float coords_range, inv_norm;
    int sin;                             // sinusoidal distance embedding (egnn.py:281-292): the edge attributes are 2 x 12 sin / cos values
};

// workspace carve-up (all offsets in bytes, 256-B aligned)
struct PkWs {
    float *H, *P, *Q, *X, *X0, *partial, *partialx, *partialA, *partialxA, *pmax, *qmax, *wgt;
    int *flags, *ntile, *tile_off, *tile_row, *col, *total;   // ntile / tile_off / tile_row count QUADS (8 edge slots)
    int* deg;                                                 // edges per receiving atom ('mean' divides by it, egnn.py:315-319)
    int* eq_tiles;                                            // tiles with a receiving atom the coordinate update keeps; count: total[1]
    size_t bytes;
};

__host__ __device__ inline size_t al256(size_t x) { return (x + 255) & ~size_t(255); }

inline PkWs carve(void* base, int B, int N) {
    const size_t V = size_t(B) * N;
    const size_t QMAX = V * (N / 8 + 1) + 4;                       // quads: every atom pads its list to a multiple of 8 slots
    const size_t TMAX = QMAX / 4 + 1;                               // MFMA tiles of 32 slots
    char* p = static_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += al256(bytes); return r; };
    PkWs w;
    w.H = reinterpret_cast<float*>(take(V * HID * 4));
    w.P = reinterpret_cast<float*>(take(V * HID * 4));
    w.Q = reinterpret_cast<float*>(take(V * HID * 4));
    w.X = reinterpret_cast<float*>(take(V * 16));
    w.X0 = reinterpret_cast<float*>(take(V * 16));
    w.pmax = reinterpret_cast<float*>(take(V * 4));
    w.qmax = reinterpret_cast<float*>(take(V * 4));
    w.flags = reinterpret_cast<int*>(take(V * 4));
    w.ntile = reinterpret_cast<int*>(take(V * 4));
    w.deg = reinterpret_cast<int*>(take(V * 4));
    w.tile_off = reinterpret_cast<int*>(take((V + 1) * 4));
    w.total = reinterpret_cast<int*>(take(256));
    w.tile_row = reinterpret_cast<int*>(take(QMAX * 4));
    w.eq_tiles = reinterpret_cast<int*>(take(TMAX * 4));
    w.col = reinterpret_cast<int*>(take(TMAX * 32 * 4));
    w.wgt = reinterpret_cast<float*>(take(TMAX * 32 * 4));     // per-edge weight (graph_type 3), 0 on padding
    // partial sums of an atom: row t of `partial` for every tile t whose FIRST quad is the atom's, row v of `partialA`
    // for the quads from the atom's first one (when that is not the first of its tile) to the end of that tile
    w.partial = reinterpret_cast<float*>(take(TMAX * HID * 4));
    w.partialx = reinterpret_cast<float*>(take(TMAX * 16));
    w.partialA = reinterpret_cast<float*>(take(V * HID * 4));
    w.partialxA = reinterpret_cast<float*>(take(V * 16));
    w.bytes = off;
    return w;
}

// words of PkWs.total: [0] quads, [1] tiles of the coordinate pass, [2] "this coordinate pass runs over EVERY tile" (written by its
// edge kernel, read by pk_xupdate_kernel), [4..6] float bits of the batch-wide maxima of |h|, |x|^2, |x0|^2 (atomicMax; never reset
// inside a forward: conservative) - what the proof behind skipping the masked coordinate sums needs (pk_edge_kernel<EQUIV>)
constexpr int TW_FULL = 2, TW_HMAX = 4, TW_X2 = 5, TW_X02 = 6;
// [7]: some f16-mode scale of this forward came from a bound beyond the fp16 range (pack_layout.h: beyond_f16_range) - the scales
// here belong to tiles, not to molecules, so pk_out_kernel reports every molecule of the call (NAN_RANGE | x | h)
constexpr int TW_RANGE = 7;
// batch-wide maximum of non-negative floats (as bits: they order like the values, NaN above all): the atomic is issued only when
// the word - read relaxed, possibly stale, i.e. LOWER - does not already hold as much; after the first few waves of a kernel nobody
// issues one (19 k atoms x 128 features of atomics on ONE address cost a forward of the pocket configuration a third of its time)
__device__ __forceinline__ void gmax_update(int* total, int word, float val) {
    unsigned* p = reinterpret_cast<unsigned*>(total) + word;
    const unsigned bits = __float_as_uint(val);
    if (bits > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, bits);
}
// atom flags
constexpr int F_REAL = 1, F_LIG = 2, F_POCK = 4, F_MOVES = 8;   // F_MOVES: linker mask != 0, the only atoms whose coordinates change

// ---------------------------------------------------------------------------------------------------
// 1. per-atom setup: masked coordinates, role flags, embedding h = We*[h_feat, t, ctx] + be
//    (egnn.py:486-512, :224).  One thread per (atom, feature).
// ---------------------------------------------------------------------------------------------------
__global__ void pk_init_kernel(PkDims d, PkWs w, const float* __restrict__ wp, const float* __restrict__ xh,
                               const float* __restrict__ t, int t_stride, const int8_t* __restrict__ node_mask,
                               const float* __restrict__ linker_mask, const float* __restrict__ context) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = gid >> 7, f = gid & (HID - 1);
    if (v >= d.V) return;
    const int b = v / d.N;
    const bool real = node_mask[v] != 0;
    const float nm = real ? 1.0f : 0.0f;
    const float* z = xh + size_t(v) * d.D;
    if (f < 4) {
        const float xv = (f < 3) ? z[f] * nm : 0.0f;
        w.X[4 * v + f] = xv;
        w.X0[4 * v + f] = xv;
    }
    if (f == 5) {
        const float n2 = (z[0] * z[0] + z[1] * z[1] + z[2] * z[2]) * nm;
        gmax_update(w.total, TW_X2, n2);
        gmax_update(w.total, TW_X02, n2);
    }
    if (f == 4) {
        bool lig = false, pock = false;
        if (d.graph_type != 3) {
            // fragment-only / pocket-only masks are the last two context channels (egnn.py:486-487)
            lig = real && ((linker_mask[v] != 0.0f) || (context[size_t(v) * d.ctx + d.ctx - 2] != 0.0f));
            pock = real && (context[size_t(v) * d.ctx + d.ctx - 1] != 0.0f);
        }
        const bool moves = real && (linker_mask == nullptr || linker_mask[v] != 0.0f);
        w.flags[v] = (real ? F_REAL : 0) | (lig ? F_LIG : 0) | (pock ? F_POCK : 0) | (moves ? F_MOVES : 0);
    }
    float acc = wp[OFF_EMB_B + f];
    const float* wrow = wp + OFF_EMB_W + f * FINP;
    for (int k = 0; k < d.fin; ++k) {
        float hin;
        if (k < d.nf) hin = z[3 + k] * nm;
        else if (k < d.nf + d.ct) hin = t[size_t(b) * t_stride];      // time feature is not masked (egnn.py:501-509)
        else hin = context[size_t(v) * d.ctx + (k - d.nf - d.ct)];
        acc = fmaf(wrow[k], hin, acc);
    }
    w.H[size_t(v) * HID + f] = acc;
    {
        float m = fabsf(acc);                                       // max |h| of the embedding, one atomic per 16 lanes
        m = fmaxf(m, dpp_mov<0xB1>(m)); m = fmaxf(m, dpp_mov<0x4E>(m)); m = fmaxf(m, dpp_mov<0x141>(m)); m = fmaxf(m, dpp_mov<0x140>(m));
        if ((threadIdx.x & 15) == 0) gmax_update(w.total, TW_HMAX, acc != acc ? acc : m);
    }
}

// edge predicate of get_dist_edges / get_dist_edges_4A (egnn.py:554-596), i != j, same molecule
__device__ __forceinline__ bool pk_adjacent(int gt, int fi, int fj, float d2) {
    if (!(fi & F_REAL) || !(fj & F_REAL)) return false;
    if (gt == 0) return d2 <= 16.0f;
    const bool li = fi & F_LIG, lj = fj & F_LIG, pi = fi & F_POCK, pj = fj & F_POCK;
    const float cut2 = (gt == 1) ? 16.0f : 100.0f;
    return (li && lj) || (pi && pj && d2 <= 16.0f) || (((li && pj) || (pi && lj)) && d2 <= cut2);
}

// 2./4. one wave per atom: count neighbours (FILL = false) or write the padded neighbour list (FILL = true)
template <bool FILL>
__global__ void pk_edges_kernel(PkDims d, PkWs w) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (v >= d.V) return;
    const int b = v / d.N;
    const int fi = w.flags[v];
    const float4 xi = *reinterpret_cast<const float4*>(w.X + 4 * v);
    int count = 0;
    const int base_tile = FILL ? w.tile_off[v] : 0;
    for (int j0 = 0; j0 < d.N; j0 += 64) {
        const int jn = j0 + lane;
        const int u = b * d.N + jn;
        bool adj = false;
        float wt = 1.0f;
        if (d.graph_type == 3) {
            // the reference's dense edge list with its mask (egnn.py:449-466 + datasets.py:366-369): every pair of the
            // molecule whose int8 mask value is non-zero - the diagonal included (value -2) - weighted by that value
            if (jn < d.N) {
                const int mv = d.emask[(size_t(b) * d.N + (v - b * d.N)) * d.N + jn];
                adj = mv != 0;
                wt = float(mv);
            }
        } else if (jn < d.N && u != v) {
            const float4 xj = *reinterpret_cast<const float4*>(w.X + 4 * u);
            const float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
            adj = pk_adjacent(d.graph_type, fi, w.flags[u], dx * dx + dy * dy + dz * dz);
        }
        const unsigned long long bal = __ballot(adj);
        if (FILL && adj) {
            const size_t slot = size_t(base_tile) * 8 + count + __popcll(bal & ((1ull << lane) - 1ull));
            w.col[slot] = u;
            w.wgt[slot] = wt;
        }
        count += __popcll(bal);
    }
    const int nt = (count + 7) >> 3;                                                                // quads of this atom
    if (!FILL) {
        if (lane == 0) { w.ntile[v] = nt; w.deg[v] = count; }
    } else {
        for (int e = count + lane; e < nt * 8; e += 64) {                                           // padding
            w.col[size_t(base_tile) * 8 + e] = -1;
            w.wgt[size_t(base_tile) * 8 + e] = 0.0f;
        }
        for (int k = lane; k < nt; k += 64) w.tile_row[base_tile + k] = v;
    }
}

// 3. exclusive scan of ntile[0..V) -> tile_off, total quad count (single workgroup)
// (also marks the slots between the last quad and the end of its tile as padding: nothing else ever writes them)
__global__ void pk_scan_kernel(int V, const int* __restrict__ ntile, int* __restrict__ tile_off, int* __restrict__ total,
                               int* __restrict__ col, float* __restrict__ wgt) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, nth = blockDim.x;
    const int per = (V + nth - 1) / nth;
    const int lo = min(tid * per, V), hi = min(lo + per, V);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += ntile[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < nth; off <<= 1) {
        const int add = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = lo; i < hi; ++i) { tile_off[i] = run; run += ntile[i]; }
    if (tid == nth - 1) { tile_off[V] = part[tid]; total[0] = part[tid]; total[1] = 0; }   // total[1]: pk_eqtiles_kernel's counter
    const int nq = part[nth - 1];                                   // every quad of every atom
    for (int e = nq * 8 + tid; e < ((nq + 3) >> 2) * 32; e += nth) { col[e] = -1; wgt[e] = 0.0f; }
}

// 4b. the tiles of the coordinate pass: the reference multiplies the coordinate sum of every atom outside the linker mask
//     by zero (egnn.py:113-116), so only tiles holding a quad of a linker atom are worth computing (12 % of them at the
//     pocket configuration).  The list order is whatever the atomics give; every tile writes its own rows, so results do
//     not depend on it.
__global__ void pk_eqtiles_kernel(PkWs w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nquads = w.total[0];
    if (t >= (nquads + 3) >> 2) return;
    bool any = false;
    for (int g = 0; g < 4; ++g) {
        const int qd = 4 * t + g;
        if (qd < nquads && (w.flags[w.tile_row[qd]] & F_MOVES)) any = true;
    }
    if (any) w.eq_tiles[atomicAdd(&w.total[1], 1)] = t;
}

// acc[32 rows x 32 features] += A[rows][k] * W'[feature][k], k = 0..127; A = LDS tile (row stride LDT), B = pre-loaded
// fragments of one unit slice.  PREC 0: fp32 MFMA, lane supplies k = 64*hh + s.  PREC 1 (f16x3): lane supplies
// k = 16*slab + 8*hh + e, A is scaled by `sa` (power of two) and split into fp16 hi+lo (see egnn_fc.hip).
// `kfac` (f16 modes, or nullptr): the A operand's features of 32-feature tile kt enter times sa * kfac[kt] (balanced packing:
// the hidden layer of the node MLP, whose second-layer columns carry the inverse factors)
template <int PREC>
__device__ __forceinline__ void gemm_lds(floatx16& acc, const float* abuf, int row, int hh, const BFrag& b, float sa,
                                         const float* __restrict__ kfac = nullptr) {
    if constexpr (PREC == 0) {
        const float4* ap = reinterpret_cast<const float4*>(abuf + row * LDT + 64 * hh);
#pragma unroll
        for (int sg = 0; sg < 16; ++sg) {
            const float4 a = ap[sg];
            acc = mfma32(a.x, b.q[sg].x, acc);
            acc = mfma32(a.y, b.q[sg].y, acc);
            acc = mfma32(a.z, b.q[sg].z, acc);
            acc = mfma32(a.w, b.q[sg].w, acc);
        }
    } else {
        const float* ap = abuf + row * LDT + 8 * hh;
#pragma unroll
        for (int slab = 0; slab < 8; ++slab) {
            const float4 a0 = *reinterpret_cast<const float4*>(ap + 16 * slab);
            const float4 a1 = *reinterpret_cast<const float4*>(ap + 16 * slab + 4);
            const float ss = kfac ? sa * kfac[slab >> 1] : sa;
            const float u[8] = {a0.x * ss, a0.y * ss, a0.z * ss, a0.w * ss, a1.x * ss, a1.y * ss, a1.z * ss, a1.w * ss};
            uint4 hi, lo;
            split8(u, hi, lo);
            const uint4 bh = __builtin_bit_cast(uint4, b.q[slab]), bl = __builtin_bit_cast(uint4, b.q[8 + slab]);
            acc = mfma_h(lo, bh, acc);
            acc = mfma_h(hi, bl, acc);
            acc = mfma_h(hi, bh, acc);
        }
    }
}

// max of non-negative floats over the workgroup, via an LDS word (zeroed before, read after a barrier)
__device__ __forceinline__ void wg_max(unsigned* slot, float val, int lane) {
    float m = val;
    m = fmaxf(m, dpp_mov<0xB1>(m));
    m = fmaxf(m, dpp_mov<0x4E>(m));
    m = fmaxf(m, dpp_mov<0x141>(m));
    m = fmaxf(m, dpp_mov<0x140>(m));
    if ((lane & 15) == 0) atomicMax(slot, __float_as_uint(m));
}

// ---------------------------------------------------------------------------------------------------
// 5. node kernel, one workgroup (4 waves = 4 feature tiles) per 32-atom row tile:
//    POST: agg = (sum of the atom's tile partials), t = SiLU(W3a' h + W3b' agg + b3'), h += W4' t + b4, masked
//          (GCL.node_model egnn.py:62-72,78-79);   PRE: P = W1a' h + b1', Q = W1b' h for the NEXT pass.
// ---------------------------------------------------------------------------------------------------
template <int PREC>
__global__ void __launch_bounds__(NODE_THREADS)
pk_node_kernel(PkDims d, PkWs w, const float* __restrict__ post, const float* __restrict__ pre_units,
               const float* __restrict__ pre_bias, const float* __restrict__ pre_scale /* [2][4]: weight scale of W1a' | W1b', per output tile */,
               const float* __restrict__ pre_ne /* [8]: 2^n of the k-slabs of the coming edge pass's hidden layer */) {
    __shared__ __attribute__((aligned(16))) float hL[32 * LDT];
    __shared__ __attribute__((aligned(16))) float aL[32 * LDT];
    __shared__ unsigned mx[4];                                    // f16x3: local |h|, |agg|, |t|, |h_new| maxima
    __shared__ unsigned rowmx[2][32];                             // f16x3: per-row max |P|, |Q| (bounds the edge A-fragments)
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, c = lane & 31, hh = lane >> 5;
    const int row0 = blockIdx.x * 32;
    if (tid < 4) mx[tid] = 0u;
    if (tid < 64) rowmx[tid >> 5][tid & 31] = 0u;
    __syncthreads();
    // stage h rows (and the aggregate) into LDS
    float hmax = 0.0f, amax = 0.0f;
    for (int e = tid; e < 32 * 32; e += NODE_THREADS) {
        const int r = e >> 5, q = e & 31;
        const int v = row0 + r;
        float4 hv = make_float4(0.f, 0.f, 0.f, 0.f), av = hv;
        if (v < d.V) {
            hv = *reinterpret_cast<const float4*>(w.H + size_t(v) * HID + 4 * q);
            if (post) {
                const int q0 = w.tile_off[v], nq = w.ntile[v];      // the atom's quads
                if (nq > 0) {                                       // fixed order: deterministic
                    if (q0 & 3) {
                        const float4 pv = *reinterpret_cast<const float4*>(w.partialA + size_t(v) * HID + 4 * q);
                        av.x += pv.x; av.y += pv.y; av.z += pv.z; av.w += pv.w;
                    }
                    for (int t = (q0 + 3) >> 2; t <= (q0 + nq - 1) >> 2; ++t) {
                        const float4 pv = *reinterpret_cast<const float4*>(w.partial + size_t(t) * HID + 4 * q);
                        av.x += pv.x; av.y += pv.y; av.z += pv.z; av.w += pv.w;
                    }
                }
                if (d.mean) {
                    // 'mean': the row's edge count, masked edges included - the whole padded row on the reference's dense list
                    // (graph_type 3), the atom's degree in a radius graph; an atom without edges divides by 1 (egnn.py:315-319)
                    const float ms = 1.0f / float(d.graph_type == 3 ? d.N : max(w.deg[v], 1));
                    av.x *= ms; av.y *= ms; av.z *= ms; av.w *= ms;
                }
            }
        }
        *reinterpret_cast<float4*>(hL + r * LDT + 4 * q) = hv;
        if (post) *reinterpret_cast<float4*>(aL + r * LDT + 4 * q) = av;
        hmax = fmaxf(hmax, fmaxf(fmaxf(fabsf(hv.x), fabsf(hv.y)), fmaxf(fabsf(hv.z), fabsf(hv.w))));
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(av.x), fabsf(av.y)), fmaxf(fabsf(av.z), fabsf(av.w))));
    }
    if (PREC == 1) { wg_max(&mx[0], hmax, lane); wg_max(&mx[1], amax, lane); }
    __syncthreads();
    const int nt = wv;                                             // this wave's 32-feature tile
    float s_h = (PREC == 1) ? scale_for(__uint_as_float(mx[0])) : 1.0f;
    if (post) {
        const float* vecs = post + G_VEC;
        const float* sc = post + G_SCALE;
        const float b3 = vecs[4 * HID + 32 * nt + c];
        float s1 = 1.0f, s2 = 1.0f, inv = 1.0f;
        if (PREC == 1) {
            const float sw3a = sc[GS_SW_W3A + nt], sw3b = sc[GS_SW_W3B + nt];     // one weight scale per output tile (balanced packing)
            const float S = fminf(s_h * sw3a, scale_for(__uint_as_float(mx[1])) * sw3b);
            if (tid == 0 && (beyond_f16_range(__uint_as_float(mx[0])) || beyond_f16_range(__uint_as_float(mx[1])))) w.total[TW_RANGE] = 1;
            s1 = S * inv_pow2(sw3a); s2 = S * inv_pow2(sw3b); inv = inv_pow2(S);
        }
        floatx16 acc = splat16(PREC == 0 ? b3 : 0.0f);
        {
            const BFrag b3a = load_bfrag(post + G_W3A + nt * (UNIT / 4), lane);
            gemm_lds<PREC>(acc, hL, c, hh, b3a, s1);
        }
        {
            const BFrag b3b = load_bfrag(post + G_W3B + nt * (UNIT / 4), lane);
            gemm_lds<PREC>(acc, aL, c, hh, b3b, s2);
        }
        __syncthreads();                                           // all waves done reading aL
        float tmax = 0.0f;
        const float tfac = (PREC == 1) ? sc[GS_NT + nt] : 1.0f;       // the hidden layer of tile nt enters W4' times 2^n (W4' carries 2^-n)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float tv = silu_u(PREC == 0 ? acc[reg] : fmaf(acc[reg], inv, b3));
            aL[acc_row(reg, hh) * LDT + 32 * nt + c] = tv;
            tmax = fmaxf(tmax, fabsf(tv) * tfac);
        }
        if (PREC == 1) wg_max(&mx[2], tmax, lane);
        __syncthreads();
        const float b4 = vecs[5 * HID + 32 * nt + c];
        float s_t = 1.0f, inv4 = 1.0f;
        if (PREC == 1) {
            s_t = scale_for(__uint_as_float(mx[2])); inv4 = inv_pow2(s_t * sc[4]);
            if (tid == 0 && beyond_f16_range(__uint_as_float(mx[2]))) w.total[TW_RANGE] = 1;
        }
        floatx16 hn;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) hn[reg] = (PREC == 0) ? hL[acc_row(reg, hh) * LDT + 32 * nt + c] + b4 : 0.0f;
        {
            const BFrag b4f = load_bfrag(post + G_W4 + nt * (UNIT / 4), lane);
            gemm_lds<PREC>(hn, aL, c, hh, b4f, s_t, PREC == 1 ? sc + GS_NT : nullptr);
        }
        if (PREC == 1) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) hn[reg] = fmaf(hn[reg], inv4, hL[acc_row(reg, hh) * LDT + 32 * nt + c] + b4);
        }
        __syncthreads();                                           // all waves done reading hL (residual) and aL
        float nmax = 0.0f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int r = acc_row(reg, hh);
            const int v = row0 + r;
            const float val = (v < d.V && (w.flags[v] & F_REAL)) ? hn[reg] : 0.0f;   // h * node_mask
            hL[r * LDT + 32 * nt + c] = val;
            if (v < d.V) w.H[size_t(v) * HID + 32 * nt + c] = val;
            nmax = fmaxf(nmax, fabsf(val));
        }
        wg_max(&mx[3], nmax, lane);
        __syncthreads();
        if (PREC == 1) {
            s_h = scale_for(__uint_as_float(mx[3]));
            if (tid == 0 && beyond_f16_range(__uint_as_float(mx[3]))) w.total[TW_RANGE] = 1;
        }
        if (tid == 0) gmax_update(w.total, TW_HMAX, __uint_as_float(mx[3]));       // batch-wide max |h| (pk_edge_kernel<EQUIV>)
    }
    if (pre_units) {
        // P (feature tile nt of W1a') and Q (feature tile nt of W1b') for the next edge pass
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const BFrag bf = load_bfrag(pre_units + which * UNIT + nt * (UNIT / 4), lane);
            const float bias = which == 0 ? pre_bias[32 * nt + c] : 0.0f;
            const float inv = (PREC == 1) ? inv_pow2(s_h * pre_scale[4 * which + nt]) : 1.0f;
            const float efac = (PREC == 1) ? pre_ne[2 * nt + (c >> 4)] : 1.0f;      // 2^n of this lane's feature: the row maxima bound the SCALED operands
            floatx16 acc = splat16(PREC == 0 ? bias : 0.0f);
            gemm_lds<PREC>(acc, hL, c, hh, bf, s_h);
            float* dst = which == 0 ? w.P : w.Q;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = acc_row(reg, hh);
                const int v = row0 + r;
                const float val = (PREC == 0) ? acc[reg] : fmaf(acc[reg], inv, bias);
                if (v < d.V) dst[size_t(v) * HID + 32 * nt + c] = val;
                if (PREC == 1) {
                    // max over this wave's 32 features of row r (lanes of one half), then across the 4 waves in LDS
                    float m = fabsf(val) * efac;
                    m = fmaxf(m, dpp_mov<0xB1>(m));
                    m = fmaxf(m, dpp_mov<0x4E>(m));
                    m = fmaxf(m, dpp_mov<0x141>(m));
                    m = fmaxf(m, dpp_mov<0x140>(m));
                    if ((lane & 15) == 0) atomicMax(&rowmx[which][r], __float_as_uint(m));
                }
            }
        }
        if (PREC == 1) {
            __syncthreads();
            if (tid < 64) {
                const int v = row0 + (tid & 31);
                if (v < d.V) (tid < 32 ? w.pmax : w.qmax)[v] = __uint_as_float(rowmx[tid >> 5][tid & 31]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 6. edge kernel: persistent workgroups (4 waves, second-layer weights in LDS), waves take tiles round-robin.
//    A tile = 32 consecutive edge slots = 4 quads; the quads of one receiving atom inside the tile form a run.
//    EQUIV = false: one row per run = sum over the run's edges of u2[f]            (edge_mask = None: weight 1)
//    EQUIV = true : one triple per run = sum over the run's edges of cdiff * (w7'.u2)
//    The run that opens the tile goes to partial[t] / partialx[t], a run that starts inside it - the first quads of its
//    atom - to partialA[atom] / partialxA[atom].
// ---------------------------------------------------------------------------------------------------
// ATT (GCL): edge attention m_ij *= sigmoid(w_att . m_ij + b_att), `vec4` = w_att' (times 1/c: the messages carry c), b_att = sc[8].
// EQUIV with tanh: `head` = coords_range (0: no tanh), the head's output goes through coords_range * tanh(s).
// SIN: sin_embedding - the rank-2 term wr' r + wd' d0 of the first layer becomes a 24-term sum over the embedded distances
// sin / cos(sqrt(. + 1e-8) * f_k), f_k = 2 pi 4^k / 15 (egnn.py:281-292; radial: :159-161, d0: :221-222), `wg` = their weight
// columns [24][128].  The arithmetic up to the sine's argument follows the reference operation by operation (the top frequency
// turns one ulp of the distance into 1e-4 of phase), sinf / cosf are the accurate library versions.
__device__ __forceinline__ void sin_embed(float r, float d0, float (&e)[SIN_K]) {
    const float two_pi = 6.283185307179586f;
    const float dr = __fsqrt_rn(__fadd_rn(r, 1e-8f)), dd = __fsqrt_rn(__fadd_rn(d0, 1e-8f));
    float f4 = 1.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float fk = __fdiv_rn(__fmul_rn(two_pi, f4), 15.0f);
        const float a = __fmul_rn(dr, fk), b = __fmul_rn(dd, fk);
        e[k] = sinf(a); e[6 + k] = cosf(a);
        e[12 + k] = sinf(b); e[18 + k] = cosf(b);
        f4 *= 4.0f;
    }
}

template <bool EQUIV, int PREC, bool WEIGHTED, bool ATT, bool SIN>
// two waves per SIMD (two workgroups per CU): <= 256 VGPR + AGPR.  The attention variants keep the 64 messages of a step until
// the logit is known, the sin_embedding variants 24 embedded distances (292..410 registers): one wave per SIMD is what they get,
// and what they ask for
__global__ void __launch_bounds__(EDGE_THREADS, (ATT || SIN) ? 1 : 2)
pk_edge_kernel(PkDims d, PkWs w, const float* __restrict__ wimg, const float* __restrict__ vecs /* wr',wd',b2',(w7') */,
               const float* __restrict__ sc /* f16x3: static scales of this pass */, int sw_index,
               const float* __restrict__ vec4, float head, const float* __restrict__ wg, float wg_l1) {
    __shared__ __attribute__((aligned(16))) float W[UNIT];
    __shared__ __attribute__((aligned(16))) float vec[4 * HID];
    __shared__ __attribute__((aligned(16))) float WG[SIN ? SIN_K * HID : 4];
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, c = lane & 31, hh = lane >> 5;
    {
        const float4* src = reinterpret_cast<const float4*>(wimg);
        float4* dst = reinterpret_cast<float4*>(W);
        for (int e = tid; e < UNIT / 4; e += EDGE_THREADS) dst[e] = src[e];
        const int nv = EQUIV ? 4 : 3;
        for (int e = tid; e < nv * HID; e += EDGE_THREADS) vec[e] = vecs[e];
        if (ATT) for (int e = tid; e < HID; e += EDGE_THREADS) vec[3 * HID + e] = vec4[e];
        if (SIN) for (int e = tid; e < SIN_K * HID; e += EDGE_THREADS) WG[e] = wg[e];
    }
    __syncthreads();
    float bias[4], w7[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        bias[nt] = vec[2 * HID + 32 * nt + c];
        w7[nt] = (EQUIV || ATT) ? vec[3 * HID + 32 * nt + c] : 0.0f;
    }
    const float att_b = ATT ? sc[8] : 0.0f;                         // b_att (dl_model_create keeps it in the pass's scale block)
    const float4* wrp = reinterpret_cast<const float4*>(vec + 64 * hh);
    const float4* wdp = reinterpret_cast<const float4*>(vec + HID + 64 * hh);
    const float4* Wp = reinterpret_cast<const float4*>(W) + (64 * hh * 32 + c);
    const int nquads = w.total[0];
    const int ntiles = (nquads + 3) >> 2;
    const int qg = c >> 3;                                           // quad of this lane's slot
    // XCD-aware tile walk: workgroup b runs on XCD b % 8 (each with its own L2); an XCD takes one contiguous eighth of the work
    // list - the tiles of a few molecules - so that the sender rows it gathers (512 B per edge) stay in its L2 instead of
    // every L2 seeing every molecule's rows
    const int xcd = blockIdx.x & 7, nblk = (int(gridDim.x) + 7 - xcd) >> 3;
    const int gw = (blockIdx.x >> 3) * (EDGE_THREADS / 64) + wv, GW = nblk * (EDGE_THREADS / 64);
    // receiving atom and sender of this lane's slot in tile tt (slots past the last quad: padding of the last atom)
    auto tile_atoms = [&](int tt, int& ii, int& jj) {
        const int qd = 4 * tt + qg;
        const bool in = qd < nquads;
        ii = w.tile_row[in ? qd : nquads - 1];
        jj = w.col[size_t(tt) * 32 + c];                             // (-1 past the last quad: pk_scan_kernel)
    };

    // Software-pipelined tile loop.  Every global access of a tile is a dependent chain (tile -> atoms -> coordinates ->
    // rows of P / Q) and the vector-memory counter retires in order, so loading at the point of use costs ~20 serialized
    // L2 round trips per tile (measured: 15 K cycles per tile against 7.4 K on the LDS-resident path).  Instead:
    //   * the geometry (i, j, r, d0, bound) of the NEXT tile is fetched while this tile computes,
    //   * this tile's 32 Q rows are requested in one burst at the top (16 x 16 B per lane, consumed slab by slab),
    //   * the P rows (one per quad: up to four receiving atoms per tile) go through a per-wave LDS stage instead of 16 more
    //     loads per lane.
    __shared__ __attribute__((aligned(16))) float Pst[EDGE_THREADS / 64][4 * LDT];   // row stride LDT: the four rows fall into different banks
    float* pst = Pst[wv];
    // work list: every tile (GCL), the tiles with a linker receiver (coordinate head)
    // The coordinate head runs over the tiles that hold a linker receiver only: the reference multiplies every other atom's sum by
    // zero (egnn.py:113-116) - its result exactly, as long as the skipped sums are FINITE (an inf / NaN there is NaN * 0 = NaN in
    // the reference's coordinates).  The bound of the head's output proves it from the batch-wide maxima of |h| and |x|^2
    // (|trans| <= |w7' . u2| <= phi; at most N terms of mask weight <= 2 per sum); where it does not - an overflowing head,
    // non-finite features: the comparison is false for inf and NaN - every tile is computed and pk_xupdate_kernel multiplies by
    // the mask as the reference does.
    bool full = false;
    if constexpr (EQUIV) {
        const unsigned* tw = reinterpret_cast<const unsigned*>(w.total);
        const float hmax = __uint_as_float(tw[TW_HMAX]), x2 = __uint_as_float(tw[TW_X2]), x02 = __uint_as_float(tw[TW_X02]);
        const float geo = SIN ? wg_l1 : 4.0f * (x2 * sc[ES_WRW] + x02 * sc[ES_WDW]);
        const float u1b = (sc[8] + sc[9]) * hmax + sc[10] + geo;
        const float phi = sc[ES_W7L1] * fmaf(sc[ES_L1_W6], u1b, sc[ES_B6]);
        full = !(2.0f * float(d.N) * phi < 1e37f);
        if (blockIdx.x == 0 && tid == 0) w.total[TW_FULL] = full ? 1 : 0;
    }
    auto tile_of = [&](int k) { return (EQUIV && !full) ? w.eq_tiles[k] : k; };
    const int nall = (EQUIV && !full) ? w.total[1] : ntiles;
    const int wlo = int((long long)nall * xcd / 8), nwork = int((long long)nall * (xcd + 1) / 8);     // this XCD's part [wlo, nwork)
    int idx = wlo + gw;
    int t = idx < nwork ? tile_of(idx) : 0;
    int i = 0, jraw = -1;
    float r = 0.0f, d0 = 0.0f, dx = 0.0f, dy = 0.0f, dz = 0.0f, pqb = 0.0f;
    if (idx < nwork) {
        tile_atoms(t, i, jraw);
        const int j0 = jraw >= 0 ? jraw : i;
        const float4 xi = *reinterpret_cast<const float4*>(w.X + 4 * i);
        const float4 xj = *reinterpret_cast<const float4*>(w.X + 4 * j0);
        const float4 yi = *reinterpret_cast<const float4*>(w.X0 + 4 * i);
        const float4 yj = *reinterpret_cast<const float4*>(w.X0 + 4 * j0);
        dx = xi.x - xj.x; dy = xi.y - xj.y; dz = xi.z - xj.z;
        const float ex = yi.x - yj.x, ey = yi.y - yj.y, ez = yi.z - yj.z;
        if (SIN) {                                                   // torch.sum(coord_diff ** 2, 1): products, then two additions
            r = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            d0 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
        } else {
            r = dx * dx + dy * dy + dz * dz;
            d0 = ex * ex + ey * ey + ez * ez;
        }
        if (PREC != 0) pqb = w.pmax[i] + w.qmax[j0];
    }
    // rows of the CURRENT tile (requested one tile ahead): the P rows of the quads' receiving atoms and the 32 senders' Q rows
    float2 p2[4] = {make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f)};
    float4 qv[16];
    auto request_rows = [&](int ii, int jj, float2 (&pr)[4], float4 (&qr)[16]) {
#pragma unroll
        for (int g = 0; g < 4; ++g)                                  // lane 8g holds quad g's receiving atom (wave-uniform)
            pr[g] = *reinterpret_cast<const float2*>(w.P + size_t(__builtin_amdgcn_readlane(ii, 8 * g)) * HID + 2 * lane);
        const float* Qrow = w.Q + size_t(jj) * HID;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int k = (PREC == 0) ? 64 * hh + 4 * q : 16 * (q >> 1) + 4 * (q & 1) + 8 * hh;
            qr[q] = *reinterpret_cast<const float4*>(Qrow + k);
        }
    };
    if (idx < nwork) request_rows(i, jraw >= 0 ? jraw : i, p2, qv);
    for (; idx < nwork; idx += GW) {
        const bool valid = jraw >= 0;
        const unsigned vmask = unsigned(__ballot(valid));           // both halves hold the same 32 slots; padding ends every atom's list
        // the quads' receiving atoms (wave-uniform)
        const int ra0 = __builtin_amdgcn_readlane(i, 0), ra1 = __builtin_amdgcn_readlane(i, 8),
                  ra2 = __builtin_amdgcn_readlane(i, 16), ra3 = __builtin_amdgcn_readlane(i, 24);
        // ---- (1) the next tile's atoms (its rows and geometry are requested mid-body, once these have arrived)
        const bool more = idx + GW < nwork;
        const int tn = more ? tile_of(idx + GW) : t;
        int i_n = i, j_n = -1;
        if (more) tile_atoms(tn, i_n, j_n);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<float2*>(pst + g * LDT + 2 * lane) = p2[g];   // P rows -> per-wave LDS stage
        float2 p2n[4];
        float4 qn[16];
        floatx16 acc0, acc1, acc2, acc3;
        float xn[12];                                              // next tile: xi, xj, yi, yj coordinates
        float pq_n = 0.0f;
        auto request_next_geometry = [&]() {
            const int jn0 = j_n >= 0 ? j_n : i_n;
            const float4 a = *reinterpret_cast<const float4*>(w.X + 4 * i_n);
            const float4 b = *reinterpret_cast<const float4*>(w.X + 4 * jn0);
            const float4 cc = *reinterpret_cast<const float4*>(w.X0 + 4 * i_n);
            const float4 dd = *reinterpret_cast<const float4*>(w.X0 + 4 * jn0);
            xn[0] = a.x; xn[1] = a.y; xn[2] = a.z; xn[3] = b.x; xn[4] = b.y; xn[5] = b.z;
            xn[6] = cc.x; xn[7] = cc.y; xn[8] = cc.z; xn[9] = dd.x; xn[10] = dd.y; xn[11] = dd.z;
            if (PREC != 0) pq_n = w.pmax[i_n] + w.qmax[jn0];
            request_rows(i_n, jn0, p2n, qn);
        };
        float emb[SIN ? SIN_K : 1];
        if constexpr (SIN) sin_embed(r, d0, emb);
        if constexpr (PREC == 0) {
            float a[64];
            {
                const float4* Pp = reinterpret_cast<const float4*>(pst + qg * LDT + 64 * hh);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 P = Pp[q], Q = qv[q];
                    float4 y = make_float4(P.x + Q.x, P.y + Q.y, P.z + Q.z, P.w + Q.w);
                    if (SIN) {
#pragma unroll
                        for (int k = 0; k < SIN_K; ++k) {
                            const float4 g4 = *reinterpret_cast<const float4*>(WG + k * HID + 64 * hh + 4 * q);
                            y.x = fmaf(emb[k], g4.x, y.x); y.y = fmaf(emb[k], g4.y, y.y); y.z = fmaf(emb[k], g4.z, y.z); y.w = fmaf(emb[k], g4.w, y.w);
                        }
                    } else {
                        const float4 wr = wrp[q], wd = wdp[q];
                        y.x = fmaf(d0, wd.x, fmaf(r, wr.x, y.x)); y.y = fmaf(d0, wd.y, fmaf(r, wr.y, y.y));
                        y.z = fmaf(d0, wd.z, fmaf(r, wr.z, y.z)); y.w = fmaf(d0, wd.w, fmaf(r, wr.w, y.w));
                    }
                    a[4 * q + 0] = silu_u(y.x); a[4 * q + 1] = silu_u(y.y); a[4 * q + 2] = silu_u(y.z); a[4 * q + 3] = silu_u(y.w);
                }
            }
            request_next_geometry();
            __builtin_amdgcn_sched_barrier(0);
            acc0 = splat16(bias[0]); acc1 = splat16(bias[1]); acc2 = splat16(bias[2]); acc3 = splat16(bias[3]);
#pragma unroll
            for (int s_ = 0; s_ < 64; ++s_) {
                const float4 b = Wp[s_ * 32];
                acc0 = mfma32(a[s_], b.x, acc0);
                acc1 = mfma32(a[s_], b.y, acc1);
                acc2 = mfma32(a[s_], b.z, acc2);
                acc3 = mfma32(a[s_], b.w, acc3);
            }
        } else {
            // f16x3: |u| <= |y| <= max|P_i| + max|Q_j| + r*max|wr'| + d0*max|wd'|; the tile's largest bound sets the scale
            // (every term with the k-slab exponents of the balanced packing applied: pmax / qmax from pk_node_kernel, the weighted maxima
            // of wr', wd' and of the embedded-distance columns from dl_model_create)
            float bound = SIN ? pqb + wg_l1 : pqb + r * sc[EQUIV ? ES_WRW : GS_WRW] + d0 * sc[EQUIV ? ES_WDW : GS_WDW];
            bound = fmaxf(bound, dpp_mov<0xB1>(bound));
            bound = fmaxf(bound, dpp_mov<0x4E>(bound));
            bound = fmaxf(bound, dpp_mov<0x141>(bound));
            bound = fmaxf(bound, dpp_mov<0x140>(bound));
            {
                const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(bound), __float_as_uint(bound), false, false);
                bound = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
            }
            // wave-uniform (both halves hold the same pairs).  (The BITS go through readfirstlane: until round 5 the float itself did,
            // i.e. converted to int and back - a bound below 1 became 0 (scale 2^60: every activation saturated at the fp16 maximum)
            // and one above 2^31 became 2^31; found by the magnitude sweep of scripts/r5/debug_range.py, pinned by
            // tests/test_gpu_round5.py::test_hbm_resident_kernels_over_twenty_binades_of_magnitude)
            const float bound_u = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bound)));
            const float sa = scale_for(bound_u);
            if (beyond_f16_range(bound_u) && lane == 0) w.total[TW_RANGE] = 1;
            const float accs = sa * sc[sw_index], inv = inv_pow2(accs);
            // accumulators start from the inline constant 0; the bias joins in the (exact) rescaling fma
            acc0 = splat16(0.0f); acc1 = splat16(0.0f); acc2 = splat16(0.0f); acc3 = splat16(0.0f);
            const uint4* Wq = reinterpret_cast<const uint4*>(W) + lane;
            const float* Pp = pst + qg * LDT + 8 * hh;
            const float* wrb = vec + 8 * hh;
            const float* wdb = vec + HID + 8 * hh;
#pragma unroll
            for (int slab = 0; slab < 8; ++slab) {
                if (slab == 2) {                                   // pinned here: the results are needed after the last slab
                    request_next_geometry();
                    __builtin_amdgcn_sched_barrier(0);
                }
                float us[8];
                const float isa = inv_pow2(sa * sc[(EQUIV ? ES_NE : GS_NE) + slab]);     // the slab's features enter times sa * 2^n
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int k0 = 16 * slab + 4 * q;
                    const float4 P = *reinterpret_cast<const float4*>(Pp + k0);
                    const float4 Q = qv[2 * slab + q];
                    float4 y = make_float4(P.x + Q.x, P.y + Q.y, P.z + Q.z, P.w + Q.w);
                    if (SIN) {
#pragma unroll
                        for (int k = 0; k < SIN_K; ++k) {
                            const float4 g4 = *reinterpret_cast<const float4*>(WG + k * HID + 8 * hh + k0);
                            y.x = fmaf(emb[k], g4.x, y.x); y.y = fmaf(emb[k], g4.y, y.y); y.z = fmaf(emb[k], g4.z, y.z); y.w = fmaf(emb[k], g4.w, y.w);
                        }
                    } else {
                        const float4 wr = *reinterpret_cast<const float4*>(wrb + k0);
                        const float4 wd = *reinterpret_cast<const float4*>(wdb + k0);
                        y.x = fmaf(d0, wd.x, fmaf(r, wr.x, y.x)); y.y = fmaf(d0, wd.y, fmaf(r, wr.y, y.y));
                        y.z = fmaf(d0, wd.z, fmaf(r, wr.z, y.z)); y.w = fmaf(d0, wd.w, fmaf(r, wr.w, y.w));
                    }
                    us[4 * q + 0] = silu_scaled(y.x, isa); us[4 * q + 1] = silu_scaled(y.y, isa);
                    us[4 * q + 2] = silu_scaled(y.z, isa); us[4 * q + 3] = silu_scaled(y.w, isa);
                }
                uint4 ah, al;
                // PREC 2 (F16X2, GCL without attention): the activation enters as one fp16 rounded to nearest, no lo part
                constexpr bool TWO = (PREC == 2) && !EQUIV && !ATT;
                if constexpr (TWO) ah = round8(us);
                else split8(us, ah, al);
                uint4 bh[4], bl[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    bh[nt] = Wq[(slab * 4 + nt) * 64];
                    bl[nt] = Wq[((8 + slab) * 4 + nt) * 64];
                }
                if constexpr (!TWO) {
                    acc0 = mfma_h(al, bh[0], acc0); acc1 = mfma_h(al, bh[1], acc1);
                    acc2 = mfma_h(al, bh[2], acc2); acc3 = mfma_h(al, bh[3], acc3);
                }
                acc0 = mfma_h(ah, bl[0], acc0); acc1 = mfma_h(ah, bl[1], acc1);
                acc2 = mfma_h(ah, bl[2], acc2); acc3 = mfma_h(ah, bl[3], acc3);
                acc0 = mfma_h(ah, bh[0], acc0); acc1 = mfma_h(ah, bh[1], acc1);
                acc2 = mfma_h(ah, bh[2], acc2); acc3 = mfma_h(ah, bh[3], acc3);
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                acc0[reg] = fmaf(acc0[reg], inv, bias[0]); acc1[reg] = fmaf(acc1[reg], inv, bias[1]);
                acc2[reg] = fmaf(acc2[reg], inv, bias[2]); acc3[reg] = fmaf(acc3[reg], inv, bias[3]);
            }
        }
        if (!EQUIV) {
            // sums over the slots of each quad: accumulator registers 4g .. 4g+3 hold the rows of quad g
            float sq[4][4];                                         // [feature tile][quad], this lane half's rows
            {
                const unsigned vm = hh ? (vmask >> 4) : vmask;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int reg = 4 * g + k;
                        // WEIGHTED: the int8 mask value of the edge (0 on padding); else 1 on real edges, 0 on padding
                        float m = WEIGHTED ? w.wgt[size_t(t) * 32 + acc_row(reg, hh)] : float((vm >> (8 * g + k)) & 1u);
                        const float u0 = silu_u(acc0[reg]), u1 = silu_u(acc1[reg]), u2 = silu_u(acc2[reg]), u3 = silu_u(acc3[reg]);
                        if (ATT) {
                            // one logit per edge over its 128 features: this lane half's 32 lanes x 4 feature tiles
                            float lg = w7[0] * u0;
                            lg = fmaf(w7[1], u1, lg); lg = fmaf(w7[2], u2, lg); lg = fmaf(w7[3], u3, lg);
                            lg = half32_allsum(lg) + att_b;
                            m *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * lg));
                        }
                        s0 = fmaf(m, u0, s0);
                        s1 = fmaf(m, u1, s1);
                        s2 = fmaf(m, u2, s2);
                        s3 = fmaf(m, u3, s3);
                    }
                    sq[0][g] = s0; sq[1][g] = s1; sq[2][g] = s2; sq[3][g] = s3;
                }
            }
            // quads of one atom are merged from the back (wave-uniform branches); then the two lane halves of every run that
            // remains are added and the run is written once
            if (ra3 == ra2) { sq[0][2] += sq[0][3]; sq[1][2] += sq[1][3]; sq[2][2] += sq[2][3]; sq[3][2] += sq[3][3]; }
            if (ra2 == ra1) { sq[0][1] += sq[0][2]; sq[1][1] += sq[1][2]; sq[2][1] += sq[2][2]; sq[3][1] += sq[3][2]; }
            if (ra1 == ra0) { sq[0][0] += sq[0][1]; sq[1][0] += sq[1][1]; sq[2][0] += sq[2][1]; sq[3][0] += sq[3][1]; }
            auto write_run = [&](float* row, int g) {
                const float a0 = xor32_sum(sq[0][g]), a1 = xor32_sum(sq[1][g]), a2 = xor32_sum(sq[2][g]), a3 = xor32_sum(sq[3][g]);
                if (hh == 0) { row[c] = a0; row[32 + c] = a1; row[64 + c] = a2; row[96 + c] = a3; }
            };
            write_run(w.partial + size_t(t) * HID, 0);
            if (ra1 != ra0) write_run(w.partialA + size_t(ra1) * HID, 1);
            if (ra2 != ra1) write_run(w.partialA + size_t(ra2) * HID, 2);
            if (ra3 != ra2) write_run(w.partialA + size_t(ra3) * HID, 3);
        } else {
            const bool want_hi = ((c >> 2) & 1) != 0;
            const int my_reg = (c & 3) + 4 * (c >> 3);
            float s_own = 0.0f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float ts = w7[0] * silu_u(acc0[reg]);
                ts = fmaf(w7[1], silu_u(acc1[reg]), ts);
                ts = fmaf(w7[2], silu_u(acc2[reg]), ts);
                ts = fmaf(w7[3], silu_u(acc3[reg]), ts);
                ts = half32_allsum(ts);
                float lo, hi;
                both_halves(ts, lo, hi);
                s_own = (reg == my_reg) ? (want_hi ? hi : lo) : s_own;
            }
            if (head != 0.0f)                                        // tanh(s) * coords_range (egnn.py:104-105); uniform
                s_own = head * (1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * s_own)));
            const float den = sqrtf(r + 1e-8f) + d.norm_constant;   // coord2diff, egnn.py:299-300
            float f = (hh == 0 && valid) ? s_own : 0.0f;
            if (WEIGHTED) f *= w.wgt[size_t(t) * 32 + c];
            // sums over the 8 slots of each quad (lanes 8g .. 8g+7 of the first half), then the runs as above
            const float ax8 = quad8_allsum((dx / den) * f), ay8 = quad8_allsum((dy / den) * f), az8 = quad8_allsum((dz / den) * f);
            float qx[4], qy[4], qz[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                qx[g] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax8), 8 * g));
                qy[g] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay8), 8 * g));
                qz[g] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az8), 8 * g));
            }
            if (ra3 == ra2) { qx[2] += qx[3]; qy[2] += qy[3]; qz[2] += qz[3]; }
            if (ra2 == ra1) { qx[1] += qx[2]; qy[1] += qy[2]; qz[1] += qz[2]; }
            if (ra1 == ra0) { qx[0] += qx[1]; qy[0] += qy[1]; qz[0] += qz[1]; }
            if (lane == 0) {
                *reinterpret_cast<float4*>(w.partialx + size_t(t) * 4) = make_float4(qx[0], qy[0], qz[0], 0.0f);
                if (ra1 != ra0) *reinterpret_cast<float4*>(w.partialxA + size_t(ra1) * 4) = make_float4(qx[1], qy[1], qz[1], 0.0f);
                if (ra2 != ra1) *reinterpret_cast<float4*>(w.partialxA + size_t(ra2) * 4) = make_float4(qx[2], qy[2], qz[2], 0.0f);
                if (ra3 != ra2) *reinterpret_cast<float4*>(w.partialxA + size_t(ra3) * 4) = make_float4(qx[3], qy[3], qz[3], 0.0f);
            }
        }
        // ---- the prefetched geometry becomes the current one
        t = tn;
        i = i_n; jraw = j_n;
        dx = xn[0] - xn[3]; dy = xn[1] - xn[4]; dz = xn[2] - xn[5];
        {
            const float ex = xn[6] - xn[9], ey = xn[7] - xn[10], ez = xn[8] - xn[11];
            if (SIN) {
                r = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                d0 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
            } else {
                r = dx * dx + dy * dy + dz * dz;
                d0 = ex * ex + ey * ey + ez * ez;
            }
        }
        pqb = pq_n;
#pragma unroll
        for (int g = 0; g < 4; ++g) p2[g] = p2n[g];
#pragma unroll
        for (int q = 0; q < 16; ++q) qv[q] = qn[q];
    }
}

// 7. coordinate update x += (sum of the atom's tile partials) * linker_mask, masked (egnn.py:110-124)
__global__ void pk_xupdate_kernel(PkDims d, PkWs w, const float* __restrict__ linker_mask) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= d.V) return;
    if (!(w.flags[v] & F_MOVES) && w.total[TW_FULL] == 0) return;  // its tiles were not even computed (pk_eqtiles_kernel; pk_edge_kernel<EQUIV>)
    const int q0 = w.tile_off[v], nq = w.ntile[v];
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (nq > 0) {
        if (q0 & 3) {
            const float4 p = *reinterpret_cast<const float4*>(w.partialxA + size_t(v) * 4);
            ax += p.x; ay += p.y; az += p.z;
        }
        for (int t = (q0 + 3) >> 2; t <= (q0 + nq - 1) >> 2; ++t) {
            const float4 p = *reinterpret_cast<const float4*>(w.partialx + size_t(t) * 4);
            ax += p.x; ay += p.y; az += p.z;
        }
    }
    // w7' carries 1/normalization_factor unless tanh or the mean need the raw head output (dl_model_create)
    const float xs = d.mean ? 1.0f / float(d.graph_type == 3 ? d.N : max(w.deg[v], 1)) : (d.tanh ? d.inv_norm : 1.0f);
    ax *= xs; ay *= xs; az *= xs;
    const float lm = linker_mask ? linker_mask[v] : 1.0f;
    const float nm = (w.flags[v] & F_REAL) ? 1.0f : 0.0f;
    float4 x = *reinterpret_cast<const float4*>(w.X + 4 * v);
    x.x = (x.x + ax * lm) * nm; x.y = (x.y + ay * lm) * nm; x.z = (x.z + az * lm) * nm;
    *reinterpret_cast<float4*>(w.X + 4 * v) = x;
    gmax_update(w.total, TW_X2, x.x * x.x + x.y * x.y + x.z * x.z);
}

// 8. output head: h_final = (Wo h + bo)[:nf] * node_mask, vel = (x - x0) * node_mask, NaN flags per molecule
__global__ void pk_out_kernel(PkDims d, PkWs w, const float* __restrict__ wp, float* __restrict__ out,
                              int* __restrict__ nan_flags) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = gid / d.D, k = gid - v * d.D;
    if (v >= d.V) return;
    const float nm = (w.flags[v] & F_REAL) ? 1.0f : 0.0f;
    float val;
    int bit;
    if (k < 3) {
        val = (w.X[4 * v + k] - w.X0[4 * v + k]) * nm;
        bit = 1;
    } else {
        const int o = k - 3;
        const float* hp = w.H + size_t(v) * HID;
        const float* wo = wp + OFF_OUT_W + o * HID;
        float acc = wp[OFF_OUT_B + o];
        for (int q = 0; q < HID; ++q) acc = fmaf(hp[q], wo[q], acc);
        val = acc * nm;
        bit = 2;
    }
    out[size_t(v) * d.D + k] = val;
    if (val != val) atomicOr(&nan_flags[v / d.N], bit);
    if (k == 0 && v % d.N == 0 && w.total[TW_RANGE] != 0) atomicOr(&nan_flags[v / d.N], NAN_RANGE | 3);
}

thread_local int g_sparse_last_hip = 0;
inline bool ok(hipError_t e) {
    if (e != hipSuccess) { g_sparse_last_hip = int(e); return false; }
    return true;
}

}  // namespace

extern "C" {

size_t dl_pocket_workspace_bytes(int32_t B, int32_t N) {
    if (B <= 0 || N <= 0) return 0;
    return carve(nullptr, B, N).bytes;
}

}  // extern "C"

namespace {
int32_t run_sparse(const dl_model* m, int32_t B, int32_t N, int32_t graph_type, const float* xh, const float* t,
                   int32_t t_is_scalar, const int8_t* node_mask, const float* linker_mask, const int8_t* emask,
                   const float* context, float* out, int32_t* nan_flags, void* workspace, size_t workspace_bytes,
                   void* stream_) {
    if (B == 0) return DL_OK;
    if (workspace_bytes < carve(nullptr, B, N).bytes) return DL_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const ModelDims md = dims_of(m);
    PkDims d;
    d.B = B; d.N = N; d.V = B * N; d.nf = md.nf; d.ctx = md.ctx; d.fin = md.fin; d.D = 3 + md.nf; d.ct = md.ct;
    d.graph_type = graph_type; d.norm_constant = md.norm_constant; d.emask = emask;
    const bool weighted = graph_type == 3;
    const PkWs w = carve(workspace, B, N);
    const float* wp = m->d_pack;
    const int V = d.V;

    if (!ok(hipMemsetAsync(nan_flags, 0, size_t(B) * 4, st))) return DL_ERR_HIP;
    if (!ok(hipMemsetAsync(w.total, 0, 256, st))) return DL_ERR_HIP;           // counters and batch-wide maxima (TW_*)
    hipLaunchKernelGGL(pk_init_kernel, dim3((V * HID + 255) / 256), dim3(256), 0, st, d, w, wp, xh, t,
                       t_is_scalar ? 0 : 1, node_mask, linker_mask, context);
    hipLaunchKernelGGL(pk_edges_kernel<false>, dim3((V + 3) / 4), dim3(256), 0, st, d, w);
    hipLaunchKernelGGL(pk_scan_kernel, dim3(1), dim3(1024), 0, st, V, w.ntile, w.tile_off, w.total, w.col, w.wgt);
    hipLaunchKernelGGL(pk_edges_kernel<true>, dim3((V + 3) / 4), dim3(256), 0, st, d, w);
    {
        const int tmax = int((size_t(V) * (N / 8 + 1) + 4) / 4 + 1);          // upper bound of the tile count (carve)
        hipLaunchKernelGGL(pk_eqtiles_kernel, dim3((tmax + 255) / 256), dim3(256), 0, st, w);
    }

    const int row_tiles = (V + 31) / 32;
    const int edge_grid = 512;                                     // 2 workgroups per CU (64 KB LDS each)
    const bool f16 = m->cfg.precision != DL_PRECISION_FP32, two = m->cfg.precision == DL_PRECISION_F16X2;
    // `pre`: the pass whose projections the kernel ends with (a GCL, or the equivariant update: `pre_equiv`)
    auto node = [&](const float* post, const float* pre, bool pre_equiv) {
        const float* pre_units = pre + (pre_equiv ? E_W5A : G_W1A);
        const float* pre_bias = pre + (pre_equiv ? E_VEC : G_VEC);
        const float* pre_scale = pre + (pre_equiv ? E_SCALE + ES_SW_W5A : G_SCALE + GS_SW_W1A);      // [2][4] per-tile weight scales
        const float* pre_ne = pre + (pre_equiv ? E_SCALE + ES_NE : G_SCALE + GS_NE);                  // [8] slab factors of its hidden layer
        if (f16) hipLaunchKernelGGL(pk_node_kernel<1>, dim3(row_tiles), dim3(NODE_THREADS), 0, st, d, w, post, pre_units, pre_bias, pre_scale, pre_ne);
        else hipLaunchKernelGGL(pk_node_kernel<0>, dim3(row_tiles), dim3(NODE_THREADS), 0, st, d, w, post, pre_units, pre_bias, pre_scale, pre_ne);
    };
    d.attention = md.attention; d.tanh = md.tanh; d.mean = md.mean; d.coords_range = md.coords_range; d.inv_norm = md.inv_norm;
    d.sin = md.sin;
    // edge pass of a GCL (equiv = false; `vec4` = w_att' with attention) or of the coordinate head (equiv = true;
    // `head` = coords_range with tanh)
    auto edge = [&](bool equiv, const float* wimg, const float* vecs, const float* sc, int sw, const float* vec4, float head,
                    const float* wg, float wg_l1) {
#define DL_EDGE5(E, P, W, A, S) hipLaunchKernelGGL((pk_edge_kernel<E, P, W, A, S>), dim3(edge_grid), dim3(EDGE_THREADS), 0, st, d, w, wimg, vecs, sc, sw, vec4, head, wg, wg_l1)
#define DL_EDGE(E, P, W, A) do { if (md.sin) DL_EDGE5(E, P, W, A, true); else DL_EDGE5(E, P, W, A, false); } while (0)
        if (!equiv) {
            if (md.attention) {
                if (f16 && weighted) DL_EDGE(false, 1, true, true);
                else if (f16) DL_EDGE(false, 1, false, true);
                else if (weighted) DL_EDGE(false, 0, true, true);
                else DL_EDGE(false, 0, false, true);
            } else {
                if (two && weighted) DL_EDGE(false, 2, true, false);
                else if (two) DL_EDGE(false, 2, false, false);
                else if (f16 && weighted) DL_EDGE(false, 1, true, false);
                else if (f16) DL_EDGE(false, 1, false, false);
                else if (weighted) DL_EDGE(false, 0, true, false);
                else DL_EDGE(false, 0, false, false);
            }
        } else {
            if (f16 && weighted) DL_EDGE(true, 1, true, false);
            else if (f16) DL_EDGE(true, 1, false, false);
            else if (weighted) DL_EDGE(true, 0, true, false);
            else DL_EDGE(true, 0, false, false);
        }
#undef DL_EDGE
#undef DL_EDGE5
    };
    // sin_embedding: the largest row L1 norm of the embedded-distance columns bounds their term (host copy of the pass's scale slot)
    auto wg_bound = [&](int blk, int which) -> float {
        if (!md.sin) return 0.0f;
        return m->sin_l1[size_t(blk) * (MAX_SUBLAYERS + 1) + which];
    };
    for (int blk = 0; blk < md.n_layers; ++blk) {
        const float* base = wp + OFF_BLOCKS + size_t(blk) * block_size(md.sub);
        const float* eq = base + md.sub * GCL_SIZE;
        // projections for gcl_0 (the previous block's node kernel already produced them, except for block 0)
        if (blk == 0) node(nullptr, base, false);
        for (int gi = 0; gi < md.sub; ++gi) {                      // inv_sublayers GCLs (2 in every released configuration) ...
            const float* g = base + gi * GCL_SIZE;
            edge(false, g + G_W2, g + G_VEC + HID, g + G_SCALE, 5, g + G_VEC + 6 * HID, 0.0f, g + G_WG, wg_bound(blk, gi));
            if (gi + 1 < md.sub) node(g, g + GCL_SIZE, false);
            else node(g, eq, true);
        }
        // ... then the equivariant update
        edge(true, eq + E_W6, eq + E_VEC + HID, eq + E_SCALE, 2, nullptr, md.tanh ? md.coords_range : 0.0f, eq + E_WG, wg_bound(blk, md.sub));
        hipLaunchKernelGGL(pk_xupdate_kernel, dim3((V + 255) / 256), dim3(256), 0, st, d, w, linker_mask);
        if (blk + 1 < md.n_layers) {
            const float* n0 = base + block_size(md.sub);           // next block's gcl_0 projections (h unchanged)
            node(nullptr, n0, false);
        }
    }
    hipLaunchKernelGGL(pk_out_kernel, dim3((V * d.D + 255) / 256), dim3(256), 0, st, d, w, wp, out, nan_flags);
    return ok(hipGetLastError()) ? DL_OK : DL_ERR_HIP;
}
}  // namespace

extern "C" {

int32_t dl_egnn_forward_pocket(const dl_model* m, int32_t B, int32_t N, int32_t graph_type, const float* xh,
                               const float* t, int32_t t_is_scalar, const int8_t* node_mask,
                               const float* linker_mask, const float* context, float* out, int32_t* nan_flags,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || !xh || !t || !node_mask || !linker_mask || !context || !out || !nan_flags || !workspace)
        return DL_ERR_BAD_ARG;
    if (B < 0 || N < 1 || graph_type < 0 || graph_type > 2) return DL_ERR_BAD_ARG;
    if (m->cfg.context_node_nf < 2) return DL_ERR_BAD_ARG;                      // needs the fragment/pocket channels
    return run_sparse(m, B, N, graph_type, xh, t, t_is_scalar, node_mask, linker_mask, nullptr, context, out, nan_flags,
                      workspace, workspace_bytes, stream);
}

int32_t dl_egnn_forward_fc_large(const dl_model* m, int32_t B, int32_t N, const float* xh, const float* t,
                                 int32_t t_is_scalar, const int8_t* node_mask, const float* linker_mask,
                                 const int8_t* edge_mask, const float* context, float* out, int32_t* nan_flags,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || !xh || !t || !node_mask || !edge_mask || !out || !nan_flags || !workspace) return DL_ERR_BAD_ARG;
    if (m->cfg.context_node_nf > 0 && !context) return DL_ERR_BAD_ARG;
    if (B < 0 || N < 1) return DL_ERR_BAD_ARG;
    return run_sparse(m, B, N, 3, xh, t, t_is_scalar, node_mask, linker_mask, edge_mask, context, out, nan_flags, workspace,
                      workspace_bytes, stream);
}

}  // extern "C"
