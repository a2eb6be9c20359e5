// pack_layout.h — definitions shared by the kernels' translation units: the packed-weight buffer layout
// (written by dl_model_create in egnn_fc.hip), the model handle, and a few device helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/difflinker_hip.h"

struct dl_model {
    dl_config cfg;
    float* d_pack;
    size_t n_floats;
    float sin_l1[64 * 5];           // sin_embedding: per block, row L1 bound of the embedded-distance columns of gcl_0 .. gcl_{k-1}, gcl_equiv (k = inv_sublayers <= 4)
};

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int HID = 128;            // hidden_nf
constexpr int UNIT = HID * HID;     // one packed 128x128 matrix
constexpr int FINP = 16;            // embedding input width, padded
constexpr int DMAX = 16;            // row stride of the per-atom state z = [x(3), h(nf)], 3+nf <= 16
constexpr int CTXMAX = 4;

// ---- packed weight buffer (floats); mirrored by pack_model() below -------------------------------
constexpr int OFF_EMB_W = 0;                          // [128][FINP]
constexpr int OFF_EMB_B = OFF_EMB_W + HID * FINP;     // [128]
constexpr int OFF_OUT_W = OFF_EMB_B + HID;            // [16][128]
constexpr int OFF_OUT_B = OFF_OUT_W + 16 * HID;       // [16]
constexpr int OFF_BLOCKS = OFF_OUT_B + 16;
// GCL: units W1a', W1b', W3a', W3b', W4' (node-fragment order), W2' (LDS image of the sparse kernels: pairs are the
// MFMA rows), W2' again as the LDS image of the LDS-resident kernels (features are the MFMA rows, see egnn_fc.hip), vectors
constexpr int G_W1A = 0, G_W1B = UNIT, G_W3A = 2 * UNIT, G_W3B = 3 * UNIT, G_W4 = 4 * UNIT, G_W2 = 5 * UNIT, G_W2T = 6 * UNIT;
constexpr int G_VEC = 7 * UNIT;                       // b1', wr', wd', b2', b3', b4, w_att'   (7 x 128)
constexpr int G_SCALE = G_VEC + 7 * HID;             // scalars of the pass (floats).  [4] sw(W4'), [5] sw(W2'), [6], [7] max |wr'|, |wd'| (the geometric
                                                     // term's own scales), [8] b_att; f16 modes: [12..19] the bounds behind the a-priori scales, [20] sin_embedding;
                                                     // [24..] the per-tile / per-slab entries of the balanced packing (GS_*, below)
// ---- f16 modes, round 5: "balanced" packing.  The hidden features of every two-layer MLP are RENUMBERED by magnitude class
// (a static proxy: the row L1 norm of the first layer + |bias|), so that 32 consecutive output rows of a first-layer matrix
// share one power-of-two weight scale (GS_SW_*: one per 32-feature tile instead of one per matrix) and 16 / 32 consecutive
// hidden features share one power-of-two exponent (GS_NE: the k-slabs of the edge model's second layer; GS_NT: the tiles of the
// node MLP's hidden layer): the post-activation operand enters the second layer as a_k 2^n, the second layer's columns are
// packed times 2^-n.  With one scale per matrix and per activation tensor a TRAINED network - rows of very different norms,
// large biases - pushed the small features tens of binades below the bound that sets the scale, into fp16's subnormal
// range, where the split's lo part has no bits left (measured: node features 7e-5 instead of 3e-7, tests/test_gpu_round5.py;
// scripts/numerics/emulate_bounds.py).  The bounds [12..19] are taken with the exponents applied.
constexpr int GS_SW_W1A = 24, GS_SW_W1B = 28, GS_SW_W3A = 32, GS_SW_W3B = 36;      // 4 each: weight scale of output tile nt
constexpr int GS_NE = 40;                            // 8: 2^n of k-slab s of the edge model's hidden layer
constexpr int GS_NT = 48;                            // 4: 2^n of tile nt of the node MLP's hidden layer
constexpr int GS_WRW = 52, GS_WDW = 53;              // max over features of 2^n |wr'|, 2^n |wd'| (the bound of the scaled activations)
constexpr int G_SCALE_SIZE = 64;
constexpr int G_WG = G_SCALE + G_SCALE_SIZE;         // sin_embedding: the 24 embedded-distance columns of edge_mlp.0, [k][128], times c
constexpr int SIN_K = 24;                            // 6 frequencies x (sin, cos) x (radial, d0)   (egnn.py:281-292, :159-161, :221-222)
// ---- round 6, f16 modes: the same five per-atom matrices once more, as the A-OPERAND STREAM of the atom-stationary per-atom
// phases of egnn_fc.hip (stream_phase): 16-feature output tiles x 32-wide k-slabs of v_mfma_f32_16x16x32_f16, in the order they
// are consumed - a unit = two 32 KB chunks of four output tiles, chunk[ot][slab][hi | lo][lane][8 fp16], lane (r, kg) holding
// part(sw * W'[16 ot + r][kslot(slab, kg, e)]) with the k-slots in the order the previous GEMM's accumulators leave them
// (stream_kslot).  G_ST_POST, after this GCL's pair loop: W3a' and W3b' interleaved chunk by chunk (tiles 0-3 of one, of the
// other, tiles 4-7 of one, of the other), then W4'; G_ST_PRE, what opens this GCL: W1a', W1b'.
// Same weights, scales and renumbering as the units above (one sc[] block serves both); the exact-fp32 mode leaves them zero.
constexpr int G_ST_POST = 7 * UNIT + 7 * HID + G_SCALE_SIZE + SIN_K * HID;
constexpr int G_ST_PRE = G_ST_POST + 3 * UNIT;
constexpr int GCL_SIZE = G_ST_PRE + 2 * UNIT;
// equivariant update: units W5a', W5b', W6' (both LDS images), vectors
constexpr int E_W5A = 0, E_W5B = UNIT, E_W6 = 2 * UNIT, E_W6T = 3 * UNIT;
constexpr int E_VEC = 4 * UNIT;                       // b5', wr', wd', b6', w7'       (5 x 128)
constexpr int E_SCALE = E_VEC + 5 * HID;             // [2] sw(W6'), [6], [7] max |wr'|, |wd'|, [8..10] bounds: L1(W5a'), L1(W5b'), max |b5'| (exponents applied), [11] sin
constexpr int ES_SW_W5A = 16, ES_SW_W5B = 20;        // 4 each: weight scale of output tile nt
constexpr int ES_NE = 24;                            // 8: 2^n of k-slab s of the coordinate model's hidden layer
constexpr int ES_WRW = 32, ES_WDW = 33;
// every arithmetic mode: what bounds the coordinate head's output, |w7' . u2| <= ES_W7L1 * (ES_L1_W6 * max |u1| + ES_B6) - the kernels
// skip the coordinate sums the linker mask multiplies by zero only while this proves them finite (egnn_fc.hip: equiv_pass2)
constexpr int ES_L1_W6 = 34, ES_B6 = 35, ES_W7L1 = 36;
constexpr int E_SCALE_SIZE = 48;
constexpr int E_WG = E_SCALE + E_SCALE_SIZE;         // sin_embedding: the same for coord_mlp.0
constexpr int E_ST_PRE = 4 * UNIT + 5 * HID + E_SCALE_SIZE + SIN_K * HID;      // stream units W5a', W5b' (see G_ST_*)
constexpr int EQ_SIZE = E_ST_PRE + 2 * UNIT;
// k-slot (slab, kg, e) of the stream's B operand = feature 32 slab + 4 kg + (e & 3) + 16 (e >> 2): lane (atom, kg) of a 16x16
// accumulator tile ot holds features 16 ot + 4 kg + 0..3, so the tiles 2 slab and 2 slab + 1 ARE k-slab `slab` of the next GEMM
__host__ __device__ inline int stream_kslot(int slab, int kg, int e) { return 32 * slab + 4 * kg + (e & 3) + 16 * (e >> 2); }
constexpr int ST_CHUNK = UNIT / 2;                    // floats of one 32 KB chunk
constexpr int MAX_SUBLAYERS = 4;                     // inv_sublayers: GCLs per block (reference default and every released config: 2)
__host__ __device__ inline size_t block_size(int sublayers) { return size_t(sublayers) * GCL_SIZE + EQ_SIZE; }

struct ModelDims {
    int nf, ctx, fin, n_layers;
    int sub, ct;                        // inv_sublayers (GCLs per block); condition_time (0 / 1: the time feature joins the node inputs)
    float norm_constant;
    int attention, tanh, mean, sin;     // optional hyper-parameters (sin_embedding: the HBM-resident kernels only)
    float coords_range, inv_norm;
};

inline ModelDims dims_of(const dl_model* m) {
    ModelDims md;
    md.nf = m->cfg.in_node_nf;
    md.ctx = m->cfg.context_node_nf;
    md.ct = m->cfg.condition_time ? 1 : 0;
    md.fin = md.nf + md.ct + md.ctx;
    md.n_layers = m->cfg.n_layers;
    md.sub = m->cfg.inv_sublayers;
    md.norm_constant = m->cfg.norm_constant;
    md.attention = m->cfg.attention; md.tanh = m->cfg.tanh; md.mean = m->cfg.aggregation_mean; md.sin = m->cfg.sin_embedding;
    md.coords_range = m->cfg.coords_range; md.inv_norm = 1.0f / m->cfg.normalization_factor;
    return md;
}

// u = y * sigmoid(-y / log2e)  ==  -log2(e) * SiLU(pre)  for  y = -log2(e) * pre
__device__ __forceinline__ float silu_u(float y) {
    return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y));
}

// the same times a power of two s, with 1/s (exact) inside the reciprocal: y * (s / (1 + 2^y)), one multiply less
__device__ __forceinline__ float silu_scaled(float y, float inv_s) {
    return y * __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_exp2f(y), inv_s, inv_s));
}

__device__ __forceinline__ floatx16 splat16(float v) {
    floatx16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = v;
    return r;
}

__device__ __forceinline__ floatx16 mfma32(float a, float b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row of the 32x32 accumulator tile held in register `reg` by a lane of half `hh`
__device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// ---- cross-lane sums without the LDS crossbar (ds_bpermute costs ~24 cycles per wave instruction; these are VALU ops)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// every lane gets the sum over its 16-lane row: quad_perm xor1, xor2, row_half_mirror, row_mirror (fused v_add_f32_dpp)
__device__ __forceinline__ float row16_allsum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
// every lane gets the sum over its group of 8 consecutive lanes (quad_perm xor1, xor2, row_half_mirror)
__device__ __forceinline__ float quad8_allsum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    return v;
}
// every lane gets the sum over its 32-lane half (gfx950 v_permlane16_swap exchanges the two rows of a half)
__device__ __forceinline__ float half32_allsum(float v) {
    v = row16_allsum(v);
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// value of lane (l & 31) and of lane 32 + (l & 31), in every lane (v_permlane32_swap)
__device__ __forceinline__ void both_halves(float v, float& lo, float& hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    float lo, hi;
    both_halves(v, lo, hi);
    return lo + hi;
}

// ---- f16x3 split arithmetic (see egnn_fc.hip) ----
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// 8 consecutive-k fp32 values (already scaled) -> MFMA fragments (8 fp16 = 4 VGPRs) of the hi and lo parts.
// v_cvt_pkrtz_f16_f32 truncates, so lo = x - hi is exact and has the sign of x.
__device__ __forceinline__ void split8(const float (&u)[8], uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const auto hp = __builtin_amdgcn_cvt_pkrtz(u[2 * q], u[2 * q + 1]);
        const float h0 = float(hp[0]), h1 = float(hp[1]);
        const auto lp = __builtin_amdgcn_cvt_pkrtz(u[2 * q] - h0, u[2 * q + 1] - h1);
        h[q] = __builtin_bit_cast(unsigned, hp);
        l[q] = __builtin_bit_cast(unsigned, lp);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// 8 scaled fp32 values -> one fragment of fp16 rounded to nearest (v_cvt_pk_f16_f32): the two-term scheme of DL_PRECISION_F16X2
__device__ __forceinline__ uint4 round8(const float (&u)[8]) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    typedef float float2v __attribute__((ext_vector_type(2)));
    unsigned h[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float2v p = {u[2 * q], u[2 * q + 1]};
        h[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, half2v));
    }
    return make_uint4(h[0], h[1], h[2], h[3]);
}

__device__ __forceinline__ floatx16 mfma_h(const uint4& a, const uint4& b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

// largest power of two s with bound * s < 2^15 (fp16 max 65504); 1 for inf/NaN bounds, clamped to 2^+-60
__device__ __forceinline__ float scale_for(float bound) {
    const int ex = int((__float_as_uint(bound) >> 23) & 0xffu);     // bound < 2^(ex - 126)
    int f = 268 - ex;                                               // biased exponent of 2^(15 - (ex - 126))
    f = min(max(f, 67), 187);
    if (ex == 255) f = 127;
    return __uint_as_float(unsigned(f) << 23);
}

// A bound of 2^75 (3.8e22) or more - +inf included - is beyond the clamp: a value behind it may exceed the fp16 range once scaled,
// and v_cvt_pkrtz would SATURATE it at 65504 - a finite wrong number, not an inf.  Every kernel that derives a scale from a
// run-time bound reports that instead (nan_flags: NAN_RANGE | x | h - the molecule's denoiser output is void: utils.FoundNaNException
// with f16_range_idx set; precision='fp32' has no such limit).  A NaN bound is not reported here: it comes from NaN data, which
// travels through the arithmetic itself.
constexpr int NAN_RANGE = 16;
__device__ __forceinline__ bool beyond_f16_range(float bound) { return bound >= 0x1p75f; }

// exact reciprocal of a power of two
__device__ __forceinline__ float inv_pow2(float s) { return __uint_as_float(0x7f000000u - __float_as_uint(s)); }

// B fragments of one packed unit slice (one 32-feature tile, K = 128): 16 x dwordx4 per lane from L2.
struct BFrag {
    float4 q[16];
};
__device__ __forceinline__ BFrag load_bfrag(const float* __restrict__ unit_nt, int lane) {
    BFrag b;
    const float4* bp = reinterpret_cast<const float4*>(unit_nt) + lane;
#pragma unroll
    for (int sg = 0; sg < 16; ++sg) b.q[sg] = bp[sg * 64];
    return b;
}

// ---- counter-based noise: Philox4x32-10 keyed by the caller's seed; the counter is
// (global molecule index, atom position inside the molecule, draw index, component / 4), so a sample does not depend on
// the batch split across GPUs, on the batch size or on the padded width.  Four 32-bit outputs -> four standard normals
// by Box-Muller on 24-bit uniforms ((r >> 8) + 0.5) * 2^-24, i.e. |z| <= sqrt(2 ln 2^25) ~ 5.9.  Restated for the CPU in oracle/philox_oracle.py.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = unsigned(p1 >> 32) ^ c1 ^ k0, n1 = unsigned(p1), n2 = unsigned(p0 >> 32) ^ c3 ^ k1, n3 = unsigned(p0);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// standard normal number `comp` of draw `draw` for atom `atom` of global molecule `mol`
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned mol, unsigned atom, unsigned draw, unsigned comp) {
    unsigned r[4];
    philox4x32_10(mol, atom, draw, comp >> 2, unsigned(seed), unsigned(seed >> 32), r);
    const unsigned pair = comp & 2u;                                   // components 0,1 use r[0],r[1]; 2,3 use r[2],r[3]
    const float u1 = (float(r[pair] >> 8) + 0.5f) * 5.9604644775390625e-08f;       // 24 bits -> (0,1)
    const float u2 = (float(r[pair + 1] >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float rad = sqrtf(-2.0f * logf(u1));
    return rad * ((comp & 1u) ? sinpif(2.0f * u2) : cospif(2.0f * u2));
}

}  // namespace
