// egnn_fc.hip — MI355X (gfx950, CDNA4) kernels for DiffLinker's EGNN denoiser on fully-connected
// molecular graphs, and the fused ancestral-sampling chain built on it.
//
// Reference behaviour (file:line into igashov/DiffLinker): Dynamics.forward src/egnn.py:374-447,
// EGNN.forward :218-238, EquivariantBlock :157-178, GCL :45-80, EquivariantUpdate :101-125,
// coord2diff :295-301, unsorted_segment_sum :304-320, EDM.sample_chain src/edm.py:126-242.
//
// Design (sections 4 and 5):
//   * one 512-thread workgroup (8 wave64) per molecule - or a TEAM of 2 / 4 / 8 workgroups; the molecule's state lives in LDS for
//     the whole forward and the whole T-step chain (sample_chain_fc_kernel: one launch per chain, the sampler algebra and the
//     Philox draws between the forwards in the kernel): first-layer projections P, Q [n,128], coordinates, masks, the chain
//     state z; every LDS address is a compile-time constant (L_* below).
//   * the O(n^2) edge pass never materialises an edge tensor (pair_phase): receiver-stationary, both edge layers TRANSPOSED
//     (features x pairs) so that the accumulator layout of the first layer is the B-operand layout of the second; messages are
//     summed in the receiver's accumulators, the 256 slot partials added in a fixed order afterwards (deterministic, no atomics).
//   * arithmetic (template PREC): 0 = exact fp32 (v_mfma_f32_32x32x2_f32), 1 = f16x3 - every operand scaled by a power of two,
//     split into fp16 hi + lo, three v_mfma_f32_32x32x16_f16 terms, fp32 accumulation: fp32-class results (the default) -,
//     2 = two terms in the GCL edge models (opt-in).  Scales come from a-priori bounds (host: row-L1 norms; device: max |h|, |x|^2).
//   * per-atom GEMMs (node MLP, the next pass's projections): version 2 - atoms as MFMA rows on all eight waves, fragments from
//     L2 -, and, for a molecule of 33..55 atoms on one workgroup in the f16 modes, version 3 (stream_phase): atom-stationary,
//     a register-to-register chain on v_mfma_f32_16x16x32_f16 with the weights streamed through an LDS ring by four loader waves.
//   * SiLU is y * rcp(1 + exp2(y)) with y = -log2(e) * pre-activation, the constant folded into the packed weights on the host
//     (dl_model_create): one v_exp_f32 and one v_rcp_f32 per activation.
//   * HBM traffic inside a forward: the (L2-resident) weights and a per-workgroup scratch for the fp32 node features; inputs are
//     read once per chain, the kept frames written once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <cmath>

#include <algorithm>
#include <atomic>
#include <vector>

#include "../../include/difflinker_hip.h"
#include "pack_layout.h"


namespace {

constexpr int LDH = 132;            // LDS row stride of [n,128] fp32 tiles: 528 B = 33 x 16 B (conflict-free b128)
constexpr int NMAX = 55;            // atoms OWNED by one workgroup: rows of P, of the h / agg / hidden fragment rows, of z
constexpr int NQMAX = 2 * NMAX;     // atoms of one molecule in a TEAM of workgroups: sender rows Q, coordinates, compaction index
// Workgroup = 8 waves = two per SIMD: they own the 2 x 4 grid of 32x32 output tiles of every per-node GEMM and 32 pair slots
// each in the pair passes.
constexpr int THREADS = 512;
constexpr int NWAVES = THREADS / 64;
constexpr int GWAVES = 8;
constexpr int GTHREADS = 64 * GWAVES;
// ---- LDS layout (floats) ---------------------------------------------------------------------------
constexpr int L_A = 0;                                // P (own atoms) / eps (aliased at the end)
constexpr int L_B = L_A + NMAX * LDH;                 // Q (one workgroup per molecule: <= 55 rows; a team: all atoms, up to 110
constexpr int L_C = L_B + NMAX * LDH;                 //    rows through L_C) / node-MLP hidden | h fragment rows / message aggregate
constexpr int L_W = L_C + NMAX * LDH;                 // 128x128 second-layer weights, [k][c][nt]
constexpr int L_VEC = L_W + UNIT;                     // wr', wd', b2'|b6', w7'
constexpr int L_XS = L_VEC + 4 * HID;                 // current coordinates of every atom of the molecule [n][4]
constexpr int L_X0 = L_XS + NQMAX * 4;                // coordinates at forward entry [n][4]
constexpr int L_Z = L_X0 + NQMAX * 4;                 // per-atom state z of the own atoms [n_own][DMAX]
constexpr int L_LM = L_Z + NMAX * DMAX;               // linker mask of the own atoms
constexpr int L_FRAG = L_LM + 56;                     // fragment mask of the own atoms
constexpr int L_IDX = L_FRAG + 56;                    // compacted atom -> padded position, every atom of the molecule (int)
constexpr int L_RCV = L_IDX + 112;                    // coordinate-pass receivers: list position -> own atom (int)
constexpr int L_RPOS = L_RCV + 56;                    // own atom -> position in that list, -1: not a receiver (int)
constexpr int L_MISC = L_RPOS + 56;                   // ints: [0] n_b, [1] nan bits, [2..15] team block (TM_*), [16..] pass context (CX_*)
constexpr int MISC_WORDS = 40;
constexpr int L_FMAX = L_MISC + MISC_WORDS;           // f16x3 magnitude bounds (float bits, atomicMax)
constexpr int L_DUMMY = L_FMAX + 8;                   // sink row for the stores of tile rows >= n_own (branch-free)
constexpr int L_PROG = L_DUMMY + LDH;                 // pair-loop progress of the 8 waves (fair sharing of a SIMD between its two waves)
constexpr int L_TOTAL = L_PROG + 8;
// A row of 128 zeros for the kernels with ONE workgroup per molecule: the coordinate rows 56..87 of v.xs, which only a team uses (LDS
// has not a byte to spare).  Slots and steps of a pair loop that hold no pair (the last wave's spare slots, a last chunk one sender
// short: 10 % of the slot-steps of the C2 batch) read their P and Q rows there and take r = d0 = 0: the first layer comes out as
// SiLU(0) = 0, the second layer multiplies ZERO columns - under the power cap the energy an idle column does not burn is clock.
constexpr int L_ZROW = L_XS + 4 * 56;
static_assert(L_ZROW + HID <= L_X0 && 56 >= NMAX + 1 && (L_ZROW % 4) == 0, "the zero row sits in the team-only part of the coordinate rows");
constexpr size_t LDS_BYTES = size_t(L_TOTAL) * 4;
static_assert(LDS_BYTES <= 163840, "LDS layout exceeds 160 KiB");
static_assert((L_B % 4) == 0 && (L_C % 4) == 0 && (L_W % 4) == 0 && (L_VEC % 4) == 0 && (L_XS % 4) == 0 &&
              (L_X0 % 4) == 0 && (L_Z % 4) == 0 && (L_DUMMY % 4) == 0, "16-byte alignment");
// f16 modes (per-atom phases version 3, stream_phase): what a coordinate pass leaves in LDS for its reduction lives BEHIND the
// weight ring (the first 64 KB from L_A, and the W2' region) - the loaders start as soon as the pair loop is over
constexpr int L_TRIP3 = L_A + 2 * ST_CHUNK;        // slot triples [256][4]
constexpr int L_AGGX3 = L_TRIP3 + 4 * 256;         // coordinate aggregate of the own atoms [55][4]
static_assert(L_AGGX3 + 4 * (NMAX + 1) <= L_W && (L_TRIP3 % 4) == 0, "version-3 coordinate scratch must fit the h region behind the ring");

// Team kernels (several workgroups share one molecule, see team_sync): words of v.misc.  Everything a team member needs
// is re-read from here at the point of use, so nothing of it lives in registers across the pair loops.
constexpr int TM_EPOCH = 2;      // exchanges completed so far
constexpr int TM_FAIL = 3;       // a team-mate did not show up in time
constexpr int TM_ROWS = 4;       // exchange rows of this molecule (device pointer: lo, hi)
constexpr int TM_FLAGS = 6;      // arrival words of this molecule's workgroups (device pointer: lo, hi)
constexpr int TM_S = 8, TM_RANK = 9, TM_NOWN = 10;   // team size, own index, number of own atoms (one workgroup: 1, 0, n_b)
constexpr int MS_NRCV = 12;      // (every kernel) length of the coordinate-pass receiver list v.rcv
constexpr int MS_FULL = 13;      // (every kernel) the receiver list holds EVERY own atom (no linker mask, or a skipped sum could not be proven finite)
constexpr int CX_V3 = 38;        // (pass context, below) this workgroup runs the per-atom phases version 3: see forward_molecule2

struct Lds {
    float *A, *B, *C, *W, *vec, *xs, *x0, *aggx, *z, *lm, *frag;
    int *idx, *rcv, *rpos, *misc;
    unsigned* fmax;
    float* dummy;
    int w;                 // this wave's index in the workgroup (an SGPR: see lane_ids)
};

__device__ __forceinline__ Lds lds_view(float* base) {
    Lds v;
    v.w = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
    v.A = base + L_A; v.B = base + L_B; v.C = base + L_C; v.W = base + L_W; v.vec = base + L_VEC;
    v.xs = base + L_XS; v.x0 = base + L_X0; v.z = base + L_Z;
    v.aggx = base + L_B;                       // coordinate aggregate of the own atoms [n_own][4]: over Q, dead once a pair loop is over
    v.lm = base + L_LM; v.frag = base + L_FRAG;
    v.idx = reinterpret_cast<int*>(base + L_IDX); v.misc = reinterpret_cast<int*>(base + L_MISC);
    v.rcv = reinterpret_cast<int*>(base + L_RCV); v.rpos = reinterpret_cast<int*>(base + L_RPOS);
    v.fmax = reinterpret_cast<unsigned*>(base + L_FMAX);
    v.dummy = base + L_DUMMY;
    return v;
}

// Optional phase timeline (diagnostics builds only: -DDL_PROFILE, scripts/phase_timeline.py): lane 0 of every wave of
// block 0 logs (tag, s_memtime) pairs into buf[wave][event][2] (dl_set_profile_buffer).  Compiled out of the product
// library: even a never-taken branch per phase keeps the buffer pointer and the event counter alive across the pair
// loops, i.e. in scratch, and each of the ~110 sites per forward then pays a reload round trip.
constexpr int PROF_MAX_EVENTS = 1024;
struct Prof {
    unsigned long long* buf;
    int n;
};
__device__ __forceinline__ void prof_event(Prof& pf, int w, int lane, int tag) {
#ifdef DL_PROFILE
    if (pf.buf != nullptr) {
        if (lane == 0 && pf.n < PROF_MAX_EVENTS) {
            unsigned long long* e = pf.buf + (size_t(w) * PROF_MAX_EVENTS + pf.n) * 2;
            e[0] = (unsigned long long)tag;
            e[1] = __builtin_amdgcn_s_memtime();
        }
        pf.n++;
    }
#else
    (void)pf; (void)w; (void)lane; (void)tag;
#endif
}

// the chunk-level events of stream_phase (an event costs a few hundred cycles - s_memtime, two stores - and ~20 of them sit in a
// phase of ~20 K cycles): -DDL_PROFILE_FINE builds only
__device__ __forceinline__ void fine_event(Prof& pf, int w, int lane, int tag) {
#ifdef DL_PROFILE_FINE
    prof_event(pf, w, lane, tag);
#else
    (void)pf; (void)w; (void)lane; (void)tag;
#endif
}

// events INSIDE the pair loop perturb it (the profile state lives across the loop): -DDL_PROFILE_LOOP builds only
__device__ __forceinline__ void loop_event(Prof& pf, int w, int lane, int tag) {
#ifdef DL_PROFILE_LOOP
    prof_event(pf, w, lane, tag);
#else
    (void)pf; (void)w; (void)lane; (void)tag;
#endif
}

// ---- f16x3 path (PREC 1): every fp32 operand of a 128-wide contraction is scaled by a power of two into the
// fp16 range and split x*s = hi + lo (both fp16); a product is taken as hi*hi' + hi*lo' + lo*hi' on the fp16
// matrix pipe (v_mfma_f32_32x32x16_f16, fp32 accumulate) and the accumulator is scaled back exactly.  The
// split keeps ~21 significant bits per operand (error ~1e-6 per product, fp32-class: measured below the
// fp32-vs-fp64 noise on a 100-step chain), unlike a bf16 split (16 bits).  Scales: static per weight matrix
// (host, dl_model_create); dynamic per pass for the activations from a magnitude bound kept in LDS.
// Unlike the f32-input MFMA - which runs at the fp32 VECTOR rate and did not overlap with this kernel's VALU
// work - the fp16 MFMA is 16x faster, so the edge pass becomes VALU-bound (SiLU + the splits).
// workgroup-wide max of non-negative floats into an LDS slot (slot zeroed earlier; read after a barrier)
// (DPP / permlane steps, not __shfl_xor: the six ds_bpermute index vectors of a shuffle reduction are loop-invariant, the
// compiler computes them once per kernel, they do not survive the pair loops in registers, and every reload from scratch
// waits for all vector-memory traffic in flight - measured round 3: 1-2 us per maximum behind a fragment prefetch)
__device__ __forceinline__ unsigned wave_max_u32(unsigned b) {
    b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xf, 0xf, false));      // quad_perm xor 1
    b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xf, 0xf, false));      // quad_perm xor 2
    b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x141, 0xf, 0xf, false));     // row_half_mirror
    b = max(b, (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x140, 0xf, 0xf, false));     // row_mirror: the 16-lane row
    const auto r16 = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    b = max((unsigned)r16[0], (unsigned)r16[1]);                                              // both rows of the 32-lane half
    const auto r32 = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return max((unsigned)r32[0], (unsigned)r32[1]);
}
// LDS atomic maximum as the bare instruction: behind atomicMax() the compiler puts s_waitcnt vmcnt(0) - i.e. a wait for every
// fragment prefetch and LDS-DMA in flight (1-2 us in the per-atom phases) - although an LDS maximum orders nothing against them
__device__ __forceinline__ void lds_max_u32(unsigned* slot, unsigned val) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)slot;
    asm volatile("ds_max_u32 %0, %1" :: "v"(addr), "v"(val) : "memory");
}
__device__ __forceinline__ void block_max(unsigned* slot, float val, int lane) {
    const unsigned b = wave_max_u32(__float_as_uint(val));
    if (lane == 0) lds_max_u32(slot, b);
}

// Workgroup barrier for LDS hand-offs that does NOT drain the vector-memory counter (unlike __syncthreads(), whose
// fence waits for every LDS-DMA and prefetch in flight): own LDS traffic retired, then s_barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// every global load / LDS-DMA this wave issued has landed (call before the barrier that publishes DMA data)
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Uniform scalars of the packed weights (scales, bounds) through the SCALAR cache: a vector load of them would queue behind
// every fragment prefetch and LDS-DMA in flight (vector memory returns in order) - s_load does not.
__device__ __forceinline__ float cload(const float* p, int k) {
    typedef const __attribute__((address_space(4))) float* cptr_t;
    return reinterpret_cast<cptr_t>(reinterpret_cast<uintptr_t>(p))[k];
}

// LDS-DMA (global_load_lds_dwordx4: global -> LDS without staging registers, 1 KB per wave instruction) of the
// [k][c][nt] image of a 128x128 matrix into v.W and of the four 128-vectors at `vecs` into v.vec (a GCL uses three;
// the fourth slot then receives the next packed vector, unused).  Grid waves only; issued as soon as the previous
// pair loop has released v.W / v.vec, so the image lands under the node phases.
__device__ __forceinline__ void stage_dma(const Lds& v, const float* __restrict__ wimg, const float* __restrict__ vecs,
                                          const float* __restrict__ vec4, int w, int tid) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    if (w < 4 * HID / 256) {              // wave-uniform; first, so that no wait the compiler adds covers the image
        // lanes 0..95: wr', wd', b2'|b6' (contiguous at `vecs`); lanes 96..127: the fourth vector (w7' / w_att') at `vec4`
        const float4* src = (tid < 96) ? reinterpret_cast<const float4*>(vecs) + tid
                                       : reinterpret_cast<const float4*>(vec4) + (tid - 96);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(v.vec + 256 * w), 16, 0, 0);
    }
    const float4* src = reinterpret_cast<const float4*>(wimg);
#pragma unroll
    for (int it = 0; it < UNIT / 4 / GTHREADS; ++it)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + it * GTHREADS + tid), (lptr_t)(v.W + 4 * (it * GTHREADS + 64 * w)),
                                         16, 0, 0);
}

// Per-lane indices re-derived where they are used.  Everything that is computed from the lane index before a pair loop and used
// again after it would have to live across the loop - i.e. be spilled (the loop takes every VGPR) and reloaded one by one, each
// reload paying an L2 round trip behind whatever vector-memory traffic is queued.  Until round 6 they were re-derived from an opaque
// copy of threadIdx.x - but that register (v0 at kernel entry) then lives across the whole kernel itself: it was spilled at
// entry and reloaded from scratch at ~40 places, an s_waitcnt vmcnt(0) each.  Now: the lane from v_mbcnt (of an opaque zero, so
// that no two derivations are merged into one long-lived value), the wave index from an SGPR set at kernel entry (Lds::w).
struct LaneIds {
    int tid, w, lane, c, hh, nt, mt;
};
__device__ __forceinline__ LaneIds lane_ids(const Lds& v) {
    int z = 0;
    asm volatile("" : "+v"(z));
    LaneIds q;
    q.lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
    q.w = v.w;
    asm volatile("" : "+s"(q.w));              // (opaque too: what is derived from it stays local to the place that derives it)
    q.tid = 64 * q.w + q.lane;
    asm volatile("" : "+v"(q.tid));            // (opaque: with the sum visible the compiler splits every index derived from it into a
                                               // scalar and a vector part and doubles the kernel's SGPR spills)
    q.c = q.lane & 31; q.hh = q.lane >> 5;
    q.nt = q.w & 3; q.mt = q.w >> 2;
    return q;
}
// what the NEXT pass needs prefetched (base == nullptr: nothing follows)
struct NextPass {
    const float* base;
    bool equiv;
};
__device__ __forceinline__ void stage_next(const Lds& v, const NextPass& nx, int w, int tid) {
    if (nx.base == nullptr) return;
    stage_dma(v, nx.base + (nx.equiv ? E_W6T : G_W2T), nx.base + (nx.equiv ? E_VEC : G_VEC) + HID,
              nx.base + (nx.equiv ? E_VEC + 4 * HID : G_VEC + 6 * HID), w, tid);
}

// store one accumulator element of tile row `row` to a [n][LDH] buffer; rows >= n_b go to the sink row
__device__ __forceinline__ void store_row(const Lds& v, float* buf, int row, int nb, int col, float val) {
    float* p = (row < nb) ? buf + row * LDH + col : v.dummy + col;
    *p = val;
}

// f16x3 magnitude-bound slots in v.fmax (float bits of non-negative maxima)
constexpr int FM_H0 = 0, FM_H1 = 1, FM_HG = 2, FM_AGG = 3, FM_XOWN = 4, FM_X2 = 5, FM_X02 = 6;
// FM_H0/1: max |h| over the OWN atoms (two slots alternate); FM_HG: over the whole molecule (teams: from the exchange headers);
// FM_X2: max |x|^2 over every atom (teams: from the exchange headers); FM_X02: the same at forward entry; FM_XOWN (teams): max |x|^2
// over the OWN atoms, kept where their coordinates change (forward entry, coordinate update) and published in the exchange header

// f16x3: common scale S1 of the rank-2 geometric term (r * wr' and d0 * wd' products in one accumulator); sc[6], sc[7] =
// max |wr'|, max |wd'|.  The sender rows Q are stored times S1 (node_pre) and enter that MFMA as its C operand.
template <bool TEAM>
__device__ __forceinline__ float geo_scale(const Lds& v, const float* __restrict__ sc) {
    const float x2 = __uint_as_float(v.fmax[FM_X2]), x02 = __uint_as_float(v.fmax[FM_X02]);
    return fminf(scale_for(4.0f * x2) * scale_for(sc[6]), scale_for(4.0f * x02) * scale_for(sc[7]));
}

// One pass over all n_b^2 ordered pairs of the molecule, "receiver-stationary":
//   * the 256 lane pairs (c, c+32) of the 8 waves are SLOTS; atom i owns g = min(256 / n_b, n_b) consecutive slots and
//     slot (i, chunk) walks the senders j = chunk*q + t, t = 0 .. q-1, q = ceil(n_b / g): one MFMA column per slot, the
//     receiving atom of a lane never changes, so the sum over j is a plain in-register accumulation (no cross-lane
//     reduction, no LDS traffic inside the loop) and its order is fixed (t ascending, then chunks ascending in
//     pair_reduce): DETERMINISTIC.
//   * both edge layers are computed TRANSPOSED, features x pairs: D1[f][pair] = (P_i + Q_j)[f] + wr[f] r + wd[f] d0 comes
//     out of the matrix pipe in the accumulator layout (lane = pair, registers = 16 features per 32-feature tile), which IS
//     the B-operand layout of the next MFMA (lane = column, registers = k) once the k-slots of W2' are permuted to match
//     (host: pack_lds_image*_t): SiLU(D1) feeds D2 = W2' * SiLU(D1) without any data movement, and D2 again has lane = pair,
//     so the message lands in the receiver's accumulators.  The rank-2 geometric term runs on the matrix pipe too
//     (f16x3: one split-fp16 MFMA per feature tile; fp32: one v_mfma_f32_32x32x2_f32).
//   EQUIV = false: agg[i][f]  = sum_j m_ij * u2_ij[f]              (GCL message sum)      -> partial rows
//   EQUIV = true : aggx[i]    = sum_j cdiff_ij * (w7'.u2_ij) * m_ij (coordinate head)      -> partial triples
// The partials of the g slots of an atom are written to LDS (over P, Q, H and W2', dead by then) after the barrier that
// ends the loop; pair_reduce_* adds them in chunk order.
constexpr int NSLOT = 32 * GWAVES;
constexpr int PB_STRIDE = LDH;                        // partial rows [NSLOT][132] from L_A on: 135 KB <= A + B + C + W
static_assert(NSLOT * PB_STRIDE <= L_W + UNIT, "partial-sum buffer must fit the P, Q, H, W2' regions");

struct SlotPlan {
    int g, q;
};
// nrec receivers share the 256 slots (all n_b atoms of the molecule, or this workgroup's part of them in a team)
__device__ __forceinline__ SlotPlan slot_plan(int nrec, int nb) {
    SlotPlan sp;
    sp.g = min(NSLOT / max(nrec, 1), nb);
    sp.q = (nb + sp.g - 1) / sp.g;
    // ... and of the plans with that many steps the one with the FEWEST slots: a wave without slots skips the loop (8 linker atoms
    // as receivers of 50 senders: 25 instead of 32 slots each, 7 waves instead of 8; 36 atoms: 6 instead of 7, 7 waves) - a step
    // costs its energy per ACTIVE wave, and under the power cap energy is time (round 6)
    sp.g = (nb + sp.q - 1) / max(sp.q, 1);
    // (one or two MORE steps where that takes fewer wave-steps still - 38 atoms: 6 waves x 8 instead of 8 x 7 - measured slower, +5 %:
    // the number of steps is the critical path; profiles/r06/ab_fewest_slots_plan.log)
    return sp;
}

// fp16 hi (truncated) / lo (nearest) fragments of 8 scaled values; the hi part as fp32 is the value with its low 13
// mantissa bits cleared (v_and), which equals what v_cvt_pkrtz keeps for everything in the normal fp16 range
__device__ __forceinline__ void split8t(const float (&u)[8], uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = u[2 * q], b = u[2 * q + 1];
        const float ah = __uint_as_float(__float_as_uint(a) & 0xffffe000u), bh = __uint_as_float(__float_as_uint(b) & 0xffffe000u);
        h[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ah, bh));
        l[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a - ah, b - bh));
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// ATT (GCL only): edge attention m_ij *= sigmoid(w_att . m_ij + b_att) (egnn.py:52-54); `head` = b_att.
// EQUIV: `head` = coords_range when the coordinate head goes through tanh (egnn.py:104-105), 0 otherwise.
template <bool EQUIV, int PREC, bool ATT, bool TEAM>
__device__ __forceinline__ void pair_phase(const Lds& v, int nb, int w, int lane, const int8_t* __restrict__ emask, int N,
                                           float norm_constant, float sa, float inv_scale, const float* __restrict__ sc,
                                           float head, Prof& pf) {
    const int c = lane & 31, hh = lane >> 5;
    // receivers of this workgroup: its OWN atoms (one workgroup per molecule: every atom; a team: atoms rank, rank + S, ...).
    // GCL: all of them.  Coordinate head: only those whose update survives the linker mask - the reference multiplies the sum
    // of every other atom by zero (egnn.py:113-116) - i.e. the entries of the list v.rcv.  Senders: every atom of the molecule.
    const int nrec = EQUIV ? v.misc[MS_NRCV] : (TEAM ? v.misc[TM_NOWN] : nb);
    const int S = TEAM ? v.misc[TM_S] : 1, rank = TEAM ? v.misc[TM_RANK] : 0;
    const SlotPlan pl = slot_plan(nrec, nb);
    const int q = pl.q;
    const int slot = 32 * w + c;
    const bool slot_ok = slot < nrec * pl.g;
    const int il = slot_ok ? slot / pl.g : 0;
    const int li = EQUIV ? (slot_ok ? v.rcv[il] : 0) : il;        // the receiving atom among the own atoms: its row of P
    const int i = rank + li * S;                                  // ... and in the molecule: coordinates, row of the edge mask
    const int j0 = slot_ok ? (slot - il * pl.g) * q : 0;
    const int jn = slot_ok ? min(q, nb - j0) : 0;                 // senders this slot really has (may be <= 0)
    const bool wave_active = 32 * w < nrec * pl.g;                // wave-uniform
    float* pb = v.A + slot * PB_STRIDE;

    floatx16 agg[4];
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) agg[mt] = splat16(0.0f);

    if (wave_active) {
        const float4 xi = *reinterpret_cast<const float4*>(v.xs + 4 * i);
        const float4 yi = *reinterpret_cast<const float4*>(v.x0 + 4 * i);
        const int8_t* mrow = emask ? emask + v.idx[i] * N : nullptr;
        const float* Pp_ = v.A + li * LDH + 4 * hh;
        const float* bias_p_ = v.vec + 2 * HID + 4 * hh;
        const float* w7_p_ = v.vec + 3 * HID + 4 * hh;
        // the hidden features of k-slab s enter the second layer times sa * 2^n_s (balanced packing: the columns of W2' carry 2^-n_s)
        float isa_s[8];
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_)
            isa_s[s_] = (PREC == 0) ? 1.0f
                : __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(inv_pow2(sa * cload(sc, (EQUIV ? ES_NE : GS_NE) + s_)))));

        // rank-2 geometric term wr'[f] r + wd'[f] d0 as an MFMA: k-slots of half 0 carry r, of half 1 carry d0
        float ga[4];                 // fp32: A operand (k = hh)
        uint4 gaf[4];                // f16x3: A fragments {hi, lo | hi, 0 | 0 ...} x {hi, hi | lo, 0 | 0 ...} of the pair side
        float sX = 1.0f, invS1 = 1.0f;
        if constexpr (PREC == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) ga[mt] = v.vec[(hh ? HID : 0) + 32 * mt + c];
        } else {
            const float s_wr = scale_for(sc[6]), s_wd = scale_for(sc[7]);
            const float S1 = geo_scale<TEAM>(v, sc);
            invS1 = inv_pow2(S1);
            sX = S1 * inv_pow2(hh ? s_wd : s_wr);
            const float s_w = hh ? s_wd : s_wr;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float wv = v.vec[(hh ? HID : 0) + 32 * mt + c] * s_w;
                const float wh = __uint_as_float(__float_as_uint(wv) & 0xffffe000u);
                const unsigned d0w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(wh, wv - wh));
                const unsigned d1w = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(wh, 0.0f));
                gaf[mt] = make_uint4(d0w, d1w, 0u, 0u);
            }
        }

        // raw mask byte of step t; it is converted (and zeroed past the end of the slot's range) where it is used, a whole
        // step later, so the wait for the global load lands there and not right behind its issue
        auto load_mask = [&](int t) -> int {
            if (mrow == nullptr) return 1;
            const int jj = (t < jn) ? j0 + t : j0;                 // always a valid row (the slot is empty when jn <= 0)
            return (jn > 0) ? int(mrow[v.idx[min(jj, nb - 1)]]) : 0;       // steps past the range: see `ok` at the uses
        };
        int m_next = load_mask(0);
        // The SIMD arbitrates VALU / MFMA issue between its two waves by priority, then AGE: with equal priorities waves 0-3 run a
        // step in ~8.1 K ticks beside waves 4-7 (~14 K), leave the loop after 65 % of a pass and wait at the barrier while their
        // partners finish alone - one wave per SIMD, at half the machine's rate (profiles/r04/wave_timeline.log).  Each wave
        // publishes its step count and, at every step, takes the higher priority when it is not ahead of its partner.
        // (the two waves of a SIMD are w and w ^ 4: a workgroup's waves are dealt to the four SIMDs round-robin - observed, not
        // architected; a different mapping would only make the hint useless.  The slots are read and written through a
        // volatile LDS pointer - nothing may be cached in a register across steps - and are zero when a pass starts: cleared
        // at kernel entry and again behind the barrier that ends every loop)
        typedef volatile __attribute__((address_space(3))) int* lds_vint_t;
        lds_vint_t prog = (lds_vint_t)(reinterpret_cast<int*>(v.A - L_A + L_PROG));

        for (int t = 0; t < q; ++t) {
            // the P row, the bias vectors and all of W2' do not depend on t: without an opaque offset the compiler hoists
            // ~100 LDS reads out of the loop into registers it does not have (= scratch) and reloads them every iteration
            int opq = 0;
            asm volatile("" : "+s"(opq));
            // (one workgroup per molecule, f16 modes) a slot-step without a pair reads the zero row: see L_ZROW
            constexpr bool ZIDLE = !TEAM && PREC != 0;
            const float* zrow = v.A - L_A + L_ZROW + 4 * hh;
            const float* Pp = ((ZIDLE && t >= jn) ? zrow : Pp_) + opq;
            const float* bias_p = bias_p_ + opq;
            const float* w7_p = w7_p_ + opq;
            loop_event(pf, w, lane, 50);
            {
                if (lane == 0) prog[w] = t;
                const int other = __builtin_amdgcn_readfirstlane(prog[w ^ 4]);
                const bool behind = (w >= 4) ? (t <= other) : (t < other);          // a tie goes to the younger wave (age favours the older)
                if (behind) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            const bool ok = t < jn;
            const int j = ok ? j0 + t : 0;
            const int m_raw = m_next;
            m_next = load_mask(t + 1);
            const float4 xj = *reinterpret_cast<const float4*>(v.xs + 4 * j);
            const float4 yj = *reinterpret_cast<const float4*>(v.x0 + 4 * j);
            const float dx = xi.x - xj.x, dy = xi.y - xj.y, dz = xi.z - xj.z;
            const float ex = yi.x - yj.x, ey = yi.y - yj.y, ez = yi.z - yj.z;
            const float r = dx * dx + dy * dy + dz * dz;             // squared distance, current x  (egnn.py:298)
            const float d0 = ex * ex + ey * ey + ez * ez;            // squared distance at forward entry (:220)
            const float* Qp = (ZIDLE && !ok) ? zrow : v.B + j * LDH + 4 * hh;
            float ssum = 0.0f;

            // bias (and w7') vectors of a 32-feature tile: four broadcast float4 per lane
            struct TileVecs { float4 b[4], w[4]; };
            auto load_vecs = [&](int mt) {
                TileVecs tv;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    tv.b[qq] = *reinterpret_cast<const float4*>(bias_p + 32 * mt + 8 * qq);
                    if (EQUIV || ATT) tv.w[qq] = *reinterpret_cast<const float4*>(w7_p + 32 * mt + 8 * qq);
                }
                return tv;
            };
            // the loaded values exist HERE (the compiler otherwise sinks each broadcast read next to its first use and waits for
            // it there: eight exposed LDS latencies per half tile)
            auto keep_vecs = [&](TileVecs& tv) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    asm volatile("" : "+v"(tv.b[qq].x), "+v"(tv.b[qq].y), "+v"(tv.b[qq].z), "+v"(tv.b[qq].w));
                    if (EQUIV || ATT) asm volatile("" : "+v"(tv.w[qq].x), "+v"(tv.w[qq].y), "+v"(tv.w[qq].z), "+v"(tv.w[qq].w));
                }
            };
            // SiLU + mask + sum over senders (GCL) / w7' dot (coordinate head) of one 32-feature tile of D2
            auto epilogue = [&](floatx16& c2, int mt, const TileVecs& tv, float m) {
                const float im = (m != 0.0f) ? __builtin_amdgcn_rcpf(m) : 1.2676506e30f;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float bb[4] = {tv.b[qq].x, tv.b[qq].y, tv.b[qq].z, tv.b[qq].w};
                    float ww[4] = {0.f, 0.f, 0.f, 0.f};
                    if (EQUIV || ATT) { ww[0] = tv.w[qq].x; ww[1] = tv.w[qq].y; ww[2] = tv.w[qq].z; ww[3] = tv.w[qq].w; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int reg = 4 * qq + k;
                        const float y2 = (PREC == 0) ? c2[reg] : fmaf(c2[reg], inv_scale, bb[k]);
                        if (PREC != 0 && !EQUIV && !ATT) {
                            // m * y2 / (1 + 2^y2) with the multiplier inside the reciprocal: im = 1 / m (collate's edge_mask holds
                            // -1 and, on the diagonal, -2: exact), 2^100 for m = 0 (the term is < 2^-100 |y2|: nothing in fp32 sums)
                            agg[mt][reg] = fmaf(y2, __builtin_amdgcn_rcpf(fmaf(__builtin_amdgcn_exp2f(y2), im, im)), agg[mt][reg]);
                            continue;
                        }
                        const float u2 = silu_u(y2);
                        if (EQUIV || ATT) ssum = fmaf(ww[k], u2, ssum);
                        if (ATT) c2[reg] = u2;                       // the message waits for its attention weight
                        else if (!EQUIV) agg[mt][reg] = fmaf(m, u2, agg[mt][reg]);
                    }
                }
            };
            // attention: one logit per pair over all 128 features (the other lane half holds the other 64), then the masked,
            // weighted message joins the receiver's sums
            auto attend = [&](floatx16 (&c2)[4], float m) {
                float lo, hi;
                both_halves(ssum, lo, hi);
                const float logit = (lo + hi) + head;
                const float mw = m * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * logit));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) agg[mt][reg] = fmaf(mw, c2[mt][reg], agg[mt][reg]);
            };

            if constexpr (PREC == 0) {
                // ---- exact fp32: v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate and blocks the VALU of both waves of the
                // SIMD (profiles/r02/ubench_pair_overlap.log), so there is nothing to interleave: phases in dependency order.
                // a1[mt][reg] = pre-activation of feature 32mt + (reg&3) + 8(reg>>2) + 4hh of this lane's pair
                floatx16 a1[4];
                const float X = hh ? d0 : r;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const float4 P = *reinterpret_cast<const float4*>(Pp + 32 * mt + 8 * qq);
                        const float4 Q = *reinterpret_cast<const float4*>(Qp + 32 * mt + 8 * qq);
                        a1[mt][4 * qq + 0] = P.x + Q.x; a1[mt][4 * qq + 1] = P.y + Q.y;
                        a1[mt][4 * qq + 2] = P.z + Q.z; a1[mt][4 * qq + 3] = P.w + Q.w;
                    }
                    a1[mt] = mfma32(ga[mt], X, a1[mt]);
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) a1[mt][reg] = silu_u(a1[mt][reg]);
                // second layer, transposed: D2[f][pair] = b2'[f] + sum_k W2'[f][k] u[k][pair]
                floatx16 c2[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const float4 b4 = *reinterpret_cast<const float4*>(bias_p + 32 * mt + 8 * qq);
                        c2[mt][4 * qq] = b4.x; c2[mt][4 * qq + 1] = b4.y; c2[mt][4 * qq + 2] = b4.z; c2[mt][4 * qq + 3] = b4.w;
                    }
                const float4* Wp = reinterpret_cast<const float4*>(v.W) + lane + opq;
#pragma unroll
                for (int s2 = 0; s2 < 64; ++s2) {
                    const float4 a4 = Wp[s2 * 64];
                    const float bv = a1[s2 >> 4][s2 & 15];
                    c2[0] = mfma32(a4.x, bv, c2[0]);
                    c2[1] = mfma32(a4.y, bv, c2[1]);
                    c2[2] = mfma32(a4.z, bv, c2[2]);
                    c2[3] = mfma32(a4.w, bv, c2[3]);
                }
                const float m = ok ? float(m_raw) : 0.0f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    TileVecs tv = load_vecs(mt);
                    epilogue(c2[mt], mt, tv, m);
                }
                if (ATT) attend(c2, m);
            } else {
                // ---- f16x3, software-pipelined IN the wave.  A wave issues at most one VALU instruction every ~4.6 cycles and
                // is blocked at an MFMA while the matrix pipe is busy, so a phase-separated stream (all SiLUs, then all MFMAs)
                // costs VALU time + MFMA time per wave, and the SIMD's second wave - running the same phases - hides little of
                // it (measured: 10.8 K cycles per step alone, 16.6 K with the partner; profiles/r02).  Here the first layer is
                // produced one 16-feature k-slab at a time and the 12 MFMAs that consume slab s are issued one by one BETWEEN
                // the three VALU chunks of each element pair of slab s+1; sched_barrier after every chunk pins the order.
                // Only slab 0 is produced without MFMAs beside it and only slab 7's MFMAs have no VALU beside them.
                // Fragments are double-buffered (2 x 8 registers), W2' fragments and P/Q rows arrive one group ahead.
                constexpr bool TWO = (PREC == 2) && !EQUIV && !ATT;
                const float xv = (ZIDLE && !ok) ? 0.0f : (hh ? d0 : r) * sX;
                const float xh_ = __uint_as_float(__float_as_uint(xv) & 0xffffe000u);
                const uint4 xf = make_uint4(__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(xh_, xh_)),
                                            __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(xv - xh_, 0.0f)), 0u, 0u);
                const uint4* Wq = reinterpret_cast<const uint4*>(v.W) + lane + opq;
                floatx16 c2[4] = {splat16(0.0f), splat16(0.0f), splat16(0.0f), splat16(0.0f)};
                uint4 fh[2], fl[2];                // [slab parity]: B fragments hi / lo of one 16-feature k-slab
                float4 Pq[2];                      // P row of one quad (8 features of this half), double-buffered
                floatx16 Qc;                       // Q row (times S1) of the 32-feature tile whose geometric MFMA is next
                uint4 af[2][4];                    // W2' fragments of one (slab, output half): ah0, ah1, al0, al1
                floatx16 g1;                       // geometric term of the tile in production
                float yy[2], ee[2], uu[2];         // element pair in flight through the three chunks

                // quad gq = 4 * tile + quad of the tile (8 features of this lane half)
                auto load_pq = [&](int gq) { Pq[gq & 1] = *reinterpret_cast<const float4*>(Pp + 8 * gq); };
                auto load_qc = [&](int mt, int qq) {
                    const float4 t4 = *reinterpret_cast<const float4*>(Qp + 32 * mt + 8 * qq);
                    Qc[4 * qq] = t4.x; Qc[4 * qq + 1] = t4.y; Qc[4 * qq + 2] = t4.z; Qc[4 * qq + 3] = t4.w;
                };
                auto load_a = [&](int slab, int oh) {
                    const int buf = oh;
                    af[buf][0] = Wq[(slab * 4 + 2 * oh) * 64]; af[buf][1] = Wq[(slab * 4 + 2 * oh + 1) * 64];
                    af[buf][2] = Wq[((8 + slab) * 4 + 2 * oh) * 64]; af[buf][3] = Wq[((8 + slab) * 4 + 2 * oh + 1) * 64];
                };
                // the three VALU chunks of element pair ge = 8 * tile + e (registers 2e, 2e+1 of the tile's accumulator)
                auto chunk_a = [&](int ge) {
                    const int e = ge & 7, gq = ge >> 1;
                    const float4 P = Pq[gq & 1];
                    const float p0 = (e & 1) ? P.z : P.x, p1 = (e & 1) ? P.w : P.y;
                    yy[0] = fmaf(g1[2 * e], invS1, p0); yy[1] = fmaf(g1[2 * e + 1], invS1, p1);
                    ee[0] = __builtin_amdgcn_exp2f(yy[0]); ee[1] = __builtin_amdgcn_exp2f(yy[1]);
                };
                auto chunk_b = [&](int ge) {
                    // SiLU in u-form, scaled into the fp16 range: y * (sa 2^n) / (1 + 2^y), n = the exponent of the element's k-slab
                    const float isa = isa_s[(ge >> 2) & 7];
                    ee[0] = __builtin_amdgcn_rcpf(fmaf(ee[0], isa, isa)); ee[1] = __builtin_amdgcn_rcpf(fmaf(ee[1], isa, isa));
                };
                auto chunk_c = [&](int ge) {
                    uu[0] = yy[0] * ee[0]; uu[1] = yy[1] * ee[1];
                    if constexpr (TWO) {
                        // activation rounded to nearest fp16 (v_cvt_pk_f16_f32), no lo part: a_rn * (W_hi + W_lo)
                        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                        typedef float float2v __attribute__((ext_vector_type(2)));
                        const float2v u2 = {uu[0], uu[1]};
                        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(u2, half2v));
                        const int buf = (ge >> 2) & 1, d = ge & 3;
                        if (d == 0) fh[buf].x = hp;
                        if (d == 1) fh[buf].y = hp;
                        if (d == 2) fh[buf].z = hp;
                        if (d == 3) fh[buf].w = hp;
                        return;
                    }
                    const float h0 = __uint_as_float(__float_as_uint(uu[0]) & 0xffffe000u);
                    const float h1 = __uint_as_float(__float_as_uint(uu[1]) & 0xffffe000u);
                    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h0, h1));
                    const unsigned lp = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(uu[0] - h0, uu[1] - h1));
                    const int buf = (ge >> 2) & 1, d = ge & 3;      // slab parity, dword of the fragment
                    if (d == 0) { fh[buf].x = hp; fl[buf].x = lp; }
                    if (d == 1) { fh[buf].y = hp; fl[buf].y = lp; }
                    if (d == 2) { fh[buf].z = hp; fl[buf].z = lp; }
                    if (d == 3) { fh[buf].w = hp; fl[buf].w = lp; }
                };
                // MFMA number i (0..11) of the stage that consumes k-slab s: group = output half, six per group
                auto mfma_i = [&](int s, int i) {
                    const int oh = i / 6, m6 = i % 6;
                    const uint4& a = af[oh][(m6 < 2 ? 2 : 0) + (m6 & 1)];          // lo, lo, hi, hi, hi, hi
                    const uint4& b = (m6 == 2 || m6 == 3) ? fl[s & 1] : fh[s & 1];
                    c2[2 * oh + (m6 & 1)] = mfma_h(a, b, c2[2 * oh + (m6 & 1)]);
                };

                // prologue: k-slab 0 of the first layer (no MFMAs to hide under yet), first W2' group, first rows
                load_pq(0);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) load_qc(0, qq);
                g1 = mfma_h(gaf[0], xf, Qc);
                load_a(0, 0);
#pragma unroll
                for (int ge = 0; ge < 4; ++ge) {
                    if ((ge & 1) == 0) load_pq((ge >> 1) + 1);
                    chunk_a(ge); chunk_b(ge); chunk_c(ge);
                }
                __builtin_amdgcn_sched_barrier(0);
                loop_event(pf, w, lane, 51);
                if constexpr (TWO) {
                    // two-term stages: 8 MFMAs per k-slab (W_lo' x a, W_hi' x a for the four output tiles), the 12 chunks of
                    // k-slab s+1 spread over them (1, 2, 1, 2, ...)
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int oh = i / 4, m4 = i % 4;
                            if (m4 == 0) {
                                const int ng = 2 * s + oh + 1;
                                if (ng < 16) load_a(ng >> 1, ng & 1);
                            }
                            {
                                const uint4& a = af[oh][(m4 < 2 ? 2 : 0) + (m4 & 1)];          // lo, lo, hi, hi
                                c2[2 * oh + (m4 & 1)] = mfma_h(a, fh[s & 1], c2[2 * oh + (m4 & 1)]);
                            }
                            if (i < 4 && s < 6 && !(s & 1)) load_qc((s + 2) >> 1, i);
                            if (i == 7 && s < 6 && !(s & 1)) g1 = mfma_h(gaf[(s + 2) >> 1], xf, Qc);
                            if (s < 7) {
#pragma unroll
                                for (int k = (3 * i) / 2; k < (3 * (i + 1)) / 2; ++k) {
                                    const int ge = 4 * (s + 1) + k / 3, ph = k % 3;
                                    if (ph == 0) { if ((ge & 1) == 0 && (ge >> 1) + 1 < 16) load_pq((ge >> 1) + 1); chunk_a(ge); }
                                    if (ph == 1) chunk_b(ge);
                                    if (ph == 2) chunk_c(ge);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    // stage s: the 12 MFMAs of k-slab s, k-slab s+1 of the first layer in their shadow (one chunk per MFMA)
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        const int m6 = i % 6, ge = 4 * (s + 1) + i / 3, ph = i % 3;
                        // W2' fragments of the next group one group ahead
                        if (m6 == 0) {
                            const int ng = 2 * s + i / 6 + 1;
                            if (ng < 16) load_a(ng >> 1, ng & 1);
                        }
                        mfma_i(s, i);
                        // the slab after the next opens a tile: its geometric term (g1's last element was read at i = 9)
                        // (its Q row was requested under the first four MFMAs of this stage)
                        if (i < 4 && s < 6 && !(s & 1)) load_qc((s + 2) >> 1, i);
                        if (i == 10 && s < 6 && !(s & 1)) g1 = mfma_h(gaf[(s + 2) >> 1], xf, Qc);
                        if (s < 7) {
                            if (ph == 0) { if ((ge & 1) == 0 && (ge >> 1) + 1 < 16) load_pq((ge >> 1) + 1); chunk_a(ge); }
                            if (ph == 1) chunk_b(ge);
                            if (ph == 2) chunk_c(ge);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                loop_event(pf, w, lane, 52);
                const float m = ok ? float(m_raw) : 0.0f;
                TileVecs tv[2];                                 // two named buffers: no register copies between tiles
                tv[0] = load_vecs(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    keep_vecs(tv[mt & 1]);
                    if (mt < 3) tv[(mt + 1) & 1] = load_vecs(mt + 1);       // lands under this tile's epilogue
                    __builtin_amdgcn_sched_barrier(0);
                    epilogue(c2[mt], mt, tv[mt & 1], m);
                }
                if (ATT) attend(c2, m);
                loop_event(pf, w, lane, 53);
            }
            if (EQUIV) {
                // s = w7'.u2 over all 128 features: this lane summed its half's 64, the other half holds the rest
                float lo, hi;
                both_halves(ssum, lo, hi);
                float s_all = lo + hi;
                if (head != 0.0f)                                   // tanh(s) * coords_range (egnn.py:104-105); wave-uniform
                    s_all = head * (1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * s_all)));
                // coord_diff = (x_i - x_j) / (sqrt(r + 1e-8) + norm_constant)   (egnn.py:299-300)
                const float den = sqrtf(r + 1e-8f) + norm_constant;
                const float f = ok ? s_all * float(m_raw) : 0.0f;
                ax += (dx / den) * f; ay += (dy / den) * f; az += (dz / den) * f;
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    prof_event(pf, w, lane, EQUIV ? 121 : 120);       // (diagnostics builds) this wave left the loop; the barrier follows
    lds_barrier();                         // every wave left the loop: P (v.A), Q (v.B), v.W are dead -> partial sums
    if (lane == 0) reinterpret_cast<int*>(v.A - L_A + L_PROG)[w] = 0;      // progress slots: zero for the next pass (nobody reads them before its loop)
    if (slot_ok) {
        if (!EQUIV) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    *reinterpret_cast<float4*>(pb + 32 * mt + 8 * qq + 4 * hh) =
                        make_float4(agg[mt][4 * qq], agg[mt][4 * qq + 1], agg[mt][4 * qq + 2], agg[mt][4 * qq + 3]);
        } else if (hh == 0) {
            // compact [slot][4] triples inside the P region: H (v.C) must survive a coordinate pass (per-atom phases version 3:
            // behind the weight ring of stream_phase, whose loaders start as soon as the loop is over)
            *reinterpret_cast<float4*>(v.A + (v.misc[CX_V3] != 0 ? L_TRIP3 - L_A : 0) + 4 * slot) = make_float4(ax, ay, az, 0.0f);
        }
    }
}

// sum of the slot partials of every atom, chunks in ascending order (deterministic).  GCL: each thread owns up to four
// (atom, 4-feature) groups and returns them in registers (the destination v.C overlaps the partial buffer); max |agg|.
struct AggRegs {
    float4 v[4];
};
__device__ __forceinline__ float pair_reduce_gcl(const Lds& v, int nown, int nb, int tid, AggRegs& out, float scale) {
    const SlotPlan pl = slot_plan(nown, nb);
    float am = 0.0f;
    // the four groups of a thread are independent chains: their reads go out together, chunk by chunk (a group past the
    // end of the molecule reads the last atom's rows and is discarded)
    const float* src[4];
    float4 s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = tid + THREADS * k;
        src[k] = v.A + max(min(e >> 5, nown - 1), 0) * pl.g * PB_STRIDE + 4 * (e & 31);
        s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // (four chunks per round: their 16 reads are in flight together, the additions keep the chunk order - a team member has up
    // to 256 / n_own slots per atom, a chain of that many LDS latencies otherwise)
    int ch = 0;
    for (; ch + 4 <= pl.g; ch += 4) {
        float4 p[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) p[u][k] = *reinterpret_cast<const float4*>(src[k] + (ch + u) * PB_STRIDE);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[k].x += p[u][k].x; s[k].y += p[u][k].y; s[k].z += p[u][k].z; s[k].w += p[u][k].w; }
    }
    for (; ch < pl.g; ++ch) {
        float4 p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = *reinterpret_cast<const float4*>(src[k] + ch * PB_STRIDE);
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k].x += p[k].x; s[k].y += p[k].y; s[k].z += p[k].z; s[k].w += p[k].w; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = tid + THREADS * k;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < nown * 32) {
            r = make_float4(s[k].x * scale, s[k].y * scale, s[k].z * scale, s[k].w * scale);   // 1, or 1/N for aggregation_method='mean'
            am = fmaxf(fmaxf(am, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
        }
        out.v[k] = r;
    }
    return am;
}
// coordinate head: aggx[i][0..2] = sum of the slot triples of receiver i (thread = atom; atoms off the list keep whatever
// aggx holds: the update skips them)
template <bool V3>
__device__ __forceinline__ void pair_reduce_equiv(const Lds& v, int nown, int nb, int tid, float scale) {
    const SlotPlan pl = slot_plan(v.misc[MS_NRCV], nb);
    const int k = tid < nown ? v.rpos[tid] : -1;
    const float* trip = V3 ? v.A - L_A + L_TRIP3 : v.A;
    float* aggx = V3 ? v.A - L_A + L_AGGX3 : v.aggx;
    if (k >= 0) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int ch = 0; ch < pl.g; ++ch) {
            const float4 p = *reinterpret_cast<const float4*>(trip + 4 * (k * pl.g + ch));
            sx += p.x; sy += p.y; sz += p.z;
        }
        aggx[4 * tid + 0] = sx * scale; aggx[4 * tid + 1] = sy * scale; aggx[4 * tid + 2] = sz * scale;
    }
}


// ---------------------------------------------------------------------------------------------------
// Teams (version 2): S workgroups (S compute units) share one molecule - when the batch is smaller than the chip, or when
// the molecule has more atoms than one workgroup's LDS holds.  Atoms are dealt round-robin: member `rank` OWNS atoms rank,
// rank + S, ... (the linker atoms, which sit together at the end of a molecule, spread over all members).  A member keeps in
// LDS the state of its own atoms only - z, the h / aggregate / hidden fragment rows, the P rows - and does the per-atom
// phases (node MLP, projections, sampler algebra) for them alone; its pair loops take its own atoms as receivers and EVERY
// atom as sender, so it needs the sender rows Q and the coordinates of the whole molecule: once per pass (and never for the
// message sums, which stay where they are used) every member publishes the Q rows and coordinates of its atoms in a
// per-molecule HBM buffer and reads everybody's:  16-byte write-through (sc1) stores -> every storing wave drains -> workgroup
// barrier -> one lane publishes the exchange number in the member's arrival word (agent-scope relaxed store) -> one wave polls
// the S arrival words (relaxed, bounded) -> workgroup barrier -> sc1 loads.  Placement-independent (MI355X_MICROARCH.md,
// inter-workgroup visibility); two row buffers alternate, a member is at most one exchange ahead of the slowest one.
// With Q in the space of two row blocks a team holds molecules of up to NQMAX = 110 atoms.
// A member that gives up waiting (the launch did not get all its workgroups resident at once) publishes a POISON arrival word:
// every member that sees it stops waiting too, from then on nobody waits, and every member ends with flag bit 3 - the host
// re-runs that batch on a path that needs no co-residency.
constexpr int TEAM_MAX = 8;                                   // arrival words per molecule
constexpr int TX_ROW = LDH;                                   // exchange row: 128 floats of Q, then x, y, z, 0
constexpr int TEAM_X_BYTES = 2 * NQMAX * TX_ROW * 4;          // both parities of a molecule's exchange rows
constexpr int TEAM_HDR_BYTES = 2 * TEAM_MAX * 16;             // per parity and member: max |h| of its atoms (float bits), 3 spare words
constexpr int TEAM_MOL_BYTES = TEAM_X_BYTES + TEAM_HDR_BYTES;
constexpr unsigned TEAM_SPIN_LIMIT = 1u << 22;                // polls (~ seconds) before a member gives up
constexpr unsigned TEAM_POISON = 0xFFFFFFFFu;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long team_ptr(const Lds& v, int word) {
    const unsigned lo = unsigned(__builtin_amdgcn_readfirstlane(v.misc[word]));
    const unsigned hi = unsigned(__builtin_amdgcn_readfirstlane(v.misc[word + 1]));
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t team_rows(const Lds& v) {
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(team_ptr(v, TM_ROWS)), 0, TEAM_MOL_BYTES, 0x00020000);
}

// the rows of exchange number `epoch` are stored: publish, wait for the team (see above for what a timeout does)
__device__ __forceinline__ void team_sync(const Lds& v, int tid, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its write-through stores have left
    __syncthreads();
    if (tid < 64) {
        typedef __attribute__((address_space(1))) unsigned gu32;
        gu32* flags = reinterpret_cast<gu32*>(team_ptr(v, TM_FLAGS));
        const int S = v.misc[TM_S], rank = v.misc[TM_RANK];
        const unsigned target = epoch + 1u;
        bool failed = v.misc[TM_FAIL] != 0;
        if (tid == 0) __hip_atomic_store(flags + rank, failed ? TEAM_POISON : target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!failed) {
            bool ok = false;
            for (unsigned spins = 0; spins < TEAM_SPIN_LIMIT; ++spins) {
                unsigned f = target;
                if (tid < S) f = __hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__any(f == TEAM_POISON)) break;            // somebody gave up: so do we
                if (__all(int(f - target) >= 0)) { ok = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) {
                failed = true;
                if (tid == 0) __hip_atomic_store(flags + rank, TEAM_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid == 0) {
            v.misc[TM_EPOCH] = int(target);
            if (failed) v.misc[TM_FAIL] = 1;
        }
    }
    __syncthreads();
}

// ===================================================================================================
// Per-atom phases, version 2 (one workgroup per molecule).  Measured on round 2's build (profiles/r03): every per-atom phase
// of a GCL pass took 2-6 us for 0.3-1 us of matrix time - each of the four waves that share 32 atoms re-read and re-split the
// same rows inside its GEMM (24 VALU + an exposed LDS wait per 3 MFMAs), every phase ended in a workgroup-wide maximum, and
// the first node-MLP layer (K = 256) sat behind the pair loop.  Here:
//   * operands of the per-atom GEMMs live in LDS as FRAGMENT ROWS: a row of 128 values = 128 fp16 "hi" parts followed by
//     128 fp16 "lo" parts of value * s (512 bytes, the footprint of the fp32 row, same conflict-free row stride), written
//     ONCE by the phase that produces them; a GEMM is 16 ds_read_b128 + 24 MFMAs, no VALU (fp32 mode: rows stay fp32).
//   * the scales s are known BEFORE a phase runs: |P|, |Q|, the node MLP's hidden layer and the new h are bounded from the
//     measured max |h| (one workgroup maximum per pass, taken where h is written) and max |agg| (taken in the reduction,
//     whose barrier exists anyway) with the row-L1 norms of the packed matrices (host, dl_model_create).  A bound that is
//     loose by 2^k costs nothing until k ~ 13: fp16 has 30 binades + 10 subnormal bits and the split keeps hi AND lo.
//   * T0 = W3a' h + b3' - the half of the node MLP's first layer that does not depend on the messages - is computed BEFORE
//     the pair loop, beside P and Q (three independent MFMA chains per wave instead of two), parked in the workgroup's HBM
//     scratch in accumulator order and read back by the very lane that wrote it; after the loop only W3b' agg (K = 128) is left.
//   * weight fragments, T0 and the residual rows are requested a phase ahead, always in the order they are needed
//     (vector-memory returns in order: a prefetch issued BEFORE a load delays that load), biases ride in T0 / the residual.
// Per-workgroup HBM scratch `hs` (floats): h rows fp32 [NMAX][HID] (residual, output head), then T0 tiles [8][16][64].
constexpr int HS_HF = 0;                      // (teams, molecules of more than 55 atoms) the h fragment rows across a coordinate pass
constexpr int HS_T0 = NMAX * HID;
constexpr int HS_HT = HS_T0 + 8 * 16 * 64;    // the h rows once more, as accumulator tiles (the residual of the node MLP)
constexpr int HS_HO = HS_HT + 8 * 16 * 64;    // the final h for the output head: tiles [reg][lane] (32 consecutive features per row)
constexpr int HS_STRIDE = HS_HO + 8 * 16 * 64;
// T0 / HT tiles: element (reg, lane) of tile t at ((4 t + reg / 4) * 64 + lane) * 4 + reg % 4 - a lane moves its 16 accumulator
// registers with four 16-byte accesses, each a fully coalesced 1 KB per wave (16 dword accesses cost four times the issue slots of
// the vector-memory pipe, which is what the node phases wait for)
__device__ __forceinline__ void tile_store16(float* tiles, int t, int lane, const float (&x)[16]) {
    float4* p = reinterpret_cast<float4*>(tiles) + size_t(4 * t) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) p[64 * q] = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
}
__device__ __forceinline__ void tile_load16(const float* tiles, int t, int lane, float (&x)[16]) {
    const float4* p = reinterpret_cast<const float4*>(tiles) + size_t(4 * t) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t4 = p[64 * q];
        x[4 * q] = t4.x; x[4 * q + 1] = t4.y; x[4 * q + 2] = t4.z; x[4 * q + 3] = t4.w;
    }
}

constexpr int FS_HS = 7;                      // v.fmax slot: scale the h fragment rows in v.C were written with (float bits)
// host-computed bounds in the scale block of a GCL / equivariant update (pack_layout.h: G_SCALE / E_SCALE)
constexpr int SC_L1_W1A = 12, SC_L1_W1B = 13, SC_L1_W3A = 14, SC_L1_W3B = 15, SC_L1_W4 = 16, SC_B1 = 17, SC_B3 = 18, SC_B4 = 19;
constexpr int SCE_L1_W5A = 8, SCE_L1_W5B = 9, SCE_B5 = 10;

// ---- pass context.  Whatever a pass needs besides the constant LDS addresses - sizes, pointers, which pass this is - sits in
// v.misc[16..] and is read back (volatile, so never forwarded from an earlier read) at the point of use, before AND again
// after the pair loop: the loop takes every VGPR and most SGPRs, a value that lives across it goes to scratch, and each
// reload afterwards waits for ALL vector-memory traffic in flight (s_waitcnt vmcnt(0): the counter is in order) - measured
// round 3: ten such reloads in the 2 us reduction alone.
constexpr int CX_N = 16, CX_EM = 17, CX_HS = 19, CX_WP = 21, CX_PASS = 23, CX_PAR = 24, CX_FLAGS = 25, CX_NORMC = 26,
              CX_CRANGE = 27, CX_INVNORM = 28, CX_NPASS = 29, CX_NF = 30, CX_FIN = 31, CX_TFEAT = 32, CX_MOL = 33, CX_CTXP = 34,
              CX_SUB = 36, CX_CT = 37;      // inv_sublayers (GCLs per block), condition_time; [38] CX_V3 (above)
static_assert(CX_CT < CX_V3 && CX_V3 < MISC_WORDS, "context block");
__device__ __forceinline__ int ctx_i(const Lds& v, int k) {
    typedef volatile __attribute__((address_space(3))) int* lds_vint_t;       // a plain ds_read_b32 (a volatile GENERIC access is a flat sc0 sc1 load)
    return __builtin_amdgcn_readfirstlane(*(lds_vint_t)(v.misc + k));
}
__device__ __forceinline__ float ctx_f(const Lds& v, int k) { return __int_as_float(ctx_i(v, k)); }
template <class T>
__device__ __forceinline__ T* ctx_p(const Lds& v, int k) {
    const unsigned lo = unsigned(ctx_i(v, k)), hi = unsigned(ctx_i(v, k + 1));
    // (through the GLOBAL address space: an integer turned into a generic pointer makes every access a flat_load / flat_store)
    return (T*)((__attribute__((address_space(1))) T*)((unsigned long long)lo | ((unsigned long long)hi << 32)));
}
__device__ __forceinline__ void ctx_set_p(const Lds& v, int k, const void* ptr) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(ptr);
    v.misc[k] = int(unsigned(u)); v.misc[k + 1] = int(unsigned(u >> 32));
}
struct PassCtx {
    int nb, N, pass, par, flags;
    const float* g;             // this pass's packed weights
    float* hs;
    const int8_t* em;
    NextPass nx;
};
// a block = `sub` GCLs (inv_sublayers: 2 in every released configuration), then the equivariant update
__device__ __forceinline__ const float* pass_weights(const float* wp, int pass, int sub) {
    const int blk = pass / (sub + 1), k = pass - blk * (sub + 1);
    return wp + OFF_BLOCKS + size_t(blk) * block_size(sub) + k * GCL_SIZE;
}
__device__ __forceinline__ PassCtx pass_ctx(const Lds& v) {
    PassCtx c;
    c.nb = ctx_i(v, 0); c.N = ctx_i(v, CX_N); c.pass = ctx_i(v, CX_PASS); c.par = ctx_i(v, CX_PAR); c.flags = ctx_i(v, CX_FLAGS);
    const float* wp = ctx_p<const float>(v, CX_WP);
    const int sub = ctx_i(v, CX_SUB);
    c.g = pass_weights(wp, c.pass, sub);
    c.hs = ctx_p<float>(v, CX_HS);
    c.em = ctx_p<const int8_t>(v, CX_EM);
    const int nxt = c.pass + 1;
    c.nx.base = nxt < ctx_i(v, CX_NPASS) ? pass_weights(wp, nxt, sub) : nullptr;
    c.nx.equiv = (nxt % (sub + 1)) == sub;
    return c;
}
// the kernel's own arguments through the scalar cache, re-read where used (see above)
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* kargs() {
    auto k = (const __attribute__((address_space(4))) T*)(__builtin_amdgcn_kernarg_segment_ptr());
    asm volatile("" : "+s"(k));
    return k;
}

struct AReg {
    float4 q[16];           // f16x3: q[s] / q[8+s] = hi / lo fragment of k-slab s; fp32: q[sg] = k = 64 hh + 4 sg ..
};
template <int PREC>
__device__ __forceinline__ void load_a(AReg& a, const float* region, int arow, int hh) {
    const float4* p = reinterpret_cast<const float4*>(region + arow * LDH);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if constexpr (PREC == 0) { a.q[2 * s] = p[16 * hh + 2 * s]; a.q[2 * s + 1] = p[16 * hh + 2 * s + 1]; }
        else { a.q[s] = p[2 * s + hh]; a.q[8 + s] = p[16 + 2 * s + hh]; }
    }
}
template <int PREC>
__device__ __forceinline__ void tile_mma(floatx16& acc, const AReg& a, const BFrag& b) {
    if constexpr (PREC == 0) {
#pragma unroll
        for (int sg = 0; sg < 16; ++sg) {
            acc = mfma32(a.q[sg].x, b.q[sg].x, acc);
            acc = mfma32(a.q[sg].y, b.q[sg].y, acc);
            acc = mfma32(a.q[sg].z, b.q[sg].z, acc);
            acc = mfma32(a.q[sg].w, b.q[sg].w, acc);
        }
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const uint4 ah = __builtin_bit_cast(uint4, a.q[s]), al = __builtin_bit_cast(uint4, a.q[8 + s]);
            const uint4 bh = __builtin_bit_cast(uint4, b.q[s]), bl = __builtin_bit_cast(uint4, b.q[8 + s]);
            acc = mfma_h(al, bh, acc);
            acc = mfma_h(ah, bl, acc);
            acc = mfma_h(ah, bh, acc);
        }
    }
}
// one element of a fragment row (f16x3: hi truncated, lo = the exact remainder rounded) / of an fp32 row
template <int PREC>
__device__ __forceinline__ void put_elem(float* rowp, int k, float val, float s) {
    if constexpr (PREC == 0) rowp[k] = val;
    else {
        const float u = val * s;
        const float uh = __uint_as_float(__float_as_uint(u) & 0xffffe000u);
        _Float16* hp = reinterpret_cast<_Float16*>(rowp);
        hp[k] = static_cast<_Float16>(uh);
        hp[HID + k] = static_cast<_Float16>(u - uh);
    }
}
// four consecutive elements k0 .. k0+3 (k0 % 4 == 0)
template <int PREC>
__device__ __forceinline__ void put_quad(float* rowp, int k0, float4 val, float s) {
    if constexpr (PREC == 0) *reinterpret_cast<float4*>(rowp + k0) = val;
    else {
        const float u[4] = {val.x * s, val.y * s, val.z * s, val.w * s};
        float h[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = __uint_as_float(__float_as_uint(u[i]) & 0xffffe000u);
        const uint2 hi = make_uint2(__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h[0], h[1])),
                                    __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h[2], h[3])));
        const uint2 lo = make_uint2(__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(u[0] - h[0], u[1] - h[1])),
                                    __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(u[2] - h[2], u[3] - h[3])));
        *reinterpret_cast<uint2*>(reinterpret_cast<_Float16*>(rowp) + k0) = hi;
        *reinterpret_cast<uint2*>(reinterpret_cast<_Float16*>(rowp) + HID + k0) = lo;
    }
}

// fragments of the projections a pass opens with, requested one phase ahead: u0 = W1a' | W1b' (W5a' | W5b') by wave half,
// u1 = W3a' (GCL only)
struct PreW2 {
    BFrag u0, u1;
};
__device__ __forceinline__ void load_pre2_u0(PreW2& pw, const NextPass& nx, int w, int lane) {
    if (nx.base == nullptr) return;
    const int nt = w & 3;
    if (nx.equiv) pw.u0 = load_bfrag(nx.base + (w < 4 ? E_W5A : E_W5B) + nt * (UNIT / 4), lane);
    else pw.u0 = load_bfrag(nx.base + (w < 4 ? G_W1A : G_W1B) + nt * (UNIT / 4), lane);
}
// (T0 of atom tile 1 is the upper wave half's: with at most 32 own atoms - a team member, a small molecule - it does not exist)
__device__ __forceinline__ void load_pre2_u1(PreW2& pw, const NextPass& nx, int w, int lane, int nown) {
    if (nx.base == nullptr || nx.equiv || (w >= 4 && nown <= 32)) return;
    pw.u1 = load_bfrag(nx.base + G_W3A + (w & 3) * (UNIT / 4), lane);
}
__device__ __forceinline__ void load_pre2(PreW2& pw, const NextPass& nx, int w, int lane, int nown) {
    load_pre2_u0(pw, nx, w, lane);
    load_pre2_u1(pw, nx, w, lane, nown);
}

// P (waves 0-3) / Q (waves 4-7) of both atom tiles of the OWN atoms and, in a GCL, T0 of atom tile (wave half), from the h
// fragment rows in v.C.  A team's Q rows are stored without the geometric scale S1 (the consumer applies its own, see
// team_exchange_q); one workgroup per molecule stores them times S1 (see geo_scale).
template <int PREC, bool GCLP, bool TEAM>
__device__ __forceinline__ void pre_phase(const Lds& v, int nown, int w, int lane, const PreW2& pw, const float* __restrict__ vecs,
                                          const float* __restrict__ sc, float* __restrict__ hs, float s_hf) {
    const int c = lane & 31, hh = lane >> 5, nt = w & 3, half = w >> 2;
    float* dst = half ? v.B : v.A;
    const float bias = half ? 0.0f : vecs[32 * nt + c];                    // b1' (b5')
    const float bias3 = GCLP ? vecs[4 * HID + 32 * nt + c] : 0.0f;         // b3'
    float inv = 1.0f, inv3 = 1.0f;
    if (PREC != 0) {
        float S1 = 1.0f;                                                   // sender rows: times S1 (see geo_scale)
        if (half && !TEAM) {
            const float x2 = __uint_as_float(v.fmax[FM_X2]), x02 = __uint_as_float(v.fmax[FM_X02]);
            S1 = fminf(scale_for(4.0f * x2) * scale_for(cload(sc, 6)), scale_for(4.0f * x02) * scale_for(cload(sc, 7)));
        }
        // weight scales: one per 32-feature output tile (balanced packing, pack_layout.h)
        inv = inv_pow2(s_hf * cload(sc, (GCLP ? GS_SW_W1A : ES_SW_W5A) + 4 * half + nt)) * S1;
        if (GCLP) inv3 = inv_pow2(s_hf * cload(sc, GS_SW_W3A + nt));
    }
    const int mtiles = nown > 32 ? 2 : 1;
    AReg a;
    for (int mt = 0; mt < mtiles; ++mt) {
        load_a<PREC>(a, v.C, max(min(32 * mt + c, nown - 1), 0), hh);
        floatx16 acc = splat16(PREC == 0 ? bias : 0.0f);
        tile_mma<PREC>(acc, a, pw.u0);
        const bool t0 = GCLP && half == mt;
        floatx16 acc3;
        if (t0) {
            acc3 = splat16(PREC == 0 ? bias3 : 0.0f);
            tile_mma<PREC>(acc3, a, pw.u1);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = 32 * mt + acc_row(reg, hh);
            store_row(v, dst, row, nown, 32 * nt + c, (PREC == 0) ? acc[reg] : fmaf(acc[reg], inv, bias));
        }
        if (t0) {
            float t0v[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) t0v[reg] = (PREC == 0) ? acc3[reg] : fmaf(acc3[reg], inv3, bias3);
            tile_store16(hs + HS_T0, 4 * mt + nt, lane, t0v);
        }
    }
}

// Team: the sender rows of the own atoms (v.B rows 0 .. n_own-1, as pre_phase left them) and their coordinates -> exchange ->
// the sender row of EVERY atom in v.B (row = atom; through v.C when the molecule has more than 55), times the geometric
// scale S1 of the pass being opened (`sc`: its scale block), the coordinates of every atom in v.xs (and in v.x0 at the first
// exchange of a forward), max |x|^2 over the molecule in the OTHER bound slot (which becomes the current one), max |h| over
// the molecule in FM_HG.  Ends with every LDS write done but NOT yet fenced by a barrier (the caller's follows).
template <int PREC>
__device__ __forceinline__ void team_exchange_q(const Lds& v, int nb, int tid, const float* __restrict__ sc, bool first, int par, Prof& pf) {
    const int S = v.misc[TM_S], rank = v.misc[TM_RANK], nown = v.misc[TM_NOWN];
    const unsigned epoch = unsigned(v.misc[TM_EPOCH]);
    const __amdgpu_buffer_rsrc_t rows = team_rows(v);
    const int pbase = int(epoch & 1u) * (NQMAX * TX_ROW * 4);
    const int hbase = TEAM_X_BYTES + int(epoch & 1u) * (TEAM_MAX * 16);
    for (int e = tid; e < nown * 33; e += THREADS) {           // 32 x 16 bytes of Q and 16 bytes of coordinates per own atom
        const int l = e / 33, q4 = e - l * 33, a = rank + l * S;
        const float4 val = (q4 < 32) ? *reinterpret_cast<const float4*>(v.B + l * LDH + 4 * q4)
                                     : *reinterpret_cast<const float4*>(v.xs + 4 * a);
        const u32x4 bits = {__float_as_uint(val.x), __float_as_uint(val.y), __float_as_uint(val.z), __float_as_uint(val.w)};
        __builtin_amdgcn_raw_buffer_store_b128(bits, rows, pbase + (a * TX_ROW + 4 * q4) * 4, 0, 16);     // aux 16: sc1
    }
    if (tid == 0) {                                            // header: the bounds of the own atoms (complete since the last barrier)
        const u32x4 hdr = {v.fmax[FM_H0 + par], v.fmax[FM_XOWN], 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(hdr, rows, hbase + rank * 16, 0, 16);
    }
    prof_event(pf, tid >> 6, tid & 63, 111);
    team_sync(v, tid, epoch);              // (its first barrier also ends every read of the own rows in v.B)
    prof_event(pf, tid >> 6, tid & 63, 112);
    // ONE round trip: every wave reads the S headers (the bounds of the molecule = their maxima: no workgroup reduction, no
    // barrier), waves 0-1 the coordinates, everybody its share of the sender rows - all requested before anything is waited for
    // (every load unconditional, on a clamped index: a predicated load into a register array makes the compiler wait for each)
    const int ln = tid & 63;
    const u32x4 hdr = __builtin_amdgcn_raw_buffer_load_b128(rows, hbase + min(ln, S - 1) * 16, 0, 16);
    const u32x4 xb = __builtin_amdgcn_raw_buffer_load_b128(rows, pbase + (min(tid, nb - 1) * TX_ROW + HID) * 4, 0, 16);
    constexpr int NQ4 = 4;                 // 4 x 512 x 16 bytes: the sender rows of 64 atoms; the rest (bigger molecules) below
    u32x4 qv[NQ4];
#pragma unroll
    for (int k = 0; k < NQ4; ++k) {
        const int e = min(tid + THREADS * k, nb * 32 - 1);
        qv[k] = __builtin_amdgcn_raw_buffer_load_b128(rows, pbase + ((e >> 5) * TX_ROW + 4 * (e & 31)) * 4, 0, 16);
    }
    const unsigned hb = wave_max_u32(ln < S ? hdr.x : 0u), x2b = wave_max_u32(ln < S ? hdr.y : 0u);
    if (tid < nb) {
        const float4 x = make_float4(__uint_as_float(xb.x), __uint_as_float(xb.y), __uint_as_float(xb.z), 0.0f);
        *reinterpret_cast<float4*>(v.xs + 4 * tid) = x;
        if (first) *reinterpret_cast<float4*>(v.x0 + 4 * tid) = x;
    }
    if (tid == 0) {                        // readers: after the caller's barrier
        v.fmax[FM_X2] = x2b; v.fmax[FM_HG] = hb;
        if (first) v.fmax[FM_X02] = x2b;
    }
    float S1 = 1.0f;
    if (PREC != 0) {
        const float x2 = __uint_as_float(x2b), x02 = first ? x2 : __uint_as_float(v.fmax[FM_X02]);
        S1 = fminf(scale_for(4.0f * x2) * scale_for(cload(sc, 6)), scale_for(4.0f * x02) * scale_for(cload(sc, 7)));
    }
#pragma unroll
    for (int k = 0; k < NQ4; ++k) {
        const int e = tid + THREADS * k;
        if (e < nb * 32)
            *reinterpret_cast<float4*>(v.B + (e >> 5) * LDH + 4 * (e & 31)) =
                make_float4(__uint_as_float(qv[k].x) * S1, __uint_as_float(qv[k].y) * S1, __uint_as_float(qv[k].z) * S1, __uint_as_float(qv[k].w) * S1);
    }
    for (int e = tid + THREADS * NQ4; e < nb * 32; e += THREADS) {
        const u32x4 bits = __builtin_amdgcn_raw_buffer_load_b128(rows, pbase + ((e >> 5) * TX_ROW + 4 * (e & 31)) * 4, 0, 16);
        *reinterpret_cast<float4*>(v.B + (e >> 5) * LDH + 4 * (e & 31)) =
            make_float4(__uint_as_float(bits.x) * S1, __uint_as_float(bits.y) * S1, __uint_as_float(bits.z) * S1, __uint_as_float(bits.w) * S1);
    }
}

// the projections (and T0) the NEXT pass opens with, from the h fragment rows in v.C; ends with every LDS operand of that
// pass's pair loop in place.  Fragments are requested and used inside one straight-line region: nothing of them is
// carried around the pass loop (128 registers that the pair loop's 256 would push to scratch).
template <int PREC, bool TEAM>
__device__ __forceinline__ void open_pass(const Lds& v, int nb, int nown, const NextPass nx, float* __restrict__ hs, Prof& pf,
                                          const PreW2& pw, int next_pass, int par) {
    if (nx.base == nullptr) return;
    const LaneIds q = lane_ids(v);
    const int w = q.w, lane = q.lane;
    const float s_hf = (PREC != 0) ? __uint_as_float(v.fmax[FS_HS]) : 1.0f;
    const float* sc = nx.base + (nx.equiv ? E_SCALE : G_SCALE);
    if (nx.equiv) pre_phase<PREC, false, TEAM>(v, nown, w, lane, pw, nx.base + E_VEC, sc, nullptr, s_hf);
    else pre_phase<PREC, true, TEAM>(v, nown, w, lane, pw, nx.base + G_VEC, sc, hs, s_hf);
    if constexpr (TEAM) {
        prof_event(pf, w, lane, 110);
        lds_barrier();                     // own Q rows complete
        team_exchange_q<PREC>(v, nb, q.tid, sc, next_pass == 0, par, pf);
    }
    if (q.tid == 0) v.misc[CX_PASS] = next_pass;
    prof_event(pf, w, lane, 11);
    dma_wait();
    lds_barrier();                         // P, Q, W2', vectors, coordinates in place; every read of the h rows (v.C) done
}

// GCL (egnn.py:45-80), per-atom phases version 2; P, Q, T0 of this pass are in place (open_pass).  Ends with the next pass opened.
template <int PREC, bool TEAM, bool ATT>
__device__ __forceinline__ void gcl_pass2(const Lds& v, Prof& pf) {
    {   // ---- pair loop
    const PassCtx cx = pass_ctx(v);
    const int nb = cx.nb, N = cx.N, par = cx.par;
    const int8_t* emask = cx.em;
    const float* sc = cx.g + G_SCALE;
    const LaneIds q = lane_ids(v);
    const int tid = q.tid, w = q.w, lane = q.lane;
    prof_event(pf, w, lane, 12);
    if (tid == 0) { v.fmax[FM_H0 + (par ^ 1)] = 0u; v.fmax[FM_AGG] = 0u; }      // (max |h| is kept in every arithmetic mode: equiv_pass2's finiteness proof)
    float sa = 1.0f, accs = 1.0f;
    if (PREC != 0) {
        const float hmax = __uint_as_float(v.fmax[TEAM ? FM_HG : FM_H0 + par]);     // senders: any atom of the molecule
        const float x2 = __uint_as_float(v.fmax[FM_X2]), x02 = __uint_as_float(v.fmax[FM_X02]);
        const float pqb = (cload(sc, SC_L1_W1A) + cload(sc, SC_L1_W1B)) * hmax + cload(sc, SC_B1);            // >= |P_i| + |Q_j|
        const float u1b = pqb + 4.0f * (x2 * cload(sc, GS_WRW) + x02 * cload(sc, GS_WDW));        // (bounds with the slab exponents applied)
        sa = scale_for(u1b);
        accs = sa * cload(sc, 5);
        // (h rows, hidden activations, r and d0 - geo_scale - of this pass; pack_layout.h: beyond_f16_range)
        if (tid == 0 && (beyond_f16_range(u1b) || beyond_f16_range(hmax) || beyond_f16_range(4.0f * x2) || beyond_f16_range(4.0f * x02)))
            atomicOr(&v.misc[1], NAN_RANGE | 3);
    }
    // (edge attention is a KERNEL variant, not a branch: its pair loop keeps the 64 messages of a step until the logit is known
    // and spills ~80 registers, which would set the scratch size - and the register pressure around the call - of every launch)
    if constexpr (ATT) pair_phase<false, PREC, true, TEAM>(v, nb, w, lane, emask, N, 0.0f, sa, inv_pow2(accs), sc, cload(sc, 8), pf);
    else pair_phase<false, PREC, false, TEAM>(v, nb, w, lane, emask, N, 0.0f, sa, inv_pow2(accs), sc, 0.0f, pf);
    prof_event(pf, w, lane, 13);
    }
    // ---- aggregate, node MLP of the own atoms (context and lane indices re-derived: nothing lives across the loop)
    const PassCtx cx = pass_ctx(v);
    const int nb = cx.nb, N = cx.N, par = cx.par;
    const int nown = TEAM ? ctx_i(v, TM_NOWN) : nb;
    const float* g = cx.g;
    float* hs = cx.hs;
    const NextPass nx = cx.nx;
    const float* vecs = g + G_VEC;
    const float* sc = g + G_SCALE;
    const bool mean = (cx.flags & 4) != 0;
    const LaneIds q = lane_ids(v);
    const int tid = q.tid, w = q.w, lane = q.lane, c = q.c, hh = q.hh, nt = q.nt, mt = q.mt;
    const bool active = (mt == 0) || (nown > 32);
    // layer 1's fragments and T0, AHEAD of the next W2' image in the memory pipeline
    BFrag b3b, b4f;
    float t0r[16], hold[16];
    float b4 = 0.0f;
    if (active) {
        b3b = load_bfrag(g + G_W3B + nt * (UNIT / 4), lane);
        tile_load16(hs + HS_T0, 4 * mt + nt, lane, t0r);
    }
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();                         // partial rows complete
    prof_event(pf, w, lane, 20);
    AggRegs ar;
    const float am = pair_reduce_gcl(v, nown, nb, tid, ar, mean ? 1.0f / float(N) : 1.0f);
    if (PREC != 0) block_max(&v.fmax[FM_AGG], am, lane);
    prof_event(pf, w, lane, 21);
    lds_barrier();                         // every partial read: P, Q, H, W2' regions are free; max |agg| known
    prof_event(pf, w, lane, 22);
    float hmax = 0.0f, aggmax = 0.0f, s_agg = 1.0f;
    if (PREC != 0) {
        hmax = __uint_as_float(v.fmax[FM_H0 + par]);
        aggmax = __uint_as_float(v.fmax[FM_AGG]);
        s_agg = scale_for(aggmax);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {          // aggregate -> fragment rows in v.C
        const int e = tid + THREADS * k;
        if (e < nown * 32) put_quad<PREC>(v.C + (e >> 5) * LDH, 4 * (e & 31), ar.v[k], s_agg);
    }
    // bounds: |y3| <= L1(W3a') max|h| + L1(W3b') max|agg| + max|b3'|  >= |t| ;  |h_new| <= max|h| + L1(W4') |t| + max|b4|
    const float y3b = cload(sc, SC_L1_W3A) * hmax + cload(sc, SC_L1_W3B) * aggmax + cload(sc, SC_B3);
    const float s_t = (PREC != 0) ? scale_for(y3b) : 1.0f;
    const float hnb = hmax + cload(sc, SC_L1_W4) * y3b + cload(sc, SC_B4);
    const float s_hn = (PREC != 0) ? scale_for(hnb) : 1.0f;
    if (PREC != 0 && tid == 0 && (beyond_f16_range(aggmax) || beyond_f16_range(y3b) || beyond_f16_range(hnb))) atomicOr(&v.misc[1], NAN_RANGE | 3);
    // (W3b': one weight scale per output tile; the hidden layer t of tile nt is written times s_t * 2^n_nt, W4' carries 2^-n per column)
    const float inv1 = (PREC != 0) ? inv_pow2(s_agg * cload(sc, GS_SW_W3B + nt)) : 1.0f, inv2 = (PREC != 0) ? inv_pow2(s_t * cload(sc, 4)) : 1.0f;
    const float s_tn = (PREC != 0) ? s_t * cload(sc, GS_NT + nt) : 1.0f;
    prof_event(pf, w, lane, 23);
    lds_barrier();
    prof_event(pf, w, lane, 14);
    // node MLP layer 1: t = SiLU(T0 + W3b' agg)  -> fragment rows in v.B
    const int arow = max(min(32 * mt + c, nown - 1), 0);
    if (active) {
        AReg a;
        load_a<PREC>(a, v.C, arow, hh);
        // for layer 2, under layer 1: its fragments, the residual tile (written by this lane, a pass ago), the bias
        b4f = load_bfrag(g + G_W4 + nt * (UNIT / 4), lane);
        tile_load16(hs + HS_HT, 4 * mt + nt, lane, hold);
        b4 = vecs[5 * HID + 32 * nt + c];
        __builtin_amdgcn_sched_barrier(0);
        floatx16 acc;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc[reg] = (PREC == 0) ? t0r[reg] : 0.0f;
        tile_mma<PREC>(acc, a, b3b);
        const float inv = inv1;
        prof_event(pf, w, lane, 105);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = 32 * mt + acc_row(reg, hh);
            const float tval = silu_u(PREC == 0 ? acc[reg] : fmaf(acc[reg], inv, t0r[reg]));
            put_elem<PREC>(row < nown ? v.B + row * LDH : v.dummy, 32 * nt + c, tval, s_tn);
        }
    }
    prof_event(pf, w, lane, 15);
    lds_barrier();
    // node MLP layer 2 + residual: new h -> fragment rows in v.C, fp32 tiles in the HBM scratch
    PreW2 pw;
    floatx16 acc2;
    if (active) {
        AReg a;
        load_a<PREC>(a, v.B, arow, hh);
        // the residual (+ bias) is taken into registers HERE: once the prefetches below are in flight a wait for it would be a
        // wait for all of them (on the path of the last pass, which issues none, the compiler's one s_waitcnt has to be 0)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) { hold[reg] += b4; acc2[reg] = (PREC == 0) ? hold[reg] : 0.0f; }
        asm volatile("" : "+v"(hold[0]), "+v"(hold[1]), "+v"(hold[2]), "+v"(hold[3]), "+v"(hold[4]), "+v"(hold[5]), "+v"(hold[6]), "+v"(hold[7]));
        asm volatile("" : "+v"(hold[8]), "+v"(hold[9]), "+v"(hold[10]), "+v"(hold[11]), "+v"(hold[12]), "+v"(hold[13]), "+v"(hold[14]), "+v"(hold[15]));
        prof_event(pf, w, lane, 106);
        tile_mma<PREC>(acc2, a, b4f);
        prof_event(pf, w, lane, 107);
    }
    __builtin_amdgcn_sched_barrier(0);
    // behind layer 2's matrix instructions (their operands have been read): the fragments the next pass opens with, then the
    // next pass's W2' image - requested in the order of use, landing under the epilogue and the barrier
    load_pre2(pw, nx, w, lane, nown);
    stage_next(v, nx, w, tid);
    __builtin_amdgcn_sched_barrier(0);
    if (active) {
        floatx16& acc = acc2;
        const float inv = inv2;
        float hm = 0.0f;
        float hnew[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = 32 * mt + acc_row(reg, hh);
            const float hv = (PREC == 0) ? acc[reg] : fmaf(acc[reg], inv, hold[reg]);
            put_elem<PREC>(row < nown ? v.C + row * LDH : v.dummy, 32 * nt + c, hv, s_hn);
            hnew[reg] = hv;
            hm = fmaxf(hm, row < nown ? fabsf(hv) : 0.0f);
        }
        tile_store16(hs + HS_HT, 4 * mt + nt, lane, hnew);
        if (cx.pass + 2 >= ctx_i(v, CX_NPASS)) {                     // the last GCL of the forward: the copy the output head reads
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) hs[HS_HO + ((4 * mt + nt) * 16 + reg) * 64 + lane] = hnew[reg];
        }
        block_max(&v.fmax[FM_H0 + (par ^ 1)], hm, lane);
    }
    if (tid == 0) {
        if (PREC != 0) v.fmax[FS_HS] = __float_as_uint(s_hn);
        v.misc[CX_PAR] = par ^ 1;
    }
    prof_event(pf, w, lane, 16);
    lds_barrier();
    if constexpr (TEAM) {
        // a molecule of more than 55 atoms: the sender rows of the coming coordinate pass reach into v.C - the h fragment rows
        // go to the HBM scratch and come back after that pair loop (equiv_pass2)
        if (nx.equiv && nb > NMAX)
            for (int e = tid; e < nown * 32; e += THREADS)
                *reinterpret_cast<float4*>(hs + HS_HF + (e >> 5) * HID + 4 * (e & 31)) = *reinterpret_cast<const float4*>(v.C + (e >> 5) * LDH + 4 * (e & 31));
    }
    prof_event(pf, w, lane, 10);
    open_pass<PREC, TEAM>(v, nb, nown, nx, hs, pf, pw, cx.pass + 1, par ^ 1);
}

// every own atom becomes a receiver of the coordinate pass (equiv_pass2: a skipped sum could not be proven finite); the caller's
// barrier follows
__device__ __forceinline__ void receivers_all(const Lds& v, int nown, int tid) {
    if (tid < NMAX + 1) {
        if (tid < nown) v.rcv[tid] = tid;
        v.rpos[tid] = tid < nown ? tid : -1;
    }
    if (tid == 0) { v.misc[MS_NRCV] = nown; v.misc[MS_FULL] = 1; }
}

// EquivariantUpdate (egnn.py:101-125), per-atom phases version 2: the h fragment rows in v.C survive the pass; P, Q of
// this pass are in place (open_pass); ends with the next pass opened.
template <int PREC, bool TEAM>
__device__ __forceinline__ void equiv_pass2(const Lds& v, Prof& pf) {
    {   // ---- pair loop
    const PassCtx cx = pass_ctx(v);
    const int nb = cx.nb, N = cx.N, par = cx.par;
    const int8_t* emask = cx.em;
    const float* sc = cx.g + E_SCALE;
    const float norm_constant = ctx_f(v, CX_NORMC);
    const LaneIds q = lane_ids(v);
    const int w = q.w, lane = q.lane;
    prof_event(pf, w, lane, 32);
    float sa = 1.0f, accs = 1.0f;
    {
        const float hmax = __uint_as_float(v.fmax[TEAM ? FM_HG : FM_H0 + par]);
        const float x2 = __uint_as_float(v.fmax[FM_X2]), x02 = __uint_as_float(v.fmax[FM_X02]);
        const float pqb = (cload(sc, SCE_L1_W5A) + cload(sc, SCE_L1_W5B)) * hmax + cload(sc, SCE_B5);
        const float u1b = pqb + 4.0f * (x2 * cload(sc, ES_WRW) + x02 * cload(sc, ES_WDW));        // >= every hidden activation of this pass
        if (PREC != 0) {
            sa = scale_for(u1b);
            accs = sa * cload(sc, 2);
            if (q.tid == 0 && (beyond_f16_range(u1b) || beyond_f16_range(hmax) || beyond_f16_range(4.0f * x2) || beyond_f16_range(4.0f * x02)))
                atomicOr(&v.misc[1], NAN_RANGE | 3);
        }
        // The pass runs for the receivers inside the linker mask only: the reference multiplies every other atom's sum by zero
        // (egnn.py:113-116).  That is the reference's result exactly as long as the skipped sums are FINITE - an inf or NaN there
        // becomes NaN * 0 = NaN in the reference's coordinates (and FoundNaNException).  The bound of the head's output proves it:
        // |trans| <= |coord_diff| |w7' . u2| <= phi, at most nb terms of mask weight <= 2 per sum.  Where it does not (an
        // overflowing head, non-finite features: the comparison is false for inf and NaN), the receiver list becomes every own
        // atom - for the rest of the launch - and the sums are formed and multiplied by the mask as the reference does.
        const float phi = cload(sc, ES_W7L1) * fmaf(cload(sc, ES_L1_W6), u1b, cload(sc, ES_B6));
        const bool proven = 2.0f * float(nb) * phi < 1e37f;
        // (`proven` is workgroup-uniform: every input is a scalar of the packed model or an LDS bound that the barrier ending
        // open_pass published.  MS_FULL is WRITTEN inside the branch below, so every wave reads it before anybody may write it -
        // a wave that saw the new value would skip the branch and its barrier, and the pair loop would start on a half-built
        // receiver list: ADVICE round 5.  The extra barrier exists on the unproven path only)
        if (!proven) {
            const bool widen = ctx_i(v, MS_FULL) == 0;
            __syncthreads();
            if (widen) {
                const int tid_ = q.tid;
                receivers_all(v, TEAM ? ctx_i(v, TM_NOWN) : nb, tid_);
                __syncthreads();
            }
        }
    }
    pair_phase<true, PREC, false, TEAM>(v, nb, w, lane, emask, N, norm_constant, sa, inv_pow2(accs), sc,
                                        (cx.flags & 2) ? ctx_f(v, CX_CRANGE) : 0.0f, pf);      // ends with the partial triples in LDS
    prof_event(pf, w, lane, 33);
    }
    // ---- coordinate update of the own atoms (context and lane indices re-derived: nothing lives across the loop)
    const PassCtx cx = pass_ctx(v);
    const int nb = cx.nb, N = cx.N;
    const int nown = TEAM ? ctx_i(v, TM_NOWN) : nb;
    const int S = TEAM ? ctx_i(v, TM_S) : 1, rank = TEAM ? ctx_i(v, TM_RANK) : 0;
    float* hs = cx.hs;
    const NextPass nx = cx.nx;
    const LaneIds q = lane_ids(v);
    const int tid = q.tid, w = q.w, lane = q.lane;
    PreW2 pw;
    load_pre2(pw, nx, w, lane, nown);            // the fragments the next block opens with, under the reduction
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();                         // partial triples complete
    // w7' carries 1/normalization_factor unless tanh or the mean need the raw head output (dl_model_create)
    const float xscale = (cx.flags & 4) ? 1.0f / float(N) : ((cx.flags & 2) ? ctx_f(v, CX_INVNORM) : 1.0f);
    pair_reduce_equiv<false>(v, nown, nb, tid, xscale);
    if (TEAM && tid == 0) v.fmax[FM_XOWN] = 0u;
    lds_barrier();                         // partials read: P, Q, W2' regions are free
    stage_next(v, nx, w, tid);
    if constexpr (TEAM) {
        if (nb > NMAX && nx.base != nullptr)                           // the h fragment rows back into v.C (see gcl_pass2)
            for (int e = tid; e < nown * 32; e += THREADS)
                *reinterpret_cast<float4*>(v.C + (e >> 5) * LDH + 4 * (e & 31)) = *reinterpret_cast<const float4*>(hs + HS_HF + (e >> 5) * HID + 4 * (e & 31));
    } else {
        if (tid == 0) v.fmax[FM_X2] = 0u;
        lds_barrier();
    }
    float n2 = 0.0f;
    if (tid < nown) {
        const float lm = v.lm[tid];
        const int a = rank + tid * S;
        const bool moves = v.rpos[tid] >= 0;                     // off the receiver list: linker mask 0, nothing was summed
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float xn = v.xs[4 * a + k];
            if (moves) {
                xn += v.aggx[4 * tid + k] * lm;
                v.xs[4 * a + k] = xn;
            }
            n2 = fmaf(xn, xn, n2);
        }
    }
    block_max(&v.fmax[TEAM ? FM_XOWN : FM_X2], n2, lane);       // (a team: the own atoms; the exchange headers carry it)
    prof_event(pf, w, lane, 34);
    lds_barrier();
    prof_event(pf, w, lane, 10);
    open_pass<PREC, TEAM>(v, nb, nown, nx, hs, pf, pw, cx.pass + 1, cx.par);
}


// ---------------------------------------------------------------------------------------------------
// Per-atom phases, version 3 (round 6, f16 modes): ATOM-STATIONARY and TRANSPOSED, like the pair loop.
//   Version 2 (above; still the exact-fp32 mode's) runs every per-atom GEMM with the atoms as MFMA rows on all eight waves: each
//   GEMM reads the activations back from LDS fragment rows, its result goes through 32 two-byte LDS stores per lane into the
//   next GEMM's rows, seven workgroup barriers per pass, and every wave pulls 16 KB of weight fragments per GEMM through the
//   vector-memory path - ~830 KB per pass and compute unit at 64 B/clk, 15-20 us per GCL pass of which ~3.5 us is matrix work
//   (profiles/r03/phase_timeline_B64_team1.log: 20 % of a forward at B = 64, 25 % at B = 256).
//   Here a wave OWNS 16 atoms (waves 0-3: own atoms 16 w .. 16 w + 15; a 16-atom tile = the N dimension of
//   v_mfma_f32_16x16x32_f16) and computes D = W' x X with the output FEATURES as MFMA rows: the accumulator layout - lane
//   (atom, kg) holds features 16 ot + 4 kg + 0..3 of tile ot - IS the B-operand layout of the next GEMM once the k-slots of its
//   weights are permuted to match (pack_layout.h: stream_kslot), so   agg -> t = SiLU(T0 + W3b' agg) -> h += W4' t + b4 ->
//   P, Q, T0 of the next pass   is one register-to-register chain per wave: no LDS round trip, no barrier, no layout conversion
//   between the GEMMs.  The weights - the same 64 KB for every wave - are STREAMED through LDS: waves 4-7 do nothing but issue
//   LDS-DMA (global_load_lds_dwordx4) of 32 KB chunks into a ring of four slots (the P, Q, h and W2' regions: dead between two
//   pair loops), two chunks ahead of the one being consumed; a chunk boundary is one workgroup barrier.  HBM scratch: the fp32
//   node features (residual) and T0, 4 x float4 per lane and tile in accumulator order, read by the lane that wrote them.
// ---------------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ floatx4 mfma16(const uint4& a, const uint4& b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
constexpr int ST_AWAVES = 4;                       // waves 0-3: 16 atoms each; waves 4-7: the loaders
constexpr int ST_PIECES = ST_CHUNK / 256 / 4;      // 1 KB LDS-DMA pieces per loader wave and chunk: 8
// ring slots (floats from the LDS base): 0, 1 = the two halves of the W2' image region, 2, 3 = the first 64 KB of the P / Q / h regions
__host__ __device__ constexpr int st_slot_off(int slot) { return slot < 2 ? L_W + slot * ST_CHUNK : L_A + (slot - 2) * ST_CHUNK; }
static_assert(L_A + 2 * ST_CHUNK <= L_W && 2 * ST_CHUNK == UNIT, "ring slots 2, 3 must fit below the W2' region");
// chunk c of a stream of NC chunks sits in slot (c + off) & 3 with off chosen so that the LAST two chunks use slots 2, 3: the
// W2' image of the next pair loop (slots 0, 1) is requested two chunks before the stream ends and lands under them
__host__ __device__ constexpr int st_slot(int c, int nc) { return (c + ((4 - nc) & 3)) & 3; }

struct BOp {
    uint4 hi[4], lo[4];                            // k-slabs 0..3 of a wave's 16 atoms: fp16 hi / lo fragments
};
// tile (w, ot) of the HBM scratch: [lane] float4 = features 16 ot + 4 kg + 0..3 of atom 16 w + n, lane = 16 kg + n
__device__ __forceinline__ float4* st_tile(float* tiles, int w, int ot, int lane) {
    return reinterpret_cast<float4*>(tiles) + (w * 8 + ot) * 64 + lane;
}
// fp32 rows [atom][LDH] in LDS -> the B operand of a wave's 16 atoms (rows past the last own atom: the last one's, discarded later)
__device__ __forceinline__ void st_load_rows(BOp& b, const float* rows, int l, int kg, float s) {
    const float* rp = rows + l * LDH + 4 * kg;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        const float4 a = *reinterpret_cast<const float4*>(rp + 32 * sl), c4 = *reinterpret_cast<const float4*>(rp + 32 * sl + 16);
        const float u[8] = {a.x * s, a.y * s, a.z * s, a.w * s, c4.x * s, c4.y * s, c4.z * s, c4.w * s};
        split8t(u, b.hi[sl], b.lo[sl]);
    }
}
// the four output tiles of one chunk: acc[t] = W'[tile t of the chunk] x B, three split terms, smallest first; tile 0, 1, 2, 3 for
// each term: an accumulator is touched every fourth matrix instruction.
// (ordered by the compiler: a hand-pipelined variant - fragments of k-slab s + 1 requested before the MFMAs of slab s,
// sched_barrier after every MFMA - keeps a clean 0 1 2 3 round robin where the scheduler walks the tiles back and forth, but its
// 64 fragment registers push ~50 others of the kernel into scratch and measured 1 % WORSE, round 6)
template <int SLOT>
__device__ __forceinline__ void st_mma_chunk(const float* lds0, const BOp& b, floatx4 (&acc)[4], int lane) {
    const uint4* Wc = reinterpret_cast<const uint4*>(lds0 + st_slot_off(SLOT)) + lane;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        uint4 ah[4], al[4];
#ifdef DL_KO_DSR
#pragma unroll
        for (int t = 0; t < 4; ++t) { ah[t] = b.hi[(sl + t) & 3]; al[t] = b.lo[(sl + t) & 3]; }
#else
#pragma unroll
        for (int t = 0; t < 4; ++t) { ah[t] = Wc[((t * 4 + sl) * 2 + 0) * 64]; al[t] = Wc[((t * 4 + sl) * 2 + 1) * 64]; }
#endif
#ifdef DL_KO_MFMA
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t][0] += __uint_as_float(al[t].x ^ ah[t].y); acc[t][1] += __uint_as_float(al[t].z ^ ah[t].w); }
#else
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(al[t], b.hi[sl], acc[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(ah[t], b.lo[sl], acc[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(ah[t], b.hi[sl], acc[t]);
#endif
    }
}
// loader waves: chunk `src` (32 KB, global) -> ring slot SLOT, 8 pieces of 1 KB per wave
template <int SLOT>
__device__ __forceinline__ void st_issue(float* lds0, const float* __restrict__ src, int hw, int lane) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float4* s4 = reinterpret_cast<const float4*>(src) + lane;
#ifdef DL_KO_DMA
    (void)s4; (void)lds0; (void)hw;
#else
#pragma unroll
    for (int it = 0; it < ST_PIECES; ++it) {
        const int piece = it * 4 + hw;
        __builtin_amdgcn_global_load_lds((gptr_t)(s4 + piece * 64), (lptr_t)(lds0 + st_slot_off(SLOT) + piece * 256), 16, 0, 0);
    }
#endif
}
// loader waves: the W2' / W6' image and the four vectors of the next pair loop (stage_dma's job, on four waves)
template <int HALF>
__device__ __forceinline__ void st_issue_image(const Lds& v, const NextPass& nx, int hw, int lane) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float* wimg = nx.base + (nx.equiv ? E_W6T : G_W2T);
    const float* vecs = nx.base + (nx.equiv ? E_VEC : G_VEC) + HID;
    const float* vec4 = nx.base + (nx.equiv ? E_VEC + 4 * HID : G_VEC + 6 * HID);
    if (HALF == 1 && hw < 2) {                     // lanes 0..95 of the pair: wr', wd', b2'|b6'; 96..127: w7' / w_att'
        const int t2 = 64 * hw + lane;
        const float4* src = (t2 < 96) ? reinterpret_cast<const float4*>(vecs) + t2 : reinterpret_cast<const float4*>(vec4) + (t2 - 96);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(v.vec + 256 * hw), 16, 0, 0);
    }
    const float4* s4 = reinterpret_cast<const float4*>(wimg) + lane;
#pragma unroll
    for (int it = 0; it < ST_PIECES; ++it) {       // 32 KB = the ring slot this half of the image region was
        const int piece = HALF * 32 + it * 4 + hw;
        __builtin_amdgcn_global_load_lds((gptr_t)(s4 + piece * 64), (lptr_t)(v.W + piece * 256), 16, 0, 0);
    }
}
template <int N>
__device__ __forceinline__ void st_wait_vm() {
    if constexpr (N <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
template <int V>
struct IC { static constexpr int value = V; };

// POST: after a GCL's pair loop (slot partials summed: `ar`, max |agg| in FM_AGG) - node MLP, then the next pass's projections;
// !POST: the projections alone, from the node features in the HBM scratch (forward entry, after a coordinate pass).
// NEXT_EQ: the next pass is the coordinate pass (P, Q from W5a', W5b') else a GCL (P, Q, T0 from W1a', W1b', W3a').
// Ends like open_pass: every LDS operand of the next pair loop in place, behind a barrier.
template <bool TEAM, bool POST, bool NEXT_EQ>
__device__ __forceinline__ void stream_phase(const Lds& v, Prof& pf, const AggRegs* ar) {
    // chunks of the stream.  POST: [W3a' tiles 0-3][W3b' tiles 0-3][W3a' 4-7][W3b' 4-7] (node MLP layer 1: the half that reads h
    // stays in registers for one chunk - until round 6 it was computed a pass earlier and parked in the HBM scratch as "T0"),
    // [W4' 0-3][W4' 4-7]; then the next pass's P and Q, two chunks each
    constexpr int NC = (POST ? 6 : 0) + 4;
    constexpr int C_P = POST ? 6 : 0;              // first chunk of the P unit; Q follows
    const PassCtx cx = pass_ctx(v);
    const LaneIds q = lane_ids(v);
    const int tid = q.tid, w = q.w, lane = q.lane;
    const int nb = cx.nb;
    const int nown = TEAM ? ctx_i(v, TM_NOWN) : nb;
    float* hs = cx.hs;
    // the pass whose operands this phase prepares: the one after the current (forward entry: CX_PASS = -1)
    const int pass_open = cx.pass + 1;
    const float* wp = ctx_p<const float>(v, CX_WP);
    const int sub = ctx_i(v, CX_SUB);
    const float* g = POST ? cx.g : nullptr;                              // this GCL (POST)
    const float* nxb = pass_weights(wp, pass_open, sub);                 // the pass being opened
    const NextPass nx = {nxb, NEXT_EQ};
    const float* sc = POST ? g + G_SCALE : nullptr;
    const float* scn = nxb + (NEXT_EQ ? E_SCALE : G_SCALE);
    const float* vecn = nxb + (NEXT_EQ ? E_VEC : G_VEC);
    float* lds0 = v.A - L_A;
    // (atom waves 0-3 sit on the four SIMDs, one each, beside a loader: waves w and w ^ 4 share a SIMD - with atom waves 0, 2, 4, 6
    // two of them share one matrix pipe and the phase takes 40.7 K cycles instead of 31.7 K, profiles/r06/ab_atom_wave_mapping.log)
    const bool loader = w >= ST_AWAVES;
    const int hw = w - ST_AWAVES, wa = w;
    const int n = lane & 15, kg = lane >> 4;
    // (tried, round 6: a workgroup with one or two 16-atom tiles - a team member, a small molecule - sharing each tile between 4 / 2
    // waves that split the output tiles of a chunk and hand the results round through LDS: a quarter / half of the matrix
    // instructions per wave, measured NEUTRAL - with so little matrix work a step is a chain of latencies (barrier, fragment
    // reads, a dependent MFMA chain, the hand-over), 1.4 K cycles whatever its size.  Such workgroups keep version 2: see
    // forward_molecule2)
    const int ta = wa;                                                   // this wave's 16-atom tile
    const int l = 16 * ta + n;                                           // own atom of this lane (atom waves)
    const bool awave = !loader && 16 * ta < nown;                        // wave-uniform
    const bool valid = awave && l < nown;
    const int lc = max(min(l, nown - 1), 0);
    auto chunk_src = [&](int c) -> const float* {                        // chunk c of this phase's stream (global)
        if (POST && c < C_P) return g + G_ST_POST + c * ST_CHUNK;
        return nxb + (NEXT_EQ ? E_ST_PRE : G_ST_PRE) + (c - C_P) * ST_CHUNK;
    };
    // ---- atom waves: the fp32 node features from the HBM scratch, requested FIRST (the round trip runs under the scalar chain of
    // the scales below): the B operand of W3a' (POST) or of the projections, and the residual of the node MLP's second layer
    float4 hold[8];
    if (awave) {
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) hold[ot] = *st_tile(hs + HS_HT, ta, ot, lane);
    }
    // ---- scales (a-priori bounds, exactly version 2's)
    const int par = cx.par;
    float hmax = 0.f, aggmax = 0.f, s_agg = 1.f, s_t = 1.f, s_hn = 1.f, inv2 = 1.f;
    if constexpr (POST) {
        hmax = __uint_as_float(v.fmax[FM_H0 + par]);
        aggmax = __uint_as_float(v.fmax[FM_AGG]);
        s_agg = scale_for(aggmax);
        const float y3b = cload(sc, SC_L1_W3A) * hmax + cload(sc, SC_L1_W3B) * aggmax + cload(sc, SC_B3);
        s_t = scale_for(y3b);
        const float hnb = hmax + cload(sc, SC_L1_W4) * y3b + cload(sc, SC_B4);
        s_hn = scale_for(hnb);
        if (tid == 0 && (beyond_f16_range(aggmax) || beyond_f16_range(y3b) || beyond_f16_range(hnb))) atomicOr(&v.misc[1], NAN_RANGE | 3);
        inv2 = inv_pow2(s_t * cload(sc, 4));
    }
    // scales of the node features as a B operand: of the h this phase starts with (measured max |h|) and, POST, of the new h (its bound)
    const float s_h0 = scale_for(__uint_as_float(v.fmax[FM_H0 + par]));
    const float s_hf = POST ? s_hn : s_h0;
    // ---- the aggregate as fp32 rows (POST): into the slot the FOURTH chunk will take, free until chunk 0 has been consumed
    constexpr int AGG_OFF = st_slot_off(st_slot(3, NC));
    if constexpr (POST) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = tid + THREADS * k;
            if (e < nown * 32) *reinterpret_cast<float4*>(lds0 + AGG_OFF + (e >> 5) * LDH + 4 * (e & 31)) = ar->v[k];
        }
    }
    // ---- TWO PROGRAMS from here to the end of the ring, one per wave kind, with the same number of workgroup barriers and no
    // control-flow merge in between: the compiler's wait-count pass treats an LDS-DMA in flight on ANY path into a block as
    // a reason to drain the vector-memory counter before the block's first LDS read - with the loaders' requests and the atom
    // waves' LDS reads in one flow, every chunk of an atom wave began with s_waitcnt vmcnt(0), i.e. with a wait for its own
    // scratch stores and bias loads (measured round 6: 2.4 K cycles per chunk instead of 0.9 K)
    if (loader) {
        // chunk C has landed -> barrier -> request chunk C + 2 (the slot of chunk C - 2 is free: every atom wave is past
        // it) and, two chunks before the end, the next pair loop's image
        // (chunk 0 ALONE first: issuing a 1 KB piece costs a wave ~90 cycles, and nothing can start before this chunk has
        // landed - chunks 1 and 2 follow behind the first barrier, under the atom waves' work on chunk 0)
        st_issue<st_slot(0, NC)>(lds0, chunk_src(0), hw, lane);
        prof_event(pf, w, lane, 200);
        auto lstep = [&](auto C_) {
            constexpr int C = decltype(C_)::value;
            // outstanding behind chunk C at this point: chunk C + 1 (not while C == 0) and what has been requested of the image -
            // its first half with chunk NC - 1 (ring slot 0 = that half of the image region is free from barrier NC - 3 on), its
            // second half one barrier later
            constexpr int behind = (C >= 1 && C + 1 < NC ? ST_PIECES : 0) + (C == NC - 2 ? ST_PIECES : 0) + (C == NC - 1 ? 2 * ST_PIECES : 0);
            st_wait_vm<behind>();
            fine_event(pf, w, lane, 220 + C);        // (diagnostics builds) this loader's part of chunk C has landed
            lds_barrier();
            fine_event(pf, w, lane, 210 + C);
            if constexpr (C == 0) {
                st_issue<st_slot(1, NC)>(lds0, chunk_src(1), hw, lane);
                st_issue<st_slot(2, NC)>(lds0, chunk_src(2), hw, lane);
            }
            if constexpr (C == 1) st_wait_vm<ST_PIECES>();       // (chunk 1 before chunk 3 is requested: the count below assumes one chunk behind)
            if constexpr (C >= 1 && C + 2 < NC) st_issue<st_slot(C + 2, NC)>(lds0, chunk_src(C + 2), hw, lane);
            if constexpr (C == NC - 3) st_issue_image<0>(v, nx, hw, lane);
            if constexpr (C == NC - 2) st_issue_image<1>(v, nx, hw, lane);
        };
        lstep(IC<0>{}); lstep(IC<1>{}); lstep(IC<2>{}); lstep(IC<3>{});
        if constexpr (NC > 4) { lstep(IC<4>{}); lstep(IC<5>{}); }
        if constexpr (NC > 6) { lstep(IC<6>{}); lstep(IC<7>{}); }
        if constexpr (NC > 8) { lstep(IC<8>{}); lstep(IC<9>{}); }
        fine_event(pf, w, lane, 240);
        lds_barrier();                               // the ring is done
        prof_event(pf, w, lane, 250);
    } else {
    // ---- atom waves
    // B operands: the node features this phase starts with (W3a' in POST; the projections otherwise), the aggregate, and the
    // operand being filled by the epilogues (t, then the new h)
    BOp bh, bagg, bout, bin;
    floatx4 acc[4];
    float4 t0p[4];                                   // W3a' h + b3 of the four tiles whose W3b' chunk comes next
    float4 res2[8], bb4[8];                          // residual (the node features once more) and b4: requested one chunk before their use
    float4 Pout[8], Qout[8];
    float hm = 0.f;
    float S1 = 1.f;
    if (!TEAM) {
        const float x2 = __uint_as_float(v.fmax[FM_X2]), x02 = __uint_as_float(v.fmax[FM_X02]);
        S1 = fminf(scale_for(4.0f * x2) * scale_for(cload(scn, 6)), scale_for(4.0f * x02) * scale_for(cload(scn, 7)));
    }
    // (the columns of a tile's padding atoms - 14 of 64 at n = 50 - enter every GEMM as ZEROS: their operand scales are 0.  Nothing
    // of them is ever read, and a zero column costs the matrix pipe less energy, which under the power cap is clock)
    const float live = valid ? 1.0f : 0.0f;
    if (awave) {
        // the B operand of W3a' / of the projections from the fp32 node features
        const float sb = s_h0 * live;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const float4 a = hold[2 * sl], c4 = hold[2 * sl + 1];
            const float u[8] = {a.x * sb, a.y * sb, a.z * sb, a.w * sb, c4.x * sb, c4.y * sb, c4.z * sb, c4.w * sb};
            split8t(u, bh.hi[sl], bh.lo[sl]);
        }
    }
    prof_event(pf, w, lane, 200);
    // one step per chunk: barrier -> the four output tiles of chunk C -> their epilogue
    auto step = [&](auto C_) {
        constexpr int C = decltype(C_)::value;
        fine_event(pf, w, lane, 230 + C);            // (diagnostics builds) this wave is done with chunk C - 1
        lds_barrier();
        fine_event(pf, w, lane, 210 + C);
        if (!awave) return;
        // what chunk C is: POST 0 / 2: W3a' tiles 0-3 / 4-7, 1 / 3: W3b' of the same tiles, 4 / 5: W4'; then P, P, Q, Q
        constexpr bool IS_T0 = POST && C < 4 && (C & 1) == 0, IS_MLP1 = POST && C < 4 && (C & 1) == 1, IS_MLP2 = POST && (C == 4 || C == 5);
        constexpr bool IS_P = C >= C_P && C < C_P + 2;
        constexpr int half = (POST && C < 4) ? (C >> 1) : (C & 1);       // tiles 4 half .. 4 half + 3 = k-slabs 2 half, 2 half + 1 of the result
        // the bias of the chunk's tiles (P: b1' / b5', W3a': b3'), requested BEFORE its matrix instructions and before its
        // stores: behind a store the compiler cannot prove disjoint it would wait for the store's completion first
        float4 bias[4];
        if constexpr (IS_P || IS_T0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                bias[t] = *reinterpret_cast<const float4*>((IS_T0 ? g + G_VEC + 4 * HID : vecn) + 16 * (4 * half + t) + 4 * kg);
        }
        if constexpr (POST && C == 3) {
            // (not kept from the phase's first loads: 64 registers across the first layer's four chunks end up in scratch)
#pragma unroll
            for (int ot = 0; ot < 8; ++ot) {
                res2[ot] = *st_tile(hs + HS_HT, ta, ot, lane);
                bb4[ot] = *reinterpret_cast<const float4*>(g + G_VEC + 5 * HID + 16 * ot + 4 * kg);
            }
        }
        if constexpr (POST && C == 0) st_load_rows(bagg, lds0 + AGG_OFF, lc, kg, s_agg * live);
        if constexpr (IS_T0) st_mma_chunk<st_slot(C, NC)>(lds0, bh, acc, lane);
        else if constexpr (IS_MLP1) st_mma_chunk<st_slot(C, NC)>(lds0, bagg, acc, lane);
        else if constexpr (IS_MLP2) st_mma_chunk<st_slot(C, NC)>(lds0, bin, acc, lane);
        else st_mma_chunk<st_slot(C, NC)>(lds0, POST ? bin : bh, acc, lane);
        // epilogue of tile t of the chunk: the four values of this lane (features 16 ot + 4 kg + 0..3 of its atom)
        float4 res[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ot = 4 * half + t, nt = ot >> 1;
            if constexpr (IS_T0) {
                // the half of the node MLP's first layer that reads h: W3a' h + b3, waiting for its other half
                const float inv = inv_pow2(s_h0 * cload(sc, GS_SW_W3A + nt));
                const float4 b3 = bias[t];
                t0p[t] = make_float4(fmaf(acc[t][0], inv, b3.x), fmaf(acc[t][1], inv, b3.y), fmaf(acc[t][2], inv, b3.z), fmaf(acc[t][3], inv, b3.w));
            } else if constexpr (IS_MLP1) {
                // node MLP layer 1: t = SiLU(W3a' h + b3 + W3b' agg), times s_t 2^n_tile
                const float inv = inv_pow2(s_agg * cload(sc, GS_SW_W3B + nt)), stn = s_t * cload(sc, GS_NT + nt) * live;
                const float4 t0 = t0p[t];
                res[t] = make_float4(silu_u(fmaf(acc[t][0], inv, t0.x)) * stn, silu_u(fmaf(acc[t][1], inv, t0.y)) * stn,
                                     silu_u(fmaf(acc[t][2], inv, t0.z)) * stn, silu_u(fmaf(acc[t][3], inv, t0.w)) * stn);
            } else if constexpr (IS_MLP2) {
                // node MLP layer 2 + residual: the new h -> HBM scratch (fp32), max |h|, B operand of the projections
                const float4 hd = res2[ot], b4 = bb4[ot];
                const float4 hv = make_float4(fmaf(acc[t][0], inv2, hd.x + b4.x), fmaf(acc[t][1], inv2, hd.y + b4.y),
                                              fmaf(acc[t][2], inv2, hd.z + b4.z), fmaf(acc[t][3], inv2, hd.w + b4.w));
                *st_tile(hs + HS_HT, ta, ot, lane) = hv;
                if (valid) hm = fmaxf(fmaxf(hm, fmaxf(fabsf(hv.x), fabsf(hv.y))), fmaxf(fabsf(hv.z), fabsf(hv.w)));
                const float shl = s_hn * live;
                res[t] = make_float4(hv.x * shl, hv.y * shl, hv.z * shl, hv.w * shl);
            } else if constexpr (IS_P) {
                // P = W1a' h + b1 (W5a' h + b5): kept in registers until the ring has let go of the P region
                const float inv = inv_pow2(s_hf * cload(scn, (NEXT_EQ ? ES_SW_W5A : GS_SW_W1A) + nt));
                const float4 b1 = bias[t];
                Pout[ot] = make_float4(fmaf(acc[t][0], inv, b1.x), fmaf(acc[t][1], inv, b1.y), fmaf(acc[t][2], inv, b1.z), fmaf(acc[t][3], inv, b1.w));
            } else {
                // Q = W1b' h (W5b' h), times the geometric scale S1 (a team applies its own after the exchange)
                const float inv = inv_pow2(s_hf * cload(scn, (NEXT_EQ ? ES_SW_W5A : GS_SW_W1A) + 4 + nt)) * S1;
                Qout[ot] = make_float4(acc[t][0] * inv, acc[t][1] * inv, acc[t][2] * inv, acc[t][3] * inv);
            }
        }
        if constexpr (IS_MLP1 || IS_MLP2) {
            // the chunk's results as the next GEMM's B operand: k-slabs 2 half, 2 half + 1
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const float4 a = res[2 * sp], c4 = res[2 * sp + 1];
                const float u[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
                split8t(u, bout.hi[2 * half + sp], bout.lo[2 * half + sp]);
            }
            if constexpr (half == 1) bin = bout;
            if constexpr (C == 5) block_max(&v.fmax[FM_H0 + (par ^ 1)], hm, lane);
        }
    };
    step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{});
    if constexpr (NC > 4) { step(IC<4>{}); step(IC<5>{}); }
    if constexpr (NC > 6) { step(IC<6>{}); step(IC<7>{}); }
    if constexpr (NC > 8) { step(IC<8>{}); step(IC<9>{}); }
    fine_event(pf, w, lane, 240);
    lds_barrier();                                   // the ring is done: the P and Q regions are free
    prof_event(pf, w, lane, 250);
    if (valid) {
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) {
            *reinterpret_cast<float4*>(v.A + l * LDH + 16 * ot + 4 * kg) = Pout[ot];
            *reinterpret_cast<float4*>(v.B + l * LDH + 16 * ot + 4 * kg) = Qout[ot];
        }
    }
    }
    if (tid == 0) {
        if constexpr (POST) { v.fmax[FS_HS] = __float_as_uint(s_hn); v.misc[CX_PAR] = par ^ 1; }
        v.misc[CX_PASS] = pass_open;
    }
    if constexpr (TEAM) {
        prof_event(pf, w, lane, 110);
        lds_barrier();                               // own Q rows complete
        team_exchange_q<1>(v, nb, tid, scn, pass_open == 0, POST ? (par ^ 1) : par, pf);
    }
    prof_event(pf, w, lane, 11);
    dma_wait();                                      // (loaders: the image; atom waves: their scratch stores are not waited for by anybody else)
    lds_barrier();
}


// GCL (egnn.py:45-80), per-atom phases version 3 (f16 modes): the pair loop as before, then stream_phase
template <int PREC, bool TEAM, bool ATT>
__device__ __forceinline__ void gcl_pass3(const Lds& v, Prof& pf) {
    {   // ---- pair loop (version 2's, unchanged)
    const PassCtx cx = pass_ctx(v);
    const int nb = cx.nb, N = cx.N, par = cx.par;
    const int8_t* emask = cx.em;
    const float* sc = cx.g + G_SCALE;
    const LaneIds q = lane_ids(v);
    const int tid = q.tid, w = q.w, lane = q.lane;
    prof_event(pf, w, lane, 12);
    if (tid == 0) { v.fmax[FM_H0 + (par ^ 1)] = 0u; v.fmax[FM_AGG] = 0u; }
    const float hmax = __uint_as_float(v.fmax[TEAM ? FM_HG : FM_H0 + par]);
    const float x2 = __uint_as_float(v.fmax[FM_X2]), x02 = __uint_as_float(v.fmax[FM_X02]);
    const float pqb = (cload(sc, SC_L1_W1A) + cload(sc, SC_L1_W1B)) * hmax + cload(sc, SC_B1);
    const float u1b = pqb + 4.0f * (x2 * cload(sc, GS_WRW) + x02 * cload(sc, GS_WDW));
    const float sa = scale_for(u1b);
    const float accs = sa * cload(sc, 5);
    if (tid == 0 && (beyond_f16_range(u1b) || beyond_f16_range(hmax) || beyond_f16_range(4.0f * x2) || beyond_f16_range(4.0f * x02)))
        atomicOr(&v.misc[1], NAN_RANGE | 3);
    if constexpr (ATT) pair_phase<false, PREC, true, TEAM>(v, nb, w, lane, emask, N, 0.0f, sa, inv_pow2(accs), sc, cload(sc, 8), pf);
    else pair_phase<false, PREC, false, TEAM>(v, nb, w, lane, emask, N, 0.0f, sa, inv_pow2(accs), sc, 0.0f, pf);
    prof_event(pf, w, lane, 13);
    }
    AggRegs ar;
    bool next_eq;
    {
        const PassCtx cx = pass_ctx(v);
        const int nown = TEAM ? ctx_i(v, TM_NOWN) : cx.nb;
        const LaneIds q = lane_ids(v);
        next_eq = cx.nx.equiv;
        lds_barrier();                         // partial rows complete
        prof_event(pf, q.w, q.lane, 20);
        const float am = pair_reduce_gcl(v, nown, cx.nb, q.tid, ar, (cx.flags & 4) ? 1.0f / float(cx.N) : 1.0f);
        block_max(&v.fmax[FM_AGG], am, q.lane);
        prof_event(pf, q.w, q.lane, 21);
        lds_barrier();                         // every partial read: the P, Q, h, W2' regions are free; max |agg| known
        prof_event(pf, q.w, q.lane, 22);
    }
    if (next_eq) stream_phase<TEAM, true, true>(v, pf, &ar);
    else stream_phase<TEAM, true, false>(v, pf, &ar);
}

// EquivariantUpdate (egnn.py:101-125), per-atom phases version 3 (f16 modes)
template <int PREC, bool TEAM>
__device__ __forceinline__ void equiv_pass3(const Lds& v, Prof& pf) {
    {   // ---- pair loop (version 2's)
    const PassCtx cx = pass_ctx(v);
    const int nb = cx.nb, N = cx.N, par = cx.par;
    const int8_t* emask = cx.em;
    const float* sc = cx.g + E_SCALE;
    const float norm_constant = ctx_f(v, CX_NORMC);
    const LaneIds q = lane_ids(v);
    const int w = q.w, lane = q.lane;
    prof_event(pf, w, lane, 32);
    const float hmax = __uint_as_float(v.fmax[TEAM ? FM_HG : FM_H0 + par]);
    const float x2 = __uint_as_float(v.fmax[FM_X2]), x02 = __uint_as_float(v.fmax[FM_X02]);
    const float pqb = (cload(sc, SCE_L1_W5A) + cload(sc, SCE_L1_W5B)) * hmax + cload(sc, SCE_B5);
    const float u1b = pqb + 4.0f * (x2 * cload(sc, ES_WRW) + x02 * cload(sc, ES_WDW));
    const float sa = scale_for(u1b);
    const float accs = sa * cload(sc, 2);
    if (q.tid == 0 && (beyond_f16_range(u1b) || beyond_f16_range(hmax) || beyond_f16_range(4.0f * x2) || beyond_f16_range(4.0f * x02)))
        atomicOr(&v.misc[1], NAN_RANGE | 3);
    // (the finiteness proof of the skipped sums: see equiv_pass2)
    const float phi = cload(sc, ES_W7L1) * fmaf(cload(sc, ES_L1_W6), u1b, cload(sc, ES_B6));
    const bool proven = 2.0f * float(nb) * phi < 1e37f;
    if (!proven) {
        const bool widen = ctx_i(v, MS_FULL) == 0;
        __syncthreads();
        if (widen) {
            const int tid_ = q.tid;
            receivers_all(v, TEAM ? ctx_i(v, TM_NOWN) : nb, tid_);
            __syncthreads();
        }
    }
    pair_phase<true, PREC, false, TEAM>(v, nb, w, lane, emask, N, norm_constant, sa, inv_pow2(accs), sc,
                                        (cx.flags & 2) ? ctx_f(v, CX_CRANGE) : 0.0f, pf);
    prof_event(pf, w, lane, 33);
    }
    // ---- coordinate update of the own atoms
    const PassCtx cx = pass_ctx(v);
    const int nb = cx.nb, N = cx.N;
    const int nown = TEAM ? ctx_i(v, TM_NOWN) : nb;
    const int S = TEAM ? ctx_i(v, TM_S) : 1, rank = TEAM ? ctx_i(v, TM_RANK) : 0;
    const bool more = cx.nx.base != nullptr;
    const LaneIds q = lane_ids(v);
    const int tid = q.tid, lane = q.lane;
    lds_barrier();                         // partial triples complete
    const float xscale = (cx.flags & 4) ? 1.0f / float(N) : ((cx.flags & 2) ? ctx_f(v, CX_INVNORM) : 1.0f);
    pair_reduce_equiv<true>(v, nown, nb, tid, xscale);
    if (tid == 0) v.fmax[TEAM ? FM_XOWN : FM_X2] = 0u;
    lds_barrier();
    const float* aggx = v.A - L_A + L_AGGX3;
    float n2 = 0.0f;
    if (tid < nown) {
        const float lm = v.lm[tid];
        const int a = rank + tid * S;
        const bool moves = v.rpos[tid] >= 0;                     // off the receiver list: linker mask 0, nothing was summed
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float xn = v.xs[4 * a + k];
            if (moves) {
                xn += aggx[4 * tid + k] * lm;
                v.xs[4 * a + k] = xn;
            }
            n2 = fmaf(xn, xn, n2);
        }
    }
    block_max(&v.fmax[TEAM ? FM_XOWN : FM_X2], n2, lane);       // (a team: the own atoms; the exchange headers carry it)
    prof_event(pf, q.w, lane, 34);
    lds_barrier();
    prof_event(pf, q.w, lane, 10);
    if (more) stream_phase<TEAM, false, false>(v, pf, nullptr);
}

template <bool TEAM>
__device__ __forceinline__ void head_phase(const Lds& v, bool v3);

// Dynamics.forward for the own atoms of the molecule resident in LDS: reads v.z (state), the linker mask v.lm, the context
// words (CX_*: sizes, pointers, model flags, time feature; set by the kernel - CX_PASS / CX_PAR are reset here);
// writes eps_hat[l][0:3+nf] of own atom l into v.A (row stride DMAX) and ORs NaN bits into v.misc[1].
template <int PREC, bool TEAM, bool ATT>
__device__ __forceinline__ void forward_molecule2(const Lds& v, Prof& pf) {
    const int tid = lane_ids(v).tid;
    const int nb = ctx_i(v, 0);
    const int nown = TEAM ? ctx_i(v, TM_NOWN) : nb;
    const int S = TEAM ? ctx_i(v, TM_S) : 1, rank = TEAM ? ctx_i(v, TM_RANK) : 0;
    const float* wp = ctx_p<const float>(v, CX_WP);
    float* hs = ctx_p<float>(v, CX_HS);
    const float tfeat = ctx_f(v, CX_TFEAT);
    const int nf = ctx_i(v, CX_NF), fin = ctx_i(v, CX_FIN);
    const int npass = ctx_i(v, CX_NPASS);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // Per-atom phases: version 3 (stream_phase: atom-stationary GEMM chain, weights streamed through LDS) where a workgroup holds
    // a whole molecule of MORE THAN 32 atoms - three or four 16-atom tiles, one wave each -, in the f16 modes; version 2 otherwise:
    // with one or two tiles a step of the stream is a chain of latencies whatever its size, and version 2's seven barriers
    // per pass beat its twelve (measured round 6, same box: n = 30: version 3 -9 %; teams of 2 / 4: -2 % / 0; n = 35 / 44 / 50
    // on one compute unit: +4 / +2.5 / +1 %, the ragged C2 chain +4.6 %).  A property of the workgroup, fixed for the launch
    const bool v3 = (PREC != 0) && !TEAM && nown > 32;
    if (tid == 0) { v.misc[CX_PASS] = v3 ? -1 : 0; v.misc[CX_PAR] = 0; v.misc[CX_V3] = v3 ? 1 : 0; }     // (version 3 opens pass CX_PASS + 1)
    if (tid < 8) reinterpret_cast<int*>(v.A - L_A + L_PROG)[tid] = 0;      // pair-loop progress slots (pair_phase); barriers follow
    if (!TEAM && tid < HID) (v.A - L_A + L_ZROW)[tid] = 0.0f;               // the zero row of the idle slot-steps (pair_phase); barriers follow
    prof_event(pf, w, lane, 1);
    {
        if (tid < 8) v.fmax[tid] = 0u;
        __syncthreads();
    }
    const NextPass first = {wp + OFF_BLOCKS, false};
    if (!v3) stage_next(v, first, w, tid);                          // first pass's W2' image (v.W, v.vec are free here); version 3: its loaders'
    // coordinates at entry of the own atoms (x, and x0 for the d0 edge attribute and the velocity); a team gets everybody's
    // from its first exchange
    if (tid < 4 * nown) {
        const int l = tid >> 2, k = tid & 3, a = rank + l * S;
        const float xv = (k < 3) ? v.z[l * DMAX + k] : 0.0f;
        v.xs[4 * a + k] = xv;
        v.x0[4 * a + k] = xv;
    }
    {                                          // max |x|^2 at entry (a team: of the own atoms, for its first exchange header)
        float n2 = 0.0f;
        if (tid < nown) {
            const float x0 = v.z[tid * DMAX], x1 = v.z[tid * DMAX + 1], x2 = v.z[tid * DMAX + 2];
            n2 = x0 * x0 + x1 * x1 + x2 * x2;
        }
        const unsigned b = wave_max_u32(__float_as_uint(n2));
        if (lane == 0) {
            if (TEAM) lds_max_u32(&v.fmax[FM_XOWN], b);
            else { lds_max_u32(&v.fmax[FM_X2], b); lds_max_u32(&v.fmax[FM_X02], b); }
        }
    }
    // embedding: h = We * [h_feat, t, context] + be   (egnn.py:396-407, :224) -> fp32 rows in v.B (and the HBM scratch)
    {
        const int f = tid & (HID - 1);
        float wrow[FINP];
        const float4* wsrc = reinterpret_cast<const float4*>(wp + OFF_EMB_W + f * FINP);
#pragma unroll
        for (int q = 0; q < FINP / 4; ++q) {
            const float4 t4 = wsrc[q];
            wrow[4 * q] = t4.x; wrow[4 * q + 1] = t4.y; wrow[4 * q + 2] = t4.z; wrow[4 * q + 3] = t4.w;
        }
        const float be = wp[OFF_EMB_B + f];
        const float* ctxp = ctx_p<const float>(v, CX_CTXP);                 // this molecule's context rows [N][nctx]
        const int ct = ctx_i(v, CX_CT);                                     // condition_time: [h_feat, t, context] or [h_feat, context] (egnn.py:396-407)
        const int nctx = fin - nf - ct;
        float hmax = 0.0f;
        for (int l = tid >> 7; l < nown; l += THREADS / HID) {
            const int pos = v.idx[rank + l * S];
            float acc = be;
#pragma unroll
            for (int k = 0; k < FINP; ++k) {
                float hin = 0.0f;
                if (k < nf) hin = v.z[l * DMAX + 3 + k];
                else if (k < nf + ct) hin = tfeat;
                else if (k < fin) hin = ctxp[pos * nctx + (k - nf - ct)];
                acc = fmaf(wrow[k], hin, acc);
            }
            if (v3) {
                // version 3: st_tile's layout - tile (l / 16, f / 16), lane 16 ((f / 4) & 3) + l % 16, component f & 3
                hs[HS_HT + ((((l >> 4) * 8 + (f >> 4)) * 64 + 16 * ((f >> 2) & 3) + (l & 15)) * 4) + (f & 3)] = acc;
            } else {
            v.B[l * LDH + f] = acc;
            {   // accumulator-tile copy (tile_store16's layout): row l of tile (l / 32, f / 32) sits in register (r & 3) + 4 (r >> 3) of lane half (r >> 2) & 1
                const int r = l & 31;
                hs[HS_HT + ((((l >> 5) * 4 + (f >> 5)) * 4 + (r >> 3)) * 64 + (f & 31) + 32 * ((r >> 2) & 1)) * 4 + (r & 3)] = acc;
            }
            }
            hmax = fmaxf(hmax, fabsf(acc));
        }
        block_max(&v.fmax[FM_H0], hmax, lane);
    }
    __syncthreads();
    if (v3) {
        if constexpr (PREC != 0 && !TEAM) {
        // version 3: the node features live in the HBM scratch (fp32, written above); the projections of the first pass
        if (tid == 0 && beyond_f16_range(__uint_as_float(v.fmax[FM_H0]))) atomicOr(&v.misc[1], NAN_RANGE | 3);
        prof_event(pf, w, lane, 2);
        stream_phase<TEAM, false, false>(v, pf, nullptr);
        const int sub = ctx_i(v, CX_SUB);
#pragma nounroll
        for (int p = 0; p < npass; p += sub + 1) {
#pragma nounroll
            for (int gi = 0; gi < sub; ++gi) gcl_pass3<PREC, TEAM, ATT>(v, pf);
            equiv_pass3<PREC, TEAM>(v, pf);
        }
        prof_event(pf, w, lane, 3);
        __syncthreads();                                           // the h tiles in the HBM scratch: written by other lanes
        head_phase<TEAM>(v, true);
        }
    } else {
    {   // fragment rows of the embedded h -> v.C
        const float s0 = (PREC != 0) ? scale_for(__uint_as_float(v.fmax[FM_H0])) : 1.0f;
        for (int e = tid; e < nown * 32; e += THREADS)
            put_quad<PREC>(v.C + (e >> 5) * LDH, 4 * (e & 31), *reinterpret_cast<const float4*>(v.B + (e >> 5) * LDH + 4 * (e & 31)), s0);
        if (PREC != 0 && tid == 0) {
            v.fmax[FS_HS] = __float_as_uint(s0);
            if (beyond_f16_range(__uint_as_float(v.fmax[FM_H0]))) atomicOr(&v.misc[1], NAN_RANGE | 3);
        }
    }
    __syncthreads();
    prof_event(pf, w, lane, 2);
    {
        PreW2 pw;
        load_pre2(pw, first, w, lane, nown);
        open_pass<PREC, TEAM>(v, nb, nown, first, hs, pf, pw, 0, 0);
    }
    const int sub = ctx_i(v, CX_SUB);
#pragma nounroll
    for (int p = 0; p < npass; p += sub + 1) {
#pragma nounroll
        for (int gi = 0; gi < sub; ++gi) gcl_pass2<PREC, TEAM, ATT>(v, pf);
        equiv_pass2<PREC, TEAM>(v, pf);
    }
    prof_event(pf, w, lane, 3);
    __syncthreads();                                               // the h tiles in the HBM scratch: written by other lanes
    // output head: h_final = (Wo h + bo)[:nf], vel = x - x0   (egnn.py:235-237, :420, :430-435)
    head_phase<TEAM>(v, false);
    }
    prof_event(pf, w, lane, 4);
}

// (its own context reads: the pass loop above must not keep these alive)
template <bool TEAM>
__device__ __forceinline__ void head_phase(const Lds& v, bool v3) {
    const int tid = lane_ids(v).tid;
    const int nb = ctx_i(v, 0), nf = ctx_i(v, CX_NF);
    const int nown = TEAM ? ctx_i(v, TM_NOWN) : nb;
    const int S = TEAM ? ctx_i(v, TM_S) : 1, rank = TEAM ? ctx_i(v, TM_RANK) : 0;
    const float* wp = ctx_p<const float>(v, CX_WP);
    const float* hs = ctx_p<const float>(v, CX_HS);
    float* eps = v.A;
    int nanbits = 0;
    for (int e = tid; e < nown * nf; e += THREADS) {
        const int a = e / nf, o = e - a * nf;
        // row a of the accumulator-order tiles: 32 consecutive floats per feature tile (k ascending: the reference's order)
        const int r = a & 31;
        const float* wo = wp + OFF_OUT_W + o * HID;
        float acc = wp[OFF_OUT_B + o];
        if (v3) {
            // per-atom phases version 3: the node features of atom a in the tiles of wave a / 16 (st_tile), k ascending as well
            const float* hp = hs + HS_HT + ((a >> 4) * 8 * 64 + (a & 15)) * 4;
#pragma unroll
            for (int ot = 0; ot < 8; ++ot)
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    const float4 hv = *reinterpret_cast<const float4*>(hp + (ot * 64 + 16 * kg) * 4);
                    const float4 wv = *reinterpret_cast<const float4*>(wo + 16 * ot + 4 * kg);
                    acc = fmaf(hv.x, wv.x, acc); acc = fmaf(hv.y, wv.y, acc);
                    acc = fmaf(hv.z, wv.z, acc); acc = fmaf(hv.w, wv.w, acc);
                }
        } else {
        const float* hp = hs + HS_HO + ((a >> 5) * 4 * 16 + (r & 3) + 4 * (r >> 3)) * 64 + 32 * ((r >> 2) & 1);
#pragma unroll
        for (int nt4 = 0; nt4 < 4; ++nt4)
#pragma unroll 8
            for (int q = 0; q < 8; ++q) {
                const float4 hv = *reinterpret_cast<const float4*>(hp + nt4 * 16 * 64 + 4 * q);
                const float4 wv = *reinterpret_cast<const float4*>(wo + 32 * nt4 + 4 * q);
                acc = fmaf(hv.x, wv.x, acc); acc = fmaf(hv.y, wv.y, acc);
                acc = fmaf(hv.z, wv.z, acc); acc = fmaf(hv.w, wv.w, acc);
            }
        }
        (void)r;
        eps[a * DMAX + 3 + o] = acc;
        if (acc != acc) nanbits |= 2;
    }
    if (tid < 4 * nown && (tid & 3) < 3) {
        const int l = tid >> 2, a = rank + l * S;
        const float vel = v.xs[4 * a + (tid & 3)] - v.x0[4 * a + (tid & 3)];
        eps[l * DMAX + (tid & 3)] = vel;
        if (vel != vel) nanbits |= 1;
    }
    if (TEAM && tid == 0 && v.misc[TM_FAIL] != 0) nanbits |= 8;       // the team never assembled: the result is void
    if (nanbits) atomicOr(&v.misc[1], nanbits);
    __syncthreads();
}

// the coordinate-pass receiver list from the linker mask of the own atoms in v.lm (<= 55: one wave); visible after the next barrier
__device__ __forceinline__ void build_receivers(const Lds& v, int nown, int tid) {
    if (tid < 64) {
        const bool rec = tid < nown && v.lm[tid] != 0.0f;
        const unsigned long long bal = __ballot(rec);
        const int pos = __popcll(bal & ((1ull << tid) - 1ull));
        if (rec) v.rcv[pos] = tid;
        if (tid < NMAX + 1) v.rpos[tid] = rec ? pos : -1;
        if (tid == 0) { v.misc[MS_NRCV] = __popcll(bal); v.misc[MS_FULL] = (__popcll(bal) == nown) ? 1 : 0; }
    }
}

// compact the real atoms of molecule b: v.idx[0..n_b) = padded positions with node_mask != 0
__device__ __forceinline__ int compact_atoms(const Lds& v, const int8_t* __restrict__ node_mask_b, int N, int tid) {
    if (tid < 64) {
        int count = 0;
        for (int base = 0; base < N; base += 64) {
            const int a = base + tid;
            const bool real = (a < N) && (node_mask_b[a] != 0);
            const unsigned long long bal = __ballot(real);
            const int pos = count + __popcll(bal & ((1ull << tid) - 1ull));
            if (real && pos < NQMAX) v.idx[pos] = a;
            count += __popcll(bal);
        }
        if (tid == 0) { v.misc[0] = count; v.misc[1] = 0; }
    }
    __syncthreads();
    return v.misc[0];
}

// ---------------------------------------------------------------------------------------------------
// Kernels.  1: one Dynamics.forward per launch (src/egnn.py:374-447).  2: EDM.sample_chain (src/edm.py:126-242) - T reverse
// steps + final decode in ONE launch.  TEAM = false: one workgroup per molecule (batches that fill the chip - the benchmarked
// configuration); TEAM = true: a team of p.team workgroups per molecule.  Everything a phase needs is re-read from the LDS
// context block or from the kernel arguments (scalar cache) where it is used.
// ---------------------------------------------------------------------------------------------------
struct FwdArgs {
    const float* wpack;
    ModelDims md;
    int B, N;
    const float* xh;
    const float* t;
    int t_stride;
    const int8_t* node_mask;
    const float* linker_mask;
    const int8_t* edge_mask;
    const float* context;
    float* out;
    int* nan_flags;
    unsigned long long* prof;
    int team;                       // team kernels: workgroups per molecule; exchange buffers [B][TEAM_MOL_BYTES], arrival words [B][TEAM_MAX]
    int team_fault;                 // tests: member 1 gives up at once (dl_debug_team_fault)
    char* team_rows;
    unsigned* team_flags;
    float* hsave;                   // [workgroups][HS_STRIDE]: per-workgroup HBM scratch of the per-atom phases
};

// Team kernels: workgroup k -> (molecule slot, member index).  The members of a team sit 8 workgroups apart, which is the
// same XCD under the dispatch pattern observed on this chip (workgroup k -> XCD k % 8): the exchange then stays inside one
// L2.  Correctness does not depend on it.
struct TeamSlot {
    int slot, rank;
};
__device__ __forceinline__ TeamSlot team_slot(int k, int S) {
    const int group = k / (8 * S), within = k - group * 8 * S;
    TeamSlot t;
    t.slot = group * 8 + (within & 7);
    t.rank = within >> 3;
    return t;
}
// after compact_atoms (which ends with a barrier): the team block of v.misc; visible after the next barrier
__device__ __forceinline__ void team_init(const Lds& v, int nb, int S, int rank, char* rows, unsigned* flags, int fault) {
    const unsigned long long pr = reinterpret_cast<unsigned long long>(rows), pf = reinterpret_cast<unsigned long long>(flags);
    v.misc[TM_EPOCH] = 0; v.misc[TM_FAIL] = (fault != 0 && rank == 1) ? 1 : 0;
    v.misc[TM_ROWS] = int(unsigned(pr)); v.misc[TM_ROWS + 1] = int(unsigned(pr >> 32));
    v.misc[TM_FLAGS] = int(unsigned(pf)); v.misc[TM_FLAGS + 1] = int(unsigned(pf >> 32));
    v.misc[TM_S] = S; v.misc[TM_RANK] = rank;
    v.misc[TM_NOWN] = nb > rank ? (nb - rank + S - 1) / S : 0;
}

__device__ __forceinline__ void ctx_store(const Lds& v, const ModelDims& md, int N, const int8_t* em, float* hs, const float* wp,
                                          const float* ctx_rows, float tfeat, int mol) {
    v.misc[CX_N] = N;
    ctx_set_p(v, CX_EM, em); ctx_set_p(v, CX_HS, hs); ctx_set_p(v, CX_WP, wp); ctx_set_p(v, CX_CTXP, ctx_rows);
    v.misc[CX_PASS] = 0; v.misc[CX_PAR] = 0;
    v.misc[CX_FLAGS] = (md.attention ? 1 : 0) | (md.tanh ? 2 : 0) | (md.mean ? 4 : 0);
    v.misc[CX_NORMC] = __float_as_int(md.norm_constant); v.misc[CX_CRANGE] = __float_as_int(md.coords_range);
    v.misc[CX_INVNORM] = __float_as_int(md.inv_norm);
    v.misc[CX_NPASS] = (md.sub + 1) * md.n_layers; v.misc[CX_NF] = md.nf; v.misc[CX_FIN] = md.fin;
    v.misc[CX_SUB] = md.sub; v.misc[CX_CT] = md.ct;
    v.misc[CX_TFEAT] = __float_as_int(tfeat); v.misc[CX_MOL] = mol;
}

template <int PREC, bool TEAM, bool ATT>
__global__ void __launch_bounds__(THREADS) egnn_forward_fc_kernel(FwdArgs p) {
    __shared__ __attribute__((aligned(16))) float lds_raw[L_TOTAL];   // static: every LDS address is a constant
    const Lds v = lds_view(lds_raw);
    int b = blockIdx.x, rank = 0, S = 1;
    if constexpr (TEAM) {
        const TeamSlot ts = team_slot(blockIdx.x, p.team);
        if (ts.slot >= p.B) return;
        b = ts.slot; rank = ts.rank; S = p.team;
    }
    {
        const int tid = threadIdx.x;
        const int N = p.N, D = 3 + p.md.nf;
        const int limit = TEAM ? NQMAX : NMAX;
        const int8_t* nm = p.node_mask + size_t(b) * N;
        float* out_b = p.out + size_t(b) * N * D;
        const int nb = compact_atoms(v, nm, N, tid);
        // padded rows of the output are exactly zero (node_mask multiply, egnn.py:420,236-237)
        if (rank == 0)
            for (int e = tid; e < N * D; e += THREADS)
                if (nm[e / D] == 0 || nb > limit) out_b[e] = 0.0f;
        if (nb > limit || nb == 0) {
            if (rank == 0 && tid == 0) {
                if (TEAM) { if (nb > limit) atomicOr(&p.nan_flags[b], 4); }
                else p.nan_flags[b] = (nb > limit) ? 4 : 0;
            }
            return;
        }
        if (tid == 0) {
            if (TEAM) team_init(v, nb, S, rank, p.team_rows + size_t(b) * TEAM_MOL_BYTES, p.team_flags + size_t(b) * TEAM_MAX, p.team_fault);
            ctx_store(v, p.md, N, p.edge_mask ? p.edge_mask + size_t(b) * N * N : nullptr, p.hsave + size_t(blockIdx.x) * HS_STRIDE,
                      p.wpack, p.context ? p.context + size_t(b) * N * p.md.ctx : nullptr, p.t[size_t(b) * p.t_stride], b);
        }
        __syncthreads();
        const int nown = TEAM ? v.misc[TM_NOWN] : nb;
        const float* xh_b = p.xh + size_t(b) * N * D;
        for (int e = tid; e < nown * D; e += THREADS) {
            const int l = e / D, d = e - l * D;
            v.z[l * DMAX + d] = xh_b[v.idx[rank + l * S] * D + d];
        }
        if (tid < nown) v.lm[tid] = p.linker_mask ? p.linker_mask[size_t(b) * N + v.idx[rank + tid * S]] : 1.0f;
        __syncthreads();
        build_receivers(v, nown, tid);
        __syncthreads();
    }
    Prof pf;
    pf.buf = (blockIdx.x == 0) ? p.prof : nullptr;
    pf.n = 0;
    forward_molecule2<PREC, TEAM, ATT>(v, pf);
    {   // results of the own atoms (arguments and sizes re-read: see the pass context)
        const auto* P = kargs<FwdArgs>();
        const int tid = lane_ids(v).tid;
        const int mol = ctx_i(v, CX_MOL), nb = ctx_i(v, 0), N = P->N, D = 3 + P->md.nf;
        const int nown = TEAM ? ctx_i(v, TM_NOWN) : nb;
        const int S2 = TEAM ? ctx_i(v, TM_S) : 1, rank2 = TEAM ? ctx_i(v, TM_RANK) : 0;
        float* out_b = P->out + size_t(mol) * N * D;
        for (int e = tid; e < nown * D; e += THREADS) {
            const int l = e / D, d = e - l * D;
            out_b[v.idx[rank2 + l * S2] * D + d] = v.A[l * DMAX + d];
        }
        if (tid == 0) {
            if (TEAM) { if (v.misc[1]) atomicOr(&P->nan_flags[mol], v.misc[1]); }
            else P->nan_flags[mol] = v.misc[1];
        }
    }
}

struct ChainArgs {
    const float* wpack;
    ModelDims md;
    dl_chain_args a;
    unsigned long long* prof;
    char* team_rows;                // team kernels: exchange buffers [B][TEAM_MOL_BYTES], arrival words [B][TEAM_MAX] (inside a.workspace)
    unsigned* team_flags;
    int team_fault;                 // tests: member 1 gives up at once (dl_debug_team_fault)
    float* hsave;                   // [workgroups][HS_STRIDE]: per-workgroup HBM scratch of the per-atom phases
};

// one reverse step (or the final decode, q == T) for the own atoms of the molecule in LDS: denoiser, then the sampler
// algebra.  false: NaN, stop (one workgroup per molecule; a team goes on - its members must keep meeting - and only records it).
template <int PREC, bool TEAM, bool ATT>
__device__ __forceinline__ bool chain_step2(const Lds& v, int q) {
    {
        const auto* P = kargs<ChainArgs>();
        const int tid = lane_ids(v).tid;
        if (tid == 0) {
            const float t = (q == P->a.T) ? 0.0f : P->a.coefs[q].t;                  // last forward: p(x,h | z_0), edm.py:210-242
            v.misc[CX_TFEAT] = __float_as_int(t);
        }
        __syncthreads();
    }
    Prof pf;
    pf.buf = (blockIdx.x == 0 && q == 0) ? kargs<ChainArgs>()->prof : nullptr;
    pf.n = 0;
    forward_molecule2<PREC, TEAM, ATT>(v, pf);
    const auto* P = kargs<ChainArgs>();
    const int tid = lane_ids(v).tid;
    const int b = ctx_i(v, CX_MOL), nb = ctx_i(v, 0);
    const int nown = TEAM ? ctx_i(v, TM_NOWN) : nb;
    const int S = TEAM ? ctx_i(v, TM_S) : 1, rank = TEAM ? ctx_i(v, TM_RANK) : 0;
    const int N = P->a.N, nf = P->md.nf, D = 3 + nf, T = P->a.T, K = P->a.keep_frames, B = P->a.B;
    if (v.misc[1] != 0) {                                      // FoundNaNException (egnn.py:441-442)
        if (!TEAM) {
            if (tid == 0) { P->a.nan_flags[b] = v.misc[1]; P->a.nan_step[b] = q; }
            return false;
        }
        if (tid == 0) {                                        // first offending forward of any member
            atomicOr(&P->a.nan_flags[b], v.misc[1]);
            atomicCAS(&P->a.nan_step[b], -1, q);
        }
    }
    const bool decode = (q == T);
    dl_step_coef cf;
    if (decode) { cf.t = 0.0f; cf.alpha_ts = 1.0f; cf.c_eps = 0.0f; cf.sigma = 0.0f; }
    else cf = P->a.coefs[q];
    const float* noise_x = P->a.noise_x;
    const float* noise_h = P->a.noise_h;
    const float* fragm = P->a.fragment_mask;
    const bool philox = (noise_x == nullptr);
    const unsigned gmol = unsigned(P->a.mol_offset + (P->a.mol_index ? P->a.mol_index[b] : b));
    const unsigned long long seed = P->a.noise_seed;
    const size_t frame = size_t(B) * N * D;
    const size_t nx_stride = size_t(B) * N * 3, nh_stride = size_t(B) * N * nf;
    float* chain_b = P->a.chain + size_t(b) * N * D;
    const float norm_x = P->a.norm_x, norm_h = P->a.norm_h, bias_h = P->a.bias_h;
    const float inv_alpha0 = P->a.inv_alpha0, sigma0 = P->a.sigma0, sigma_x = P->a.sigma_x;
    const int s = T - 1 - q;
    const int widx = decode ? 0 : (s * K) / T;
    const bool last_writer = (s == 0) || (((s - 1) * K) / T != widx);
    const bool write = !decode && last_writer && widx != 0;   // frame 0 is overwritten by the decode
    for (int e = tid; e < nown * D; e += THREADS) {
        const int l = e / D, d = e - l * D;
        const int pos = v.idx[rank + l * S];
        const size_t n = size_t(b) * N + pos;
        const float lm = v.lm[l], fr = v.frag[l];
        const float zt = v.z[l * DMAX + d];
        const float eh = __fmul_rn(v.A[l * DMAX + d], lm);
        float nz;
        if (philox) nz = philox_normal(seed, gmol, unsigned(pos), unsigned(q + 1), unsigned(d));
        else nz = (d < 3) ? noise_x[(q + 1) * nx_stride + n * 3 + d] : noise_h[(q + 1) * nh_stride + n * nf + d - 3];
        float zn;
        if (!decode) {
            // z_s = z_t*frag + (z_t/alpha - c_eps*(eps*lm) + sigma*(noise*lm))*lm   (edm.py:196-206)
            const float mu = __fsub_rn(__fdiv_rn(zt, cf.alpha_ts), __fmul_rn(cf.c_eps, eh));
            const float zs = __fadd_rn(mu, __fmul_rn(cf.sigma, __fmul_rn(nz, lm)));
            zn = __fadd_rn(__fmul_rn(zt, fr), __fmul_rn(zs, lm));
            if (write) {                                   // chain[widx] = unnormalize_z(z) (edm.py:162-163)
                const float o = (d < 3) ? __fmul_rn(zn, norm_x) : __fadd_rn(__fmul_rn(zn, norm_h), bias_h);
                chain_b[widx * frame + pos * D + d] = o;
            }
        } else {
            // xh = z_0*frag + (1/alpha_0*(z_0 - sigma_0*eps) + sigma_x*(noise*lm))*lm, then unnormalize
            const float mu = __fmul_rn(inv_alpha0, __fsub_rn(zt, __fmul_rn(sigma0, eh)));
            const float xh = __fadd_rn(mu, __fmul_rn(sigma_x, __fmul_rn(nz, lm)));
            const float zz = __fadd_rn(__fmul_rn(zt, fr), __fmul_rn(xh, lm));
            zn = (d < 3) ? __fmul_rn(zz, norm_x) : __fadd_rn(__fmul_rn(zz, norm_h), bias_h);
        }
        v.z[l * DMAX + d] = zn;
    }
    (void)fragm;
    if (TEAM && tid == 0) v.misc[1] = 0;                       // the team goes on: later forwards report afresh (the first one is recorded)
    __syncthreads();
    return true;
}

template <int PREC, bool TEAM, bool ATT>
__global__ void __launch_bounds__(THREADS) sample_chain_fc_kernel(ChainArgs p) {
    __shared__ __attribute__((aligned(16))) float lds_raw[L_TOTAL];   // static: every LDS address is a constant
    const Lds v = lds_view(lds_raw);
    int T, qb, qe;
    {
        const dl_chain_args& g = p.a;
        const int tid = threadIdx.x;
        int k = blockIdx.x, rank = 0, S = 1;
        const int count = g.order_count > 0 ? g.order_count : g.B;        // molecules of THIS launch (a part of the batch, or all of it)
        if constexpr (TEAM) {
            const TeamSlot ts = team_slot(blockIdx.x, g.team);
            if (ts.slot >= count) return;
            k = ts.slot; rank = ts.rank; S = g.team;
        }
        const int b = g.order ? g.order[g.order_first + k] : k;
        const int N = g.N, nf = p.md.nf, D = 3 + nf, K = g.keep_frames, B = g.B;
        const int limit = TEAM ? NQMAX : NMAX;
        T = g.T;
        // a chain in two launches (dl_chain_args.q_begin / q_end): this launch's part of the T + 1 denoiser calls of molecule b
        qb = g.q_begin ? g.q_begin[b] : 0;
        qe = g.q_end ? g.q_end[b] : T + 1;
        if (g.skip_flags && g.skip_flags[b] != 0) return;          // (it ended in the first launch; every member takes this branch)
        const int8_t* nm = g.node_mask + size_t(b) * N;
        const size_t frame = size_t(B) * N * D;
        float* chain_b = g.chain + size_t(b) * N * D;
        const int nb = compact_atoms(v, nm, N, tid);
        if (rank == 0) {
            if (tid == 0) {
                if (TEAM) { if (nb > limit) atomicOr(&g.nan_flags[b], 4); }      // (a team's flags start at 0 / -1: set by the host)
                else { g.nan_flags[b] = (nb > limit) ? 4 : 0; g.nan_step[b] = -1; }
            }
            // padded rows of every frame are zero (z is masked; chain starts from torch.zeros, edm.py:143)
            for (int kf = 0; kf < K; ++kf)
                for (int e = tid; e < N * D; e += THREADS)
                    if (nm[e / D] == 0 || nb > limit) chain_b[kf * frame + e] = 0.0f;
        }
        if (nb > limit || nb == 0) return;
        if (tid == 0) {
            if (TEAM) team_init(v, nb, S, rank, p.team_rows + size_t(k) * TEAM_MOL_BYTES, p.team_flags + size_t(k) * TEAM_MAX, p.team_fault);
            ctx_store(v, p.md, N, g.edge_mask ? g.edge_mask + size_t(b) * N * N : nullptr, p.hsave + size_t(blockIdx.x) * HS_STRIDE,
                      p.wpack, g.context ? g.context + size_t(b) * N * p.md.ctx : nullptr, 0.0f, b);
        }
        __syncthreads();
        const int nown = TEAM ? v.misc[TM_NOWN] : nb;
        if (tid < nown) {
            const size_t n = size_t(b) * N + v.idx[rank + tid * S];
            v.lm[tid] = g.linker_mask[n];
            v.frag[tid] = g.fragment_mask[n];
        }
        __syncthreads();
        build_receivers(v, nown, tid);
        const bool philox = (g.noise_x == nullptr);                // draws generated in place (pack_layout.h: philox_normal)
        const unsigned gmol = unsigned(g.mol_offset + (g.mol_index ? g.mol_index[b] : b));     // global molecule index: the noise key
        // z = normalize(x,h) * fragment_mask + noise_0 * linker_mask   (edm.py:132-137,347-350)
        for (int e = tid; e < nown * D; e += THREADS) {
            const int l = e / D, d = e - l * D;
            const int pos = v.idx[rank + l * S];
            const size_t n = size_t(b) * N + pos;
            float val, eps0;
            if (d < 3) { val = __fdiv_rn(g.x[n * 3 + d], g.norm_x); eps0 = philox ? 0.0f : g.noise_x[n * 3 + d]; }
            else { val = __fdiv_rn(__fsub_rn(g.h[n * nf + d - 3], g.bias_h), g.norm_h); eps0 = philox ? 0.0f : g.noise_h[n * nf + d - 3]; }
            if (philox) eps0 = philox_normal(g.noise_seed, gmol, unsigned(pos), 0u, unsigned(d));
            const float lm = v.lm[l];
            float z0 = __fadd_rn(__fmul_rn(val, v.frag[l]), __fmul_rn(__fmul_rn(eps0, lm), lm));
            if (qb > 0) z0 = g.z_state[n * D + d];                 // resumed: the state the first launch left
            v.z[l * DMAX + d] = z0;
        }
        __syncthreads();
    }
#pragma nounroll
    for (int q = qb; q < qe; ++q)
        if (!chain_step2<PREC, TEAM, ATT>(v, q)) return;
    if (qe <= T) {                                                 // stopped early: hand the state over (dl_chain_args.z_state)
        const auto* P = kargs<ChainArgs>();
        const int tid = lane_ids(v).tid;
        const int b = ctx_i(v, CX_MOL), nb = ctx_i(v, 0), D = 3 + P->md.nf, N = P->a.N;
        const int nown = TEAM ? ctx_i(v, TM_NOWN) : nb;
        const int S = TEAM ? ctx_i(v, TM_S) : 1, rank = TEAM ? ctx_i(v, TM_RANK) : 0;
        for (int e = tid; e < nown * D; e += THREADS) {
            const int l = e / D, d = e - l * D;
            P->a.z_state[(size_t(b) * N + v.idx[rank + l * S]) * D + d] = v.z[l * DMAX + d];
        }
        return;
    }
    {   // frame 0: the final sample [x, one_hot(h)] of the own atoms
        const auto* P = kargs<ChainArgs>();
        const int tid = lane_ids(v).tid;
        const int b = ctx_i(v, CX_MOL), nb = ctx_i(v, 0), nf = P->md.nf, D = 3 + nf;
        const int nown = TEAM ? ctx_i(v, TM_NOWN) : nb;
        const int S = TEAM ? ctx_i(v, TM_S) : 1, rank = TEAM ? ctx_i(v, TM_RANK) : 0;
        if (tid < nown) {
            const int l = tid;
            float* o = P->a.chain + size_t(b) * P->a.N * D + v.idx[rank + l * S] * D;
            o[0] = v.z[l * DMAX + 0]; o[1] = v.z[l * DMAX + 1]; o[2] = v.z[l * DMAX + 2];
            int best = 0;                                          // torch.argmax: first maximal index
            float bv = v.z[l * DMAX + 3];
            for (int kk = 1; kk < nf; ++kk) {
                const float hv = v.z[l * DMAX + 3 + kk];
                if (hv > bv) { bv = hv; best = kk; }
            }
            for (int kk = 0; kk < nf; ++kk) o[3 + kk] = (kk == best) ? 1.0f : 0.0f;   // one_hot * node_mask (=1 here)
        }
        if (TEAM && tid == 0 && v.misc[TM_FAIL] != 0) atomicOr(&P->a.nan_flags[b], 8);
    }
}

// ---------------------------------------------------------------------------------------------------
// Kernel 3: fused tail of one reverse step for callers that drive the loop from the host.
// ---------------------------------------------------------------------------------------------------
__global__ void sampler_step_kernel(int total, int D, const float* __restrict__ z_t, const float* __restrict__ eps_hat,
                                    const float* __restrict__ noise, const float* __restrict__ frag,
                                    const float* __restrict__ lmask, dl_step_coef cf, float* __restrict__ z_s) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int n = e / D;
    const float lm = lmask[n];
    const float zt = z_t[e];
    const float eh = __fmul_rn(eps_hat[e], lm);
    const float mu = __fsub_rn(__fdiv_rn(zt, cf.alpha_ts), __fmul_rn(cf.c_eps, eh));
    const float zs = __fadd_rn(mu, __fmul_rn(cf.sigma, __fmul_rn(noise[e], lm)));
    z_s[e] = __fadd_rn(__fmul_rn(zt, frag[n]), __fmul_rn(zs, lm));
}

// ---------------------------------------------------------------------------------------------------
// Host side: weight packing and the C ABI
// ---------------------------------------------------------------------------------------------------
thread_local int g_last_hip = 0;
#ifdef DL_TEST_HOOKS
std::atomic<int> g_team_fault{0};          // tests only (dl_debug_team_fault): that many of the NEXT team launches fail (member 1 gives up at once)
#endif
unsigned long long* g_prof_buf = nullptr;   // diagnostics only (dl_set_profile_buffer)

// one injected failure per team launch while the counter is positive (it cannot stay on by accident: ADVICE round 3)
// (the product library has no such switch: the hook exists in -DDL_TEST_HOOKS builds only - libdifflinker_hip_testhooks.so,
// loaded by tests/test_gpu_team.py alone)
inline int take_team_fault() {
#ifdef DL_TEST_HOOKS
    int n = g_team_fault.load();
    while (n > 0 && !g_team_fault.compare_exchange_weak(n, n - 1)) {}
    return n > 0 ? 1 : 0;
#else
    return 0;
#endif
}

inline bool hip_ok(hipError_t e) {
    if (e != hipSuccess) { g_last_hip = int(e); return false; }
    return true;
}

// node-fragment order: unit[nt][sg][lane][j] = W[f = 32nt + (lane&31)][k = 4sg + j + 64(lane>>5)]
void pack_unit(float* dst, const float* w, int ld, int col0, double scale) {
    for (int nt = 0; nt < 4; ++nt)
        for (int sg = 0; sg < 16; ++sg)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int f = 32 * nt + (lane & 31);
                    const int k = 4 * sg + j + 64 * (lane >> 5);
                    dst[((nt * 16 + sg) * 64 + lane) * 4 + j] = float(double(w[size_t(f) * ld + col0 + k]) * scale);
                }
}

// LDS image: img[k][c][nt] = W[f = 32nt + c][k]
void pack_lds_image(float* dst, const float* w, int ld, double scale) {
    for (int k = 0; k < HID; ++k)
        for (int c = 0; c < 32; ++c)
            for (int nt = 0; nt < 4; ++nt)
                dst[(k * 32 + c) * 4 + nt] = float(double(w[size_t(32 * nt + c) * ld + k]) * scale);
}

// ---- f16x3 packing: W' = scale*W is multiplied by a power of two sw that puts max|W'| into [2^14, 2^15) and split
// into fp16 hi (RNE) + lo (RNE of the exact remainder); the device rescales accumulators by 1/(sa*sw).
inline double f16_weight_scale(const float* w, int ld, int col0, int ncols, double scale) {
    double m = 0.0;
    for (int f = 0; f < HID; ++f)
        for (int k = 0; k < ncols; ++k) m = fmax(m, fabs(double(w[size_t(f) * ld + col0 + k]) * scale));
    if (!(m > 0.0) || !std::isfinite(m)) return 1.0;
    int e;
    frexp(m, &e);                                    // m < 2^e
    e = 15 - e;
    if (e > 60) e = 60;
    if (e < -60) e = -60;
    return ldexp(1.0, e);
}
inline void split_f16(double v, uint16_t& hi, uint16_t& lo) {
    const float x = float(v);
    const _Float16 h = static_cast<_Float16>(x);
    const _Float16 l = static_cast<_Float16>(x - static_cast<float>(h));
    memcpy(&hi, &h, 2);
    memcpy(&lo, &l, 2);
}

// f16x3 node-fragment order: unit[nt][part*8 + slab][lane][e] (fp16), value = part(sw[nt]*W'[f = 32nt + (lane&31)][k]),
// k = 16*slab + 8*(lane>>5) + e; same bytes as the fp32 unit (64 KB), read as 16 x dwordx4 per lane.  One power-of-two scale
// per 32-row output tile (`sw4`): with the hidden features renumbered by magnitude (balance_hidden) the rows of a tile are of
// one size class, and a small row no longer sits tens of binades below a matrix-wide maximum
inline double f16_tile_scale(const float* w, int ld, int col0, int row0, double scale) {
    double m = 0.0;
    for (int f = row0; f < row0 + 32; ++f)
        for (int k = 0; k < HID; ++k) m = fmax(m, fabs(double(w[size_t(f) * ld + col0 + k]) * scale));
    if (!(m > 0.0) || !std::isfinite(m)) return 1.0;
    int e;
    frexp(m, &e);                                    // m < 2^e
    e = 15 - e;
    if (e > 60) e = 60;
    if (e < -60) e = -60;
    return ldexp(1.0, e);
}
// (one scale for the whole matrix: W4', whose rows are not renumbered)
double pack_unit_f16_uniform(float* dstf, const float* w, int ld, int col0, double scale) {
    const double sw = f16_weight_scale(w, ld, col0, HID, scale);
    uint16_t* dst = reinterpret_cast<uint16_t*>(dstf);
    for (int nt = 0; nt < 4; ++nt)
        for (int slab = 0; slab < 8; ++slab)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int f = 32 * nt + (lane & 31);
                    const int k = 16 * slab + 8 * (lane >> 5) + e;
                    uint16_t hi, lo;
                    split_f16(double(w[size_t(f) * ld + col0 + k]) * scale * sw, hi, lo);
                    dst[(((nt * 16 + slab) * 64 + lane) * 8) + e] = hi;
                    dst[(((nt * 16 + 8 + slab) * 64 + lane) * 8) + e] = lo;
                }
    return sw;
}
void pack_unit_f16(float* dstf, const float* w, int ld, int col0, double scale, float* sw4) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(dstf);
    for (int nt = 0; nt < 4; ++nt) {
        const double sw = f16_tile_scale(w, ld, col0, 32 * nt, scale);
        sw4[nt] = float(sw);
        for (int slab = 0; slab < 8; ++slab)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int f = 32 * nt + (lane & 31);
                    const int k = 16 * slab + 8 * (lane >> 5) + e;
                    uint16_t hi, lo;
                    split_f16(double(w[size_t(f) * ld + col0 + k]) * scale * sw, hi, lo);
                    dst[(((nt * 16 + slab) * 64 + lane) * 8) + e] = hi;
                    dst[(((nt * 16 + 8 + slab) * 64 + lane) * 8) + e] = lo;
                }
    }
}

// f16x3 A-operand stream unit (pack_layout.h: G_ST_*): unit[chunk][ot & 3][slab][part][lane][e], ot = 4 chunk + (ot & 3),
// value = part(sw * W'[f = 16 ot + (lane & 15)][col0 + stream_kslot(slab, lane >> 4, e)]); sw: one power of two per 32 output
// rows (`tile` = true: the first-layer matrices, as pack_unit_f16) or per matrix (W4', as pack_unit_f16_uniform)
void pack_stream_f16(float* dstf, const float* w, int ld, int col0, double scale, bool tile) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(dstf);
    const double sw_all = tile ? 1.0 : f16_weight_scale(w, ld, col0, HID, scale);
    for (int ot = 0; ot < 8; ++ot) {
        const double sw = tile ? f16_tile_scale(w, ld, col0, 32 * (ot >> 1), scale) : sw_all;
        for (int slab = 0; slab < 4; ++slab)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int f = 16 * ot + (lane & 15);
                    const int k = stream_kslot(slab, lane >> 4, e);
                    uint16_t hi, lo;
                    split_f16(double(w[size_t(f) * ld + col0 + k]) * scale * sw, hi, lo);
                    const size_t base = (size_t((ot >> 2) * 4 + (ot & 3)) * 4 + slab) * 2;
                    dst[((base + 0) * 64 + lane) * 8 + e] = hi;
                    dst[((base + 1) * 64 + lane) * 8 + e] = lo;
                }
    }
}

// f16x3 LDS image: img[part*8 + slab][nt][lane][e]
double pack_lds_image_f16(float* dstf, const float* w, int ld, double scale) {
    const double sw = f16_weight_scale(w, ld, 0, HID, scale);
    uint16_t* dst = reinterpret_cast<uint16_t*>(dstf);
    for (int slab = 0; slab < 8; ++slab)
        for (int nt = 0; nt < 4; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int f = 32 * nt + (lane & 31);
                    const int k = 16 * slab + 8 * (lane >> 5) + e;
                    uint16_t hi, lo;
                    split_f16(double(w[size_t(f) * ld + k]) * scale * sw, hi, lo);
                    dst[(((slab * 4 + nt) * 64 + lane) * 8) + e] = hi;
                    dst[((((8 + slab) * 4 + nt) * 64 + lane) * 8) + e] = lo;
                }
    return sw;
}

// ---- images of the LDS-resident pair loop (egnn_fc.hip: pair_phase), where the second edge layer is computed transposed:
// W2' is the A operand (rows = output features) and its k-slots follow the accumulator layout of the first layer:
// the lane half hh holds, in register reg of 32-feature tile mt1, input feature 32*mt1 + (reg&3) + 8*(reg>>2) + 4*hh.
inline int feat_of(int mt1, int reg, int hh) { return 32 * mt1 + (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// fp32: img[s2 = 16*mt1 + reg][lane][mt] = W[f = 32*mt + (lane&31)][k = feat_of(mt1, reg, lane>>5)]
void pack_lds_image_t(float* dst, const float* w, int ld, double scale) {
    for (int s2 = 0; s2 < 64; ++s2)
        for (int lane = 0; lane < 64; ++lane)
            for (int mt = 0; mt < 4; ++mt)
                dst[(s2 * 64 + lane) * 4 + mt] =
                    float(double(w[size_t(32 * mt + (lane & 31)) * ld + feat_of(s2 >> 4, s2 & 15, lane >> 5)]) * scale);
}

// f16x3: img[part*8 + slab][mt][lane][e], k-slot (slab, hh, e) = feat_of(slab>>1, 8*(slab&1) + e, hh)
double pack_lds_image_f16_t(float* dstf, const float* w, int ld, double scale) {
    const double sw = f16_weight_scale(w, ld, 0, HID, scale);
    uint16_t* dst = reinterpret_cast<uint16_t*>(dstf);
    for (int slab = 0; slab < 8; ++slab)
        for (int mt = 0; mt < 4; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int f = 32 * mt + (lane & 31);
                    const int k = feat_of(slab >> 1, 8 * (slab & 1) + e, lane >> 5);
                    uint16_t hi, lo;
                    split_f16(double(w[size_t(f) * ld + k]) * scale * sw, hi, lo);
                    dst[(((slab * 4 + mt) * 64 + lane) * 8) + e] = hi;
                    dst[((((8 + slab) * 4 + mt) * 64 + lane) * 8) + e] = lo;
                }
    return sw;
}

// largest row L1 norm of scale * W[:, col0 : col0 + 128] (a bound: rounded up)
float row_l1(const float* w, int ld, int col0, double scale) {
    double m = 0.0;
    for (int f = 0; f < HID; ++f) {
        double r = 0.0;
        for (int k = 0; k < HID; ++k) r += fabs(double(w[size_t(f) * ld + col0 + k]) * scale);
        m = fmax(m, r);
    }
    return float(m * 1.0001);
}

// sin_embedding: dst[k][f] = scale * W[f][256 + k], k < 24; returns the largest row L1 norm (a bound on the term: |sin|, |cos| <= 1)
float pack_sin_columns(float* dst, const float* w, int ld, double scale) {
    double m = 0.0;
    for (int f = 0; f < HID; ++f) {
        double r = 0.0;
        for (int k = 0; k < SIN_K; ++k) {
            const double v = double(w[size_t(f) * ld + 2 * HID + k]) * scale;
            dst[k * HID + f] = float(v);
            r += fabs(v);
        }
        m = fmax(m, r);
    }
    return float(m * 1.0001);
}

float vec_absmax(const float* v) {
    float m = 0.0f;
    for (int f = 0; f < HID; ++f) m = fmaxf(m, fabsf(v[f]));
    return m;
}

void pack_vec(float* dst, const float* src, int stride, double scale) {
    for (int f = 0; f < HID; ++f) dst[f] = float(double(src[size_t(f) * stride]) * scale);
}

// ---- balanced packing (f16 modes; pack_layout.h: GS_*).  The hidden features of a two-layer MLP in the order of their
// magnitude proxy, largest first: perm[p] = original index of the feature at position p; fac[p] = 2^n of its group of `group`
// consecutive positions, n = how many whole binades the group's largest proxy sits below the overall largest (0 .. 40); the
// features of a zero-padded narrow model are dealt out over all the groups (the group's largest stays its first position).
// `on` = false (exact-fp32 mode): identity, all factors 1 - that mode packs exactly as before.
struct Balance {
    int perm[HID];
    double fac[HID];
};
void balance_hidden(Balance& b, const double* proxy, int group, bool on) {
    for (int p = 0; p < HID; ++p) { b.perm[p] = p; b.fac[p] = 1.0; }
    if (!on) return;
    // (a checkpoint with NaN / inf weights must reach the kernels - and FoundNaNException - not an inconsistent sort order: such
    // features count as the largest)
    double key[HID];
    for (int f = 0; f < HID; ++f) key[f] = std::isfinite(proxy[f]) ? proxy[f] : 1.7e308;
    proxy = key;
    std::stable_sort(b.perm, b.perm + HID, [&](int x, int y) { return proxy[x] > proxy[y]; });
    const double top = proxy[b.perm[0]];
    if (!(top > 0.0) || !std::isfinite(top)) return;
    // a model narrower than the kernels (hidden_nf < 128, zero-padded: proxy exactly 0) has fewer features to share the same
    // number of groups: deal them out evenly - ceil(real / groups) to a group, still in order, the padding behind them - so that
    // a group spans fewer binades instead of most groups holding nothing
    int real = 0;
    while (real < HID && proxy[b.perm[real]] > 0.0) ++real;
    if (real < HID) {
        const int groups = HID / group, per = (real + groups - 1) / groups;
        int sorted[HID];
        memcpy(sorted, b.perm, sizeof(sorted));
        int pad = real;
        for (int g = 0; g < groups; ++g)
            for (int q = 0; q < group; ++q) {
                const int j = g * per + q;
                b.perm[g * group + q] = (q < per && j < real) ? sorted[j] : sorted[pad++];
            }
    }
    for (int p0 = 0; p0 < HID; p0 += group) {
        const double m = proxy[b.perm[p0]];                     // the group's largest
        int n = 0;
        if (m > 0.0 && std::isfinite(m)) {
            int et, em;
            frexp(top, &et); frexp(m, &em);
            n = std::min(std::max(et - em, 0), 40);
        } else n = 40;
        for (int p = p0; p < p0 + group; ++p) b.fac[p] = ldexp(1.0, n);
    }
}
// rows of a first-layer matrix [128][ld] (and its bias) in the balanced order
void permute_rows(std::vector<float>& wp, std::vector<float>& bp, const float* w, const float* bias, int ld, const Balance& b) {
    wp.resize(size_t(HID) * ld); bp.resize(HID);
    for (int p = 0; p < HID; ++p) {
        memcpy(&wp[size_t(p) * ld], w + size_t(b.perm[p]) * ld, size_t(ld) * sizeof(float));
        bp[p] = bias[b.perm[p]];
    }
}
// columns of a second-layer matrix [128][128] in the balanced order, times 2^-n (exact)
void permute_cols(std::vector<float>& wp, const float* w, const Balance& b) {
    wp.resize(size_t(HID) * HID);
    for (int f = 0; f < HID; ++f)
        for (int p = 0; p < HID; ++p) wp[size_t(f) * HID + p] = float(double(w[size_t(f) * HID + b.perm[p]]) / b.fac[p]);
}
// max over output rows of fac[row] * sum_k |scale * W[row][col0 + k]|, k < ncols  (a bound: rounded up)
float row_l1_weighted(const float* w, int ld, int col0, int ncols, double scale, const Balance& b) {
    double m = 0.0;
    for (int p = 0; p < HID; ++p) {
        double r = 0.0;
        for (int k = 0; k < ncols; ++k) r += fabs(double(w[size_t(p) * ld + col0 + k]) * scale);
        m = fmax(m, r * b.fac[p]);
    }
    return float(m * 1.0001);
}
float vec_absmax_weighted(const float* v, const Balance& b) {
    double m = 0.0;
    for (int p = 0; p < HID; ++p) m = fmax(m, fabs(double(v[p])) * b.fac[p]);
    return float(m * 1.0001);
}

}  // namespace

extern "C" {

int32_t dl_abi_version(void) { return DL_ABI_VERSION; }
#ifdef DL_PROFILE
int32_t dl_profile_max_events(void) { return PROF_MAX_EVENTS; }
#else
int32_t dl_profile_max_events(void) { return 0; }      // the phase timeline exists in -DDL_PROFILE builds only
#endif
void dl_set_profile_buffer(void* device_buf) { g_prof_buf = static_cast<unsigned long long*>(device_buf); }
int32_t dl_last_hip_error(void) { return g_last_hip; }
int32_t dl_max_atoms(void) { return NMAX; }
#ifdef DL_TEST_HOOKS
void dl_debug_team_fault(int32_t launches) { g_team_fault.store(launches < 0 ? 0 : launches); }
#endif

const char* dl_error_string(int32_t s) {
    switch (s) {
        case DL_OK: return "ok";
        case DL_ERR_BAD_ARG: return "bad argument";
        case DL_ERR_UNSUPPORTED: return "hyper-parameter not supported by the HIP path";
        case DL_ERR_TOO_MANY_ATOMS: return "molecule exceeds dl_max_atoms()";
        case DL_ERR_HIP: return "HIP runtime error";
        case DL_ERR_NO_DEVICE: return "no HIP device";
        case DL_ERR_ALLOC: return "allocation failed";
        default: return "unknown";
    }
}

static int32_t check_cfg(const dl_config* c) {
    if (!c) return DL_ERR_BAD_ARG;
    if (c->n_dims != 3 || c->hidden_nf != HID) return DL_ERR_UNSUPPORTED;
    if (c->inv_sublayers < 1 || c->inv_sublayers > MAX_SUBLAYERS) return DL_ERR_UNSUPPORTED;
    if ((c->condition_time | 1) != 1) return DL_ERR_BAD_ARG;
    if (c->in_node_nf < 1 || 3 + c->in_node_nf > DMAX || c->in_node_nf > 16) return DL_ERR_UNSUPPORTED;
    if (c->context_node_nf < 0 || c->context_node_nf > CTXMAX) return DL_ERR_UNSUPPORTED;
    if (c->in_node_nf + c->condition_time + c->context_node_nf > FINP) return DL_ERR_UNSUPPORTED;
    if (c->n_layers < 1 || c->n_layers > 64) return DL_ERR_UNSUPPORTED;
    if (!(c->normalization_factor > 0.0f)) return DL_ERR_BAD_ARG;
    if (c->precision != DL_PRECISION_FP32 && c->precision != DL_PRECISION_F16X3 && c->precision != DL_PRECISION_F16X2) return DL_ERR_UNSUPPORTED;
    if ((c->attention | 1) != 1 || (c->tanh | 1) != 1 || (c->aggregation_mean | 1) != 1) return DL_ERR_BAD_ARG;
    if ((c->sin_embedding | 1) != 1) return DL_ERR_BAD_ARG;
    if (c->tanh && !(c->coords_range > 0.0f)) return DL_ERR_BAD_ARG;
    return DL_OK;
}

int32_t dl_model_num_tensors(const dl_config* cfg) {
    if (!cfg) return DL_ERR_BAD_ARG;
    return 4 + cfg->n_layers * (cfg->inv_sublayers * (8 + (cfg->attention ? 2 : 0)) + 5);
}

int32_t dl_model_create(const dl_config* cfg, const float* const* w, int32_t n_tensors, dl_model** out) {
    int32_t st = check_cfg(cfg);
    if (st != DL_OK) return st;
    if (!w || !out || n_tensors != dl_model_num_tensors(cfg)) return DL_ERR_BAD_ARG;
    for (int i = 0; i < n_tensors; ++i)
        if (!w[i]) return DL_ERR_BAD_ARG;
    const int nf = cfg->in_node_nf, fin = nf + cfg->condition_time + cfg->context_node_nf, L = cfg->n_layers, SUB = cfg->inv_sublayers;
    const size_t total = size_t(OFF_BLOCKS) + size_t(L) * block_size(SUB);
    float* hp = static_cast<float*>(calloc(total, sizeof(float)));
    if (!hp) return DL_ERR_ALLOC;
    const bool f16 = cfg->precision != DL_PRECISION_FP32;
    auto image = [&](float* d, const float* ww, int ld, double sc) -> float {
        if (f16) return float(pack_lds_image_f16(d, ww, ld, sc));
        pack_lds_image(d, ww, ld, sc);
        return 1.0f;
    };
    auto image_t = [&](float* d, const float* ww, int ld, double sc) {     // same scale as image(): one sc[] slot serves both
        if (f16) (void)pack_lds_image_f16_t(d, ww, ld, sc);
        else pack_lds_image_t(d, ww, ld, sc);
    };
    const double c = -1.4426950408889634;            // -log2(e): y = c * pre-activation
    const double inv_norm = 1.0 / double(cfg->normalization_factor);
    int ti = 0;
    // embedding [128][fin], bias; embedding_out [fin][128] (first nf rows kept), bias
    const float* ew = w[ti++]; const float* eb = w[ti++];
    const float* ow = w[ti++]; const float* ob = w[ti++];
    for (int f = 0; f < HID; ++f) {
        for (int k = 0; k < fin; ++k) hp[OFF_EMB_W + f * FINP + k] = ew[size_t(f) * fin + k];
        hp[OFF_EMB_B + f] = eb[f];
    }
    for (int o = 0; o < nf; ++o) {
        memcpy(hp + OFF_OUT_W + o * HID, ow + size_t(o) * HID, HID * sizeof(float));
        hp[OFF_OUT_B + o] = ob[o];
    }
    std::vector<float> w1p, b1p, w2p, w3p, b3p, w4p;
    std::vector<double> proxy(HID);
    for (int blk = 0; blk < L; ++blk) {
        float* base = hp + OFF_BLOCKS + size_t(blk) * block_size(SUB);
        for (int gi = 0; gi < SUB; ++gi) {
            float* g = base + gi * GCL_SIZE;
            const float* w1 = w[ti++]; const float* b1 = w[ti++];     // edge_mlp.0 [128][258]
            const float* w2 = w[ti++]; const float* b2 = w[ti++];     // edge_mlp.2 [128][128]
            const float* w3 = w[ti++]; const float* b3 = w[ti++];     // node_mlp.0 [128][256]
            const float* w4 = w[ti++]; const float* b4 = w[ti++];     // node_mlp.2 [128][128]
            const float* watt = nullptr; const float* batt = nullptr;
            if (cfg->attention) { watt = w[ti++]; batt = w[ti++]; }   // att_mlp.0 [1][128], [1]
            const int ld1 = 2 * HID + (cfg->sin_embedding ? SIN_K : 2);
            const int nattr = ld1 - 2 * HID;
            // agg arrives as c * true message sum: the 1/normalization_factor of 'sum' is folded into W3b', the 1/N of 'mean'
            // is applied where the slot partials are added (N is a property of the batch, not of the model)
            const double s3b = cfg->aggregation_mean ? 1.0 : inv_norm;
            // balanced order of the two hidden layers (f16 modes): the edge model's (rows of W1, columns of W2) in k-slabs of 16,
            // the node MLP's (rows of W3, columns of W4) in tiles of 32.  Proxy: what the feature's pre-activation can reach for
            // |h| ~ 1 (the distance columns weighted by a typical squared distance of 32 A^2)
            Balance be, bt;
            for (int f = 0; f < HID; ++f) {
                double r = fabs(double(b1[f]));
                for (int k = 0; k < 2 * HID; ++k) r += fabs(double(w1[size_t(f) * ld1 + k]));
                for (int k = 0; k < nattr; ++k) r += (cfg->sin_embedding ? 1.0 : 32.0) * fabs(double(w1[size_t(f) * ld1 + 2 * HID + k]));
                proxy[f] = r;
            }
            balance_hidden(be, proxy.data(), 16, f16);
            for (int f = 0; f < HID; ++f) {
                double r = fabs(double(b3[f]) * c);
                for (int k = 0; k < HID; ++k) r += fabs(double(w3[size_t(f) * 2 * HID + k]) * c) + fabs(double(w3[size_t(f) * 2 * HID + HID + k]) * s3b);
                proxy[f] = r;
            }
            balance_hidden(bt, proxy.data(), 32, f16);
            permute_rows(w1p, b1p, w1, b1, ld1, be);
            permute_cols(w2p, w2, be);
            permute_rows(w3p, b3p, w3, b3, 2 * HID, bt);
            permute_cols(w4p, w4, bt);
            float* sc = g + G_SCALE;
            if (f16) {
                pack_unit_f16(g + G_W1A, w1p.data(), ld1, 0, c, sc + GS_SW_W1A);
                pack_unit_f16(g + G_W1B, w1p.data(), ld1, HID, c, sc + GS_SW_W1B);
                pack_unit_f16(g + G_W3A, w3p.data(), 2 * HID, 0, c, sc + GS_SW_W3A);
                pack_unit_f16(g + G_W3B, w3p.data(), 2 * HID, HID, s3b, sc + GS_SW_W3B);
                // W4': its rows are the node features themselves (no renumbering there): one scale for the matrix
                sc[4] = float(pack_unit_f16_uniform(g + G_W4, w4p.data(), HID, 0, 1.0 / c));
                // the same matrices as the A-operand stream of the atom-stationary per-atom phases (round 6)
                {
                    // (W3a' and W3b' chunk by chunk: the node MLP's first layer finishes four output tiles before it opens the next)
                    std::vector<float> ua(UNIT), ub(UNIT);
                    pack_stream_f16(ua.data(), w3p.data(), 2 * HID, 0, c, true);
                    pack_stream_f16(ub.data(), w3p.data(), 2 * HID, HID, s3b, true);
                    for (int h2 = 0; h2 < 2; ++h2) {
                        memcpy(g + G_ST_POST + (2 * h2 + 0) * ST_CHUNK, ua.data() + h2 * ST_CHUNK, ST_CHUNK * sizeof(float));
                        memcpy(g + G_ST_POST + (2 * h2 + 1) * ST_CHUNK, ub.data() + h2 * ST_CHUNK, ST_CHUNK * sizeof(float));
                    }
                }
                pack_stream_f16(g + G_ST_POST + 2 * UNIT, w4p.data(), HID, 0, 1.0 / c, false);
                pack_stream_f16(g + G_ST_PRE, w1p.data(), ld1, 0, c, true);
                pack_stream_f16(g + G_ST_PRE + UNIT, w1p.data(), ld1, HID, c, true);
            } else {
                pack_unit(g + G_W1A, w1p.data(), ld1, 0, c); pack_unit(g + G_W1B, w1p.data(), ld1, HID, c);
                pack_unit(g + G_W3A, w3p.data(), 2 * HID, 0, c); pack_unit(g + G_W3B, w3p.data(), 2 * HID, HID, s3b);
                pack_unit(g + G_W4, w4p.data(), HID, 0, 1.0 / c);
                for (int k = 0; k < 16; ++k) sc[GS_SW_W1A + k] = 1.0f;
                sc[4] = 1.0f;
            }
            sc[5] = image(g + G_W2, w2p.data(), HID, 1.0);
            image_t(g + G_W2T, w2p.data(), HID, 1.0);
            for (int s_ = 0; s_ < 8; ++s_) sc[GS_NE + s_] = float(be.fac[16 * s_]);
            for (int nt = 0; nt < 4; ++nt) sc[GS_NT + nt] = float(bt.fac[32 * nt]);
            float* vv = g + G_VEC;
            pack_vec(vv + 0 * HID, b1p.data(), 1, c);
            if (!cfg->sin_embedding) {
                pack_vec(vv + 1 * HID, w1p.data() + 2 * HID, ld1, c);         // radial column
                pack_vec(vv + 2 * HID, w1p.data() + 2 * HID + 1, ld1, c);     // d0 column
            } else {
                (void)pack_sin_columns(g + G_WG, w1p.data(), ld1, c);          // 24 embedded-distance columns
                sc[20] = row_l1_weighted(w1p.data(), ld1, 2 * HID, SIN_K, c, be);     // |sin|, |cos| <= 1: a bound on the term, exponents applied
            }
            pack_vec(vv + 3 * HID, b2, 1, c);
            pack_vec(vv + 4 * HID, b3p.data(), 1, c);
            pack_vec(vv + 5 * HID, b4, 1, 1.0);
            if (watt) { pack_vec(vv + 6 * HID, watt, 1, 1.0 / c); sc[8] = batt[0]; }    // logit = w_att . (u2 / c) + b_att
            sc[6] = vec_absmax(vv + 1 * HID);
            sc[7] = vec_absmax(vv + 2 * HID);
            sc[GS_WRW] = vec_absmax_weighted(vv + 1 * HID, be);
            sc[GS_WDW] = vec_absmax_weighted(vv + 2 * HID, be);
            // bounds for the a-priori scales of the per-atom phases (version 2): row L1 norms of the packed matrices, bias maxima -
            // of the operands AS THEY ENTER the second layers, i.e. with the hidden layers' exponents applied
            sc[12] = row_l1_weighted(w1p.data(), ld1, 0, HID, c, be); sc[13] = row_l1_weighted(w1p.data(), ld1, HID, HID, c, be);
            sc[14] = row_l1_weighted(w3p.data(), 2 * HID, 0, HID, c, bt); sc[15] = row_l1_weighted(w3p.data(), 2 * HID, HID, HID, s3b, bt);
            sc[16] = row_l1(w4p.data(), HID, 0, 1.0 / c);
            sc[17] = vec_absmax_weighted(vv + 0 * HID, be); sc[18] = vec_absmax_weighted(vv + 4 * HID, bt);
            sc[19] = vec_absmax(vv + 5 * HID) * 1.0001f;
        }
        float* e = base + SUB * GCL_SIZE;
        const float* w5 = w[ti++]; const float* b5 = w[ti++];         // coord_mlp.0 [128][258]
        const float* w6 = w[ti++]; const float* b6 = w[ti++];         // coord_mlp.2 [128][128]
        const float* w7 = w[ti++];                                    // coord_mlp.4 [1][128], no bias
        const int ld5 = 2 * HID + (cfg->sin_embedding ? SIN_K : 2);
        Balance bc;
        for (int f = 0; f < HID; ++f) {
            double r = fabs(double(b5[f]));
            for (int k = 0; k < 2 * HID; ++k) r += fabs(double(w5[size_t(f) * ld5 + k]));
            for (int k = 0; k < ld5 - 2 * HID; ++k) r += (cfg->sin_embedding ? 1.0 : 32.0) * fabs(double(w5[size_t(f) * ld5 + 2 * HID + k]));
            proxy[f] = r;
        }
        balance_hidden(bc, proxy.data(), 16, f16);
        permute_rows(w1p, b1p, w5, b5, ld5, bc);
        permute_cols(w2p, w6, bc);
        float* sc = e + E_SCALE;
        if (f16) {
            pack_unit_f16(e + E_W5A, w1p.data(), ld5, 0, c, sc + ES_SW_W5A);
            pack_unit_f16(e + E_W5B, w1p.data(), ld5, HID, c, sc + ES_SW_W5B);
            pack_stream_f16(e + E_ST_PRE, w1p.data(), ld5, 0, c, true);
            pack_stream_f16(e + E_ST_PRE + UNIT, w1p.data(), ld5, HID, c, true);
        } else {
            pack_unit(e + E_W5A, w1p.data(), ld5, 0, c); pack_unit(e + E_W5B, w1p.data(), ld5, HID, c);
            for (int k = 0; k < 8; ++k) sc[ES_SW_W5A + k] = 1.0f;
        }
        sc[2] = image(e + E_W6, w2p.data(), HID, 1.0);
        image_t(e + E_W6T, w2p.data(), HID, 1.0);
        for (int s_ = 0; s_ < 8; ++s_) sc[ES_NE + s_] = float(bc.fac[16 * s_]);
        float* vv = e + E_VEC;
        pack_vec(vv + 0 * HID, b1p.data(), 1, c);
        if (!cfg->sin_embedding) {
            pack_vec(vv + 1 * HID, w1p.data() + 2 * HID, ld5, c);
            pack_vec(vv + 2 * HID, w1p.data() + 2 * HID + 1, ld5, c);
        } else {
            (void)pack_sin_columns(e + E_WG, w1p.data(), ld5, c);
            sc[11] = row_l1_weighted(w1p.data(), ld5, 2 * HID, SIN_K, c, bc);
        }
        pack_vec(vv + 3 * HID, b6, 1, c);
        // s = w7 . SiLU(..): with tanh or the mean the head's raw output is needed, the normalisation follows at run time
        pack_vec(vv + 4 * HID, w7, 1, (cfg->tanh || cfg->aggregation_mean) ? 1.0 / c : inv_norm / c);
        sc[6] = vec_absmax(vv + 1 * HID);
        sc[7] = vec_absmax(vv + 2 * HID);
        sc[ES_WRW] = vec_absmax_weighted(vv + 1 * HID, bc);
        sc[ES_WDW] = vec_absmax_weighted(vv + 2 * HID, bc);
        sc[8] = row_l1_weighted(w1p.data(), ld5, 0, HID, c, bc); sc[9] = row_l1_weighted(w1p.data(), ld5, HID, HID, c, bc);
        sc[10] = vec_absmax_weighted(vv + 0 * HID, bc);
        // bound of the head's output (every arithmetic mode): the second layer on the operands as they enter it, then w7'
        sc[ES_L1_W6] = row_l1(w2p.data(), HID, 0, 1.0);
        sc[ES_B6] = vec_absmax(vv + 3 * HID) * 1.0001f;
        {
            double l1 = 0.0;
            for (int f = 0; f < HID; ++f) l1 += fabs(double(vv[4 * HID + f]));
            sc[ES_W7L1] = float(l1 * 1.0001);
        }
    }
    dl_model* m = static_cast<dl_model*>(calloc(1, sizeof(dl_model)));
    if (!m) { free(hp); return DL_ERR_ALLOC; }
    m->cfg = *cfg;
    if (cfg->sin_embedding)
        for (int blk = 0; blk < L; ++blk) {
            const float* base = hp + OFF_BLOCKS + size_t(blk) * block_size(SUB);
            for (int gi = 0; gi < SUB; ++gi) m->sin_l1[blk * (MAX_SUBLAYERS + 1) + gi] = base[gi * GCL_SIZE + G_SCALE + 20];
            m->sin_l1[blk * (MAX_SUBLAYERS + 1) + SUB] = base[SUB * GCL_SIZE + E_SCALE + 11];
        }
    m->n_floats = total;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { free(hp); free(m); return DL_ERR_NO_DEVICE; }
    if (!hip_ok(hipMalloc(reinterpret_cast<void**>(&m->d_pack), total * sizeof(float))) ||
        !hip_ok(hipMemcpy(m->d_pack, hp, total * sizeof(float), hipMemcpyHostToDevice))) {
        free(hp); free(m); return DL_ERR_HIP;
    }
    free(hp);
    *out = m;
    return DL_OK;
}

void dl_model_destroy(dl_model* m) {
    if (!m) return;
    if (m->d_pack) (void)hipFree(m->d_pack);
    free(m);
}

// ---- workspace (caller-owned, sized by dl_workspace_bytes): [HBM scratch of every workgroup][team exchange buffers][arrival
// words] and launch geometry.  The library allocates nothing after dl_model_create.
static int fc_grid(int32_t B, int32_t team) { return team <= 1 ? B : (B + 7) / 8 * 8 * team; }
static size_t hsave_bytes(int32_t B, int32_t team) { return size_t(fc_grid(B, team)) * HS_STRIDE * sizeof(float); }
static size_t team_rows_bytes(int32_t B) { return size_t(B) * TEAM_MOL_BYTES; }

size_t dl_workspace_bytes(int32_t B, int32_t team) {
    if (B <= 0 || team < 0 || team > TEAM_MAX) return 0;
    size_t n = hsave_bytes(B, team);
    if (team > 1) n += team_rows_bytes(B) + size_t(B) * TEAM_MAX * sizeof(unsigned);
    return n;
}

int32_t dl_team_max_atoms(int32_t team) { return team >= 2 ? NQMAX : NMAX; }

// Largest team (1, 2, 4 or 8 workgroups per molecule) the current device holds for a batch of B with every workgroup
// resident at once (one workgroup per compute unit: LDS).  No margin is kept: the reference's default batch of 64 on teams of
// four is exactly the chip, and halving that team for headroom would cost a third of its throughput every time, while the
// rare co-tenant case is handled where it happens - the cooperative launch refuses a grid the device cannot hold, a team that
// does not assemble fails together (flag bit 3) and the caller re-runs the batch on one compute unit per molecule.
int32_t dl_team_max(int32_t B) {
    if (B <= 0) return 1;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 1;
    const int slots = (B + 7) / 8 * 8;                  // team members sit 8 workgroups apart: whole groups of 8 molecules
    int S = 1;
    while (S < TEAM_MAX && slots * (S * 2) <= cus) S *= 2;
    return S;
}

// splits the caller's workspace; for a team request validates it and zeroes the arrival words on `stream`
struct FcWorkspace {
    float* hsave;
    char* rows;
    unsigned* flags;
    int grid;
};
static int32_t fc_workspace(int32_t B, int32_t team, void* ws, size_t ws_bytes, hipStream_t stream, FcWorkspace* out) {
    if (team > 1 && team != 2 && team != 4 && team != 8) return DL_ERR_BAD_ARG;
    if (team < 1) team = 1;
    if (!ws || ws_bytes < dl_workspace_bytes(B, team) || (reinterpret_cast<uintptr_t>(ws) & 15u) != 0) return DL_ERR_BAD_ARG;
    if (team > 1 && team > dl_team_max(B)) return DL_ERR_BAD_ARG;    // every workgroup of every team must be resident at once
    out->grid = fc_grid(B, team);
    out->hsave = static_cast<float*>(ws);
    out->rows = nullptr; out->flags = nullptr;
    if (team > 1) {
        char* p = static_cast<char*>(ws) + hsave_bytes(B, team);
        out->rows = p;
        out->flags = reinterpret_cast<unsigned*>(p + team_rows_bytes(B));
        if (!hip_ok(hipMemsetAsync(out->flags, 0, size_t(B) * TEAM_MAX * sizeof(unsigned), stream))) return DL_ERR_HIP;
    }
    return DL_OK;
}

// A team's workgroups wait for each other inside the launch: all of them must be resident at once.  The cooperative launch
// checks the grid against the occupancy of the kernel on this device and fails (hipErrorCooperativeLaunchTooLarge) instead of
// starting a launch that could not assemble; what it cannot see - another stream or process holding compute units - ends in
// the bounded wait of team_sync and flag bit 3.
// (DIFFLINKER_TEAM_LAUNCH_PLAIN=1: the same kernel, grid and arguments through hipLaunchKernel - for counter collection only:
// rocprofv3 --pmc of ROCm 7.2 dies with a segmentation fault on a cooperative launch (profiles/r05/README.md); the grid has been
// checked against dl_team_max() either way)
static hipError_t launch_team(const void* kernel, int grid, hipStream_t st, void* args) {
    void* params[] = {args};
    static const bool plain = [] { const char* e = getenv("DIFFLINKER_TEAM_LAUNCH_PLAIN"); return e && e[0] == '1'; }();
    if (plain) return hipLaunchKernel(kernel, dim3(grid), dim3(THREADS), params, 0, st);
    return hipLaunchCooperativeKernel(kernel, dim3(grid), dim3(THREADS), params, 0, st);
}

int32_t dl_egnn_forward_fc_team(const dl_model* m, int32_t B, int32_t N, const float* xh, const float* t,
                                int32_t t_is_scalar, const int8_t* node_mask, const float* linker_mask,
                                const int8_t* edge_mask, const float* context, float* out, int32_t* nan_flags,
                                int32_t team, void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || !xh || !t || !node_mask || !out || !nan_flags || B < 0 || N < 1) return DL_ERR_BAD_ARG;
    if (m->cfg.context_node_nf > 0 && !context) return DL_ERR_BAD_ARG;
    if (m->cfg.sin_embedding) return DL_ERR_UNSUPPORTED;       // sinusoidal distance embedding: dl_egnn_forward_fc_large / _pocket only
    if (B == 0) return DL_OK;
    FwdArgs a;
    a.wpack = m->d_pack; a.md = dims_of(m); a.B = B; a.N = N; a.xh = xh; a.t = t;
    a.t_stride = t_is_scalar ? 0 : 1; a.node_mask = node_mask; a.linker_mask = linker_mask;
    a.edge_mask = edge_mask; a.context = context; a.out = out; a.nan_flags = nan_flags; a.prof = g_prof_buf;
    hipStream_t st = static_cast<hipStream_t>(stream);
    FcWorkspace ws;
    const int32_t rc = fc_workspace(B, team, workspace, workspace_bytes, st, &ws);
    if (rc != DL_OK) return rc;
    a.team = team <= 1 ? 1 : team; a.team_rows = ws.rows; a.team_flags = ws.flags; a.hsave = ws.hsave; a.team_fault = team > 1 ? take_team_fault() : 0;
    // (the two-term GCL loop of F16X2 has no attention variant: an attention model runs its F16X3 kernels)
    const bool f16 = m->cfg.precision != DL_PRECISION_FP32, att = m->cfg.attention != 0, two = m->cfg.precision == DL_PRECISION_F16X2 && !att;
    const void* kernel;
    if (team <= 1) kernel = two ? (const void*)&egnn_forward_fc_kernel<2, false, false>
                          : f16 ? (att ? (const void*)&egnn_forward_fc_kernel<1, false, true> : (const void*)&egnn_forward_fc_kernel<1, false, false>)
                                : (att ? (const void*)&egnn_forward_fc_kernel<0, false, true> : (const void*)&egnn_forward_fc_kernel<0, false, false>);
    else kernel = two ? (const void*)&egnn_forward_fc_kernel<2, true, false>
                : f16 ? (att ? (const void*)&egnn_forward_fc_kernel<1, true, true> : (const void*)&egnn_forward_fc_kernel<1, true, false>)
                      : (att ? (const void*)&egnn_forward_fc_kernel<0, true, true> : (const void*)&egnn_forward_fc_kernel<0, true, false>);
    if (team <= 1) {
        void* params[] = {&a};
        return hip_ok(hipLaunchKernel(kernel, dim3(ws.grid), dim3(THREADS), params, 0, st)) ? DL_OK : DL_ERR_HIP;
    }
    if (!hip_ok(hipMemsetAsync(nan_flags, 0, size_t(B) * sizeof(int32_t), st))) return DL_ERR_HIP;     // the members OR their bits in
    const hipError_t e = launch_team(kernel, ws.grid, st, &a);
    return hip_ok(e) ? DL_OK : DL_ERR_HIP;
}

int32_t dl_egnn_forward_fc(const dl_model* m, int32_t B, int32_t N, const float* xh, const float* t,
                           int32_t t_is_scalar, const int8_t* node_mask, const float* linker_mask,
                           const int8_t* edge_mask, const float* context, float* out, int32_t* nan_flags,
                           void* workspace, size_t workspace_bytes, void* stream) {
    return dl_egnn_forward_fc_team(m, B, N, xh, t, t_is_scalar, node_mask, linker_mask, edge_mask, context, out, nan_flags,
                                   1, workspace, workspace_bytes, stream);
}

int32_t dl_sample_chain_fc(const dl_model* m, const dl_chain_args* g, void* stream) {
    if (!m || !g) return DL_ERR_BAD_ARG;
    if (!g->x || !g->h || !g->node_mask || !g->fragment_mask || !g->linker_mask ||
        !g->coefs || !g->chain || !g->nan_flags || !g->nan_step) return DL_ERR_BAD_ARG;
    if ((g->noise_x == nullptr) != (g->noise_h == nullptr)) return DL_ERR_BAD_ARG;    // both (bank) or neither (Philox)
    if (m->cfg.context_node_nf > 0 && !g->context) return DL_ERR_BAD_ARG;
    if (g->B < 0 || g->N < 1 || g->T < 1 || g->keep_frames < 1 || g->keep_frames > g->T || g->team < 0) return DL_ERR_BAD_ARG;
    if (g->order_first < 0 || g->order_count < 0 || g->order_first + g->order_count > g->B) return DL_ERR_BAD_ARG;
    if ((g->order_first != 0 || g->order_count != 0) && !g->order) return DL_ERR_BAD_ARG;
    if ((g->q_begin || g->q_end) && !g->z_state) return DL_ERR_BAD_ARG;
    if (m->cfg.sin_embedding) return DL_ERR_UNSUPPORTED;       // (host-driven loop over dl_egnn_forward_fc_large instead)
    if (g->B == 0) return DL_OK;
    const int32_t count = g->order_count > 0 ? g->order_count : g->B;        // molecules of this launch
    ChainArgs a;
    a.wpack = m->d_pack; a.md = dims_of(m); a.a = *g; a.prof = g_prof_buf;
    hipStream_t st = static_cast<hipStream_t>(stream);
    FcWorkspace ws;
    const int32_t rc = fc_workspace(count, g->team, g->workspace, g->workspace_bytes, st, &ws);
    if (rc != DL_OK) return rc;
    a.team_rows = ws.rows; a.team_flags = ws.flags; a.hsave = ws.hsave; a.team_fault = g->team > 1 ? take_team_fault() : 0;
    const bool f16 = m->cfg.precision != DL_PRECISION_FP32, att = m->cfg.attention != 0, two = m->cfg.precision == DL_PRECISION_F16X2 && !att;
    const void* kernel;
    if (g->team <= 1) kernel = two ? (const void*)&sample_chain_fc_kernel<2, false, false>
                             : f16 ? (att ? (const void*)&sample_chain_fc_kernel<1, false, true> : (const void*)&sample_chain_fc_kernel<1, false, false>)
                                   : (att ? (const void*)&sample_chain_fc_kernel<0, false, true> : (const void*)&sample_chain_fc_kernel<0, false, false>);
    else kernel = two ? (const void*)&sample_chain_fc_kernel<2, true, false>
                : f16 ? (att ? (const void*)&sample_chain_fc_kernel<1, true, true> : (const void*)&sample_chain_fc_kernel<1, true, false>)
                      : (att ? (const void*)&sample_chain_fc_kernel<0, true, true> : (const void*)&sample_chain_fc_kernel<0, true, false>);
    if (g->team <= 1) {
        a.a.team = 1;
        void* params[] = {&a};
        return hip_ok(hipLaunchKernel(kernel, dim3(ws.grid), dim3(THREADS), params, 0, st)) ? DL_OK : DL_ERR_HIP;
    }
    if (!hip_ok(hipMemsetAsync(g->nan_flags, 0, size_t(g->B) * sizeof(int32_t), st)) ||
        !hip_ok(hipMemsetAsync(g->nan_step, 0xFF, size_t(g->B) * sizeof(int32_t), st))) return DL_ERR_HIP;
    const hipError_t e = launch_team(kernel, ws.grid, st, &a);
    return hip_ok(e) ? DL_OK : DL_ERR_HIP;
}

namespace {
// one 256-thread workgroup per molecule; D = 3 + nf <= DMAX
__global__ void __launch_bounds__(256) inpaint_step_kernel(int N, int nf, const float* __restrict__ z_t,
        const float* __restrict__ eps_hat, const float* __restrict__ xh_frag, const float* __restrict__ npx,
        const float* __restrict__ nph, const float* __restrict__ nqx, const float* __restrict__ nqh,
        const float* __restrict__ node_mask, const float* __restrict__ fragment_mask,
        const float* __restrict__ linker_mask, dl_inpaint_coef cf, float* __restrict__ z_s) {
    __shared__ float red[12][4];                 // per-wave partial sums
    __shared__ float tot[12];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int D = 3 + nf;
    const size_t base = size_t(b) * N;
    // sums over atoms: [0..2] noise_p.x * node, [3..5] noise_q.x * qmask, [6..8] eps.x (masked by the denoiser), 9: #node, 10: #qmask
    float part[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) part[k] = 0.0f;
    for (int a = tid; a < N; a += 256) {
        const float nm = node_mask[base + a];
        const float qm = cf.decode ? nm : fragment_mask[base + a];     // q(x,h|z_0) draws its noise on the node mask
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            part[d] += npx[(base + a) * 3 + d] * nm;
            part[3 + d] += nqx[(base + a) * 3 + d] * qm;
            part[6 + d] += eps_hat[(base + a) * D + d];
        }
        part[9] += nm;
        part[10] += qm;
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        float s_ = part[k];
        for (int off = 32; off >= 1; off >>= 1) s_ += __shfl_xor(s_, off);
        if (lane == 0) red[k][wv] = s_;
    }
    __syncthreads();
    if (tid < 11) tot[tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    __syncthreads();
    const float n_node = tot[9], n_q = tot[10];
    // new state (before the centre-of-gravity projection), and the sum of its positions over the node mask
    float zsum[3] = {0.0f, 0.0f, 0.0f};
    for (int e = tid; e < N * D; e += 256) {
        const int a = e / D, d = e - a * D;
        const float nm = node_mask[base + a], fm = fragment_mask[base + a], lm = linker_mask[base + a];
        const float qm = cf.decode ? nm : fm;
        const float z = z_t[(base + a) * D + d];
        float eps = eps_hat[(base + a) * D + d];
        float np_, nq_;
        if (d < 3) {
            eps = __fsub_rn(eps, __fmul_rn(__fdiv_rn(tot[6 + d], n_node), nm));              // centred velocity
            np_ = __fsub_rn(__fmul_rn(npx[(base + a) * 3 + d], nm), __fmul_rn(__fdiv_rn(tot[d], n_node), nm));
            nq_ = __fsub_rn(__fmul_rn(nqx[(base + a) * 3 + d], qm), __fmul_rn(__fdiv_rn(tot[3 + d], n_q), qm));
        } else {
            np_ = __fmul_rn(nph[(base + a) * nf + d - 3], nm);
            nq_ = __fmul_rn(nqh[(base + a) * nf + d - 3], qm);
        }
        float zl, zf;
        if (!cf.decode) {
            zl = __fadd_rn(__fsub_rn(__fdiv_rn(z, cf.alpha_ts), __fmul_rn(cf.c_eps, eps)), __fmul_rn(cf.sigma, np_));
            zf = __fadd_rn(__fadd_rn(__fmul_rn(cf.a_q, z), __fmul_rn(cf.b_q, xh_frag[(base + a) * D + d])),
                           __fmul_rn(cf.sigma, nq_));
        } else {
            zl = __fadd_rn(__fmul_rn(cf.inv_alpha0, __fsub_rn(z, __fmul_rn(cf.sigma0, eps))), __fmul_rn(cf.sigma_x, np_));
            zf = __fsub_rn(__fmul_rn(cf.inv_alpha0, z), __fmul_rn(__fmul_rn(cf.sigma0, cf.inv_alpha0), nq_));
            zl = (d < 3) ? __fmul_rn(zl, cf.norm_x) : __fadd_rn(__fmul_rn(zl, cf.norm_h), cf.bias_h);
            zf = (d < 3) ? __fmul_rn(zf, cf.norm_x) : __fadd_rn(__fmul_rn(zf, cf.norm_h), cf.bias_h);
        }
        const float zn = __fadd_rn(__fmul_rn(zl, lm), __fmul_rn(zf, fm));
        z_s[(base + a) * D + d] = zn;
        if (d < 3) zsum[d] += zn;                   // z_s is zero outside the node mask (lm + fm = node mask)
    }
    if (!cf.decode) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float s_ = zsum[k];
            for (int off = 32; off >= 1; off >>= 1) s_ += __shfl_xor(s_, off);
            if (lane == 0) red[k][wv] = s_;
        }
        __syncthreads();
        if (tid < 3) tot[tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
        __syncthreads();
        for (int e = tid; e < N * 3; e += 256) {
            const int a = e / 3, d = e - a * 3;
            const float nm = node_mask[base + a];
            float* p = z_s + (base + a) * D + d;
            *p = __fsub_rn(*p, __fmul_rn(__fdiv_rn(tot[d], n_node), nm));
        }
    } else {
        __syncthreads();                             // one-hot of the features, per atom (first maximal index), * node mask
        for (int a = tid; a < N; a += 256) {
            float* h = z_s + (base + a) * D + 3;
            const float nm = node_mask[base + a];
            int best = 0;
            float bv = h[0];
            for (int k = 1; k < nf; ++k)
                if (h[k] > bv) { bv = h[k]; best = k; }
            for (int k = 0; k < nf; ++k) h[k] = (k == best) ? nm : 0.0f;
        }
    }
}

__global__ void philox_fill_kernel(unsigned long long seed, int mol_offset, const int* __restrict__ mol_index, int B, int N, int nf,
                                   int draw0, int n_draws, float* noise_x, float* noise_h) {
    const int D = 3 + nf;
    const long long total = (long long)n_draws * B * N * D;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int d = int(e % D);
        const long long node = e / D;                      // (k * B + b) * N + n
        const int n = int(node % N);
        const int b = int((node / N) % B);
        const int k = int(node / ((long long)N * B));
        const float val = philox_normal(seed, unsigned(mol_offset + (mol_index ? mol_index[b] : b)), unsigned(n), unsigned(draw0 + k), unsigned(d));
        if (d < 3) noise_x[node * 3 + d] = val;
        else noise_h[node * nf + d - 3] = val;
    }
}
}  // namespace

int32_t dl_inpaint_step(int32_t B, int32_t N, int32_t nf, const float* z_t, const float* eps_hat, const float* xh_frag,
                        const float* noise_px, const float* noise_ph, const float* noise_qx, const float* noise_qh,
                        const float* node_mask, const float* fragment_mask, const float* linker_mask,
                        dl_inpaint_coef coef, float* z_s, void* stream) {
    if (!z_t || !eps_hat || !xh_frag || !noise_px || !noise_ph || !noise_qx || !noise_qh || !node_mask ||
        !fragment_mask || !linker_mask || !z_s || B < 0 || N < 1 || nf < 1 || 3 + nf > DMAX) return DL_ERR_BAD_ARG;
    if (B == 0) return DL_OK;
    hipLaunchKernelGGL(inpaint_step_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), N, nf, z_t, eps_hat,
                       xh_frag, noise_px, noise_ph, noise_qx, noise_qh, node_mask, fragment_mask, linker_mask, coef, z_s);
    return hip_ok(hipGetLastError()) ? DL_OK : DL_ERR_HIP;
}

int32_t dl_philox_fill(uint64_t seed, int32_t mol_offset, const int32_t* mol_index, int32_t B, int32_t N, int32_t nf, int32_t draw0,
                       int32_t n_draws, float* noise_x, float* noise_h, void* stream) {
    if (!noise_x || !noise_h || B < 0 || N < 1 || nf < 1 || n_draws < 0 || draw0 < 0 || mol_offset < 0) return DL_ERR_BAD_ARG;
    const long long total = (long long)n_draws * B * N * (3 + nf);
    if (total == 0) return DL_OK;
    const int blocks = int(std::min<long long>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(philox_fill_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (unsigned long long)seed, mol_offset, mol_index, B, N, nf, draw0, n_draws, noise_x, noise_h);
    return hip_ok(hipGetLastError()) ? DL_OK : DL_ERR_HIP;
}

int32_t dl_sampler_step(int32_t B, int32_t N, int32_t D, const float* z_t, const float* eps_hat, const float* noise,
                        const float* fragment_mask, const float* linker_mask, dl_step_coef coef, float* z_s,
                        void* stream) {
    if (!z_t || !eps_hat || !noise || !fragment_mask || !linker_mask || !z_s || B < 0 || N < 1 || D < 1)
        return DL_ERR_BAD_ARG;
    const int total = B * N * D;
    if (total == 0) return DL_OK;
    hipLaunchKernelGGL(sampler_step_kernel, dim3((total + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), total, D, z_t, eps_hat, noise, fragment_mask,
                       linker_mask, coef, z_s);
    return hip_ok(hipGetLastError()) ? DL_OK : DL_ERR_HIP;
}

}  // extern "C"
